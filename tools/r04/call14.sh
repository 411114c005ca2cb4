#!/bin/bash
# Round 4, GPU call 14: rocprof kernel stats of the bench with the anti-diagonal init kernel (how long does pm_init_kernel take now?)
set -u
OUT=gpurun_out/r04_call14; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python "$R/bench.py" --no-cpu-baseline --no-extras > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/rocprof.err" ); echo "rc $?"
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \; ; rm -rf /tmp/prof_stats
grep "pm_" "$OUT/bench_kernel_stats.csv" | cut -c1-200; head -c 200 "$OUT/bench_under_rocprof.json"; echo
