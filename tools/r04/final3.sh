#!/bin/bash
# Round 4, last evidence run on the final tree (after the SGM ragged-range aggregation, the views of different sizes with masks, and the reverted init-kernel experiments):
# gpu suite, smoke, default bench line, rocprof kernel stats of the same command.  Counters: profiles/traffic.json, r04_final_pmc/ (kernel sources unchanged: same digest).
set -u
OUT=gpurun_out/r04_final3; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
step() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/steps.log"; }
step "gpu suite"
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > "$OUT/gpu_suite.log" 2>&1; echo "suite exit $?" | tee -a "$OUT/gpu_suite.log"; tail -6 "$OUT/gpu_suite.log"
step "smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee "$OUT/smoke.log"
step "bench"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"; head -c 300 "$OUT/bench.json"; echo; tail -2 "$OUT/bench.err"
step "rocprof kernel stats of bench.py"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python "$R/bench.py" --no-cpu-baseline --no-extras > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/rocprof.err" ); echo "rc $?"
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \; ; rm -rf /tmp/prof_stats
grep "pm_" "$OUT/bench_kernel_stats.csv" | head -6 | cut -c1-160
step done
