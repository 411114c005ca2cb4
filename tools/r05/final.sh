#!/bin/bash
# Round 5, final evidence run (one gpurun call): the whole gpu suite at HEAD, the default bench line, rocprofv3 kernel stats of bench.py, and the N-rank path on one GPU.
set -u
OUT=gpurun_out/r05_final; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout 420 python -m pytest tests -m gpu -q --durations=6 > "$OUT/gpu_suite.log" 2>&1; echo "suite rc $?"; tail -12 "$OUT/gpu_suite.log"
timeout 360 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"; tail -c 600 "$OUT/bench.json"; tail -2 "$OUT/bench.err"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-shard-rates \
    > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/rocprof.err" ); echo "rocprof rc $?"
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \; ; rm -rf /tmp/prof_stats
head -8 "$OUT/bench_kernel_stats.csv"; tail -c 300 "$OUT/bench_under_rocprof.json"
timeout 200 bash tools/r05/ranks_on_one_gpu.sh 2>&1 | tail -8
