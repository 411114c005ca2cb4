#!/bin/bash
# Round 3, GPU call 9: evidence for the final tree: memory-side and SQ counters of the default sweep kernel (24 views, as call 1), rocprofv3 kernel stats of
# bench.py, the bench line, and the config 5 slice after the component-size fix.
set -u
OUT=gpurun_out/r03_call9; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PMC_TIMEOUT=200 bash tools/pmc/run_pmc.sh "$OUT/pmc" 24 libpmhip.so 2>&1 | grep -v "^W2026" | tail -30
timeout 600 python tools/config5_probe.py 32 > "$OUT/config5_32_views_4k.json" 2> "$OUT/config5.err"; tail -c 700 "$OUT/config5_32_views_4k.json"
BENCH_ARGS="--steps 2 --warmup 1 --no-extras" bash tools/gpu_call.sh r03_call9 prof
BENCH_ARGS="--steps 3 --warmup 1" bash tools/gpu_call.sh r03_call9 bench
