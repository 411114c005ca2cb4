#!/bin/bash
# Round 4, GPU call 8: whole gpu suite and the default bench line on the state with pipelined tap rows and the two-wide kernel up to 64 views.
set -u
OUT=gpurun_out/r04_call8; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > "$OUT/gpu_suite.log" 2>&1; echo "suite exit $?" | tee -a "$OUT/gpu_suite.log"; tail -15 "$OUT/gpu_suite.log"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"; head -c 600 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
