#!/bin/bash
# Round 5, GPU call 6 (last): counters on the benchmark's workload with one view group (see tools/r05/pmc_bench.sh).
set -u
OUT=gpurun_out/r05_call6; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/r05/pmc_bench.sh "$OUT/pmc" 2>&1 | tail -120
