#!/bin/bash
# Round 3, GPU call 3: quad images (one 16-byte load per bilinear sample) in the window-less tap rows: per-diagonal kernel and band kernel, 100 views;
# SQ counters of the band kernel.
set -u
OUT=gpurun_out/r03_call3; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_patchmatch.py -m gpu -q -x -k "views_per_lane or N8 or geometric or config2" > "$OUT/gpu_patchmatch_band.log" 2>&1; echo "parity rc $?"; tail -3 "$OUT/gpu_patchmatch_band.log"
V="libpmhip.so:2:16:0 libpmhip_nt.so:2:16:0 libpmhip_nt.so:2:4:0 libpmhip.so:1:16:1 libpmhip.so:1:4:1 libpmhip_bmw4.so:1:16:1 libpmhip_bmw4.so:1:4:1"
VARIANTS="$V" bash tools/gpu_call.sh r03_call3 variants
PMC_GROUPS=1 PMHIP_BAND=1 bash tools/pmc/run_pmc_sq.sh "$OUT/pmc_band" 24 libpmhip.so 2>&1 | tail -40
