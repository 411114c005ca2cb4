#!/bin/bash
# Round 3, GPU call 6: per-diagonal launches with pm_band.hip's visit body (pm_sweep2_kernel: state in LDS, 131-145 VGPRs, no scratch; 128 with
# PM_BAND_MINWAVES=4) against pm_sweep_kernel without windows, 100 and 13 views.  Parity of the new kernel first.
set -u
OUT=gpurun_out/r03_call6; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
PMHIP_BAND=0 timeout 600 python -m pytest tests/test_gpu_patchmatch.py -m gpu -q -x -k "views_per_lane or N8 or geometric or config2 or mask or batch" > "$OUT/gpu_patchmatch_sweep2.log" 2>&1; echo "parity rc $?"; tail -3 "$OUT/gpu_patchmatch_sweep2.log"
V="libpmhip_nt.so:2:4:0:::0 libpmhip.so:2:4:0:::1 libpmhip.so:2:16:0:::1 libpmhip_bmw4.so:2:4:0:::1 libpmhip_bmw4.so:2:16:0:::1 libpmhip_bmw4.so:3:4:0:::1 libpmhip.so:1:4:0:::1"
VARIANTS="$V" bash tools/gpu_call.sh r03_call6 variants
SMALL_VIEWS=13 SMALL_VARIANTS="libpmhip_nt.so:2:16:0:::0 libpmhip.so:2:16:0:::1 libpmhip_bmw4.so:2:16:0:::1 libpmhip.so:1:16:0:::1" bash tools/gpu_call.sh r03_call6 small
