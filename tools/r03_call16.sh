#!/bin/bash
# Round 3, GPU call 16: the one-wave-per-pixel kernel compiled for 3 (default now: 165 / 167 VGPRs, it was 2 for the geometric instantiation) and 4 waves per
# SIMD (128 VGPRs, 144 B scratch), batches of 1 .. 13 views.
set -u
OUT=gpurun_out/r03_call16; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for lib in libpmhip.so libpmhip_wmw4.so libpmhip_wmw2.so; do
  echo "PMHIP_LIB=$lib" | tee -a "$OUT/small.log"
  PMHIP_LIB=$PWD/openmvs_amd/$lib timeout 500 python tools/small_batch_probe.py 1 4 8 13 2>&1 | grep "batch" | tee -a "$OUT/small.log"
done
