#!/bin/bash
# Round 4, GPU call 9: kernel choice per diagonal launch.  The kernels give the same bits, so a batch may sweep its short diagonals (and its coarse levels) with the
# speculative kernels and its long ones with pm_sweep2: PMHIP_WIDE_PIXELS / PMHIP_WIDE8_PIXELS = largest launch (pixels) that uses the two-wide / eight-wide kernel.
set -u
OUT=gpurun_out/r04_call9; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 700 python tools/r04/probe_lanes.py 100 "sweep2 only:" "wide<=8k:PMHIP_WIDE_PIXELS=8000" "wide<=16k:PMHIP_WIDE_PIXELS=16000" "wide<=24k:PMHIP_WIDE_PIXELS=24000" "wide<=32k:PMHIP_WIDE_PIXELS=32000" \
   "wide<=16k wide8<=1k:PMHIP_WIDE_PIXELS=16000,PMHIP_WIDE8_PIXELS=1000" "wide<=16k wide8<=3k:PMHIP_WIDE_PIXELS=16000,PMHIP_WIDE8_PIXELS=3000" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_100.log"
timeout 300 python tools/r04/probe_lanes.py 13 "widen2 only:" "wide8<=500:PMHIP_WIDE8_PIXELS=500" "wide8<=1500:PMHIP_WIDE8_PIXELS=1500" "wide8<=3000:PMHIP_WIDE8_PIXELS=3000" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_13.log"
timeout 300 python tools/r04/probe_lanes.py 50 "widen2 only:" "sweep2 only:PMHIP_WIDE=0" "sweep2, wide<=16k:PMHIP_WIDE=0,PMHIP_WIDE_PIXELS=16000" "sweep2, wide<=24k:PMHIP_WIDE=0,PMHIP_WIDE_PIXELS=24000" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_50.log"
