#!/bin/bash
# Round 4, GPU call 18: three kernels per batch -- launches above widePixels but of at most PMHIP_MID_PIXELS pixels through pm_sweep2_kernel<8,1> (one view per lane) instead of <4,2>
set -u
OUT=gpurun_out/r04_call18; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PROBE_STEPS=1
timeout 500 python tools/r04/probe_lanes.py 100 "two kernels:" "mid<=30k:PMHIP_MID_PIXELS=30000" "mid<=40k:PMHIP_MID_PIXELS=40000" "two kernels again:" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_100.log"
