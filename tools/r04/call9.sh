#!/bin/bash
# Round 4, GPU call 9: (a) is a sweep of a small batch bound by the host's launch rate?  tools/probes/launch_rate.hip: host microseconds per launch with one thread feeding
# 1 / 2 / 4 streams, and with one thread per stream; the engine with PMHIP_LAUNCH_THREADS (one enqueueing thread per view group).  (b) kernel choice per diagonal launch: the
# kernels give the same bits, so a batch may sweep its short diagonals (and its coarse levels) with the speculative kernels and its long ones with pm_sweep2:
# PMHIP_WIDE_PIXELS / PMHIP_WIDE8_PIXELS = largest launch (pixels) that uses the two-wide / eight-wide kernel.
set -u
OUT=gpurun_out/r04_call9; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
hipcc --offload-arch=gfx950 -O2 tools/probes/launch_rate.hip -o /tmp/launch_rate 2>/dev/null && timeout 200 /tmp/launch_rate 2>&1 | tee "$OUT/launch_rate.log"
timeout 300 python -m pytest tests/test_gpu_patchmatch.py -m gpu -q -k "tuning" 2>&1 | tail -3 | tee "$OUT/tuning_test.log"
P="timeout 700 python tools/r04/probe_lanes.py"
$P 13 "widen2 only:" "threads2:PMHIP_LAUNCH_THREADS=2" "groups4 threads4:PMHIP_GROUPS=4,PMHIP_LAUNCH_THREADS=4" "groups3 threads3:PMHIP_GROUPS=3,PMHIP_LAUNCH_THREADS=3" \
   "wide8<=1500:PMHIP_WIDE8_PIXELS=1500" "wide8<=1500 threads2:PMHIP_WIDE8_PIXELS=1500,PMHIP_LAUNCH_THREADS=2" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_13.log"
$P 25 "widen2 only:" "threads2:PMHIP_LAUNCH_THREADS=2" "groups4 threads4:PMHIP_GROUPS=4,PMHIP_LAUNCH_THREADS=4" "groups3 threads3:PMHIP_GROUPS=3,PMHIP_LAUNCH_THREADS=3" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_25.log"
$P 100 "sweep2 only:" "threads2:PMHIP_LAUNCH_THREADS=2" "wide<=8k:PMHIP_WIDE_PIXELS=8000" "wide<=16k:PMHIP_WIDE_PIXELS=16000" "wide<=24k:PMHIP_WIDE_PIXELS=24000" \
   "wide<=16k wide8<=2k threads2:PMHIP_WIDE_PIXELS=16000,PMHIP_WIDE8_PIXELS=2000,PMHIP_LAUNCH_THREADS=2" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_100.log"
$P 50 "widen2 only:" "threads2:PMHIP_LAUNCH_THREADS=2" "sweep2, wide<=16k:PMHIP_WIDE=0,PMHIP_WIDE_PIXELS=16000" "sweep2, wide<=16k threads2:PMHIP_WIDE=0,PMHIP_WIDE_PIXELS=16000,PMHIP_LAUNCH_THREADS=2" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_50.log"
