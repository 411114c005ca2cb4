#!/bin/bash
# Round 4, GPU call 10: why are four view groups so much slower than two or three (13 views: 15.7 against 26.9 Mpix/s, profiles/r04_call9_lanes_13.log) when the host is not
# the limit (tools/probes/launch_rate.hip: 2.5-3.5 us per launch, four streams of 30 us kernels overlap perfectly)?  Hypothesis: hardware queues -- HIP maps streams onto
# GPU_MAX_HW_QUEUES = 4 queues, and N groups use N + 1 streams (the engine's own stream idles while the groups sweep) next to the process's null stream.
# (a) GPU_MAX_HW_QUEUES=8; (b) PMHIP_G0_MAIN=1: group 0 sweeps on the engine's own stream.  Also the new defaults (per-launch kernel choice) once more at 100 views.
set -u
OUT=gpurun_out/r04_call10; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
P="timeout 700 python tools/r04/probe_lanes.py"
$P 13 "default:" "groups4:PMHIP_GROUPS=4" "groups4 g0main:PMHIP_GROUPS=4,PMHIP_G0_MAIN=1" "groups3 g0main:PMHIP_GROUPS=3,PMHIP_G0_MAIN=1" "groups2 g0main:PMHIP_G0_MAIN=1" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_13.log"
GPU_MAX_HW_QUEUES=8 $P 13 "hwq8 default:" "hwq8 groups4:PMHIP_GROUPS=4" "hwq8 groups6:PMHIP_GROUPS=6" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_13.log"
$P 100 "default:" "groups3 g0main:PMHIP_GROUPS=3,PMHIP_G0_MAIN=1" "groups4 g0main:PMHIP_GROUPS=4,PMHIP_G0_MAIN=1" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_100.log"
GPU_MAX_HW_QUEUES=8 $P 100 "hwq8 groups4:PMHIP_GROUPS=4" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_100.log"
