#!/bin/bash
# Round 5, GPU call 1: view groups that run their whole pass on their own stream and start out of phase (pm_engine.hip estimateClass, PMHipTuning::groupOffset).
# Full schedule of a V-view 1080p scene for several (groups, offset per mille) settings; bit-comparison of view 0 against the first configuration; then the two parity
# tests that exercise the new schedule (tuning through the ABI; 81 golden maps incl. the timed mix with 2 and 3 groups out of phase).
set -u
OUT=gpurun_out/r05_call1; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
P="timeout 900 python tools/r05/probe_groups.py"
$P 100 "lockstep start:PMHIP_GROUP_OFFSET=0" "off 50:PMHIP_GROUP_OFFSET=50" "off 100:PMHIP_GROUP_OFFSET=100" "off 160:PMHIP_GROUP_OFFSET=160" "off 250:PMHIP_GROUP_OFFSET=250" "off 400:PMHIP_GROUP_OFFSET=400" \
   "3 groups off 160:PMHIP_GROUPS=3,PMHIP_GROUP_OFFSET=160" "3 groups off 330:PMHIP_GROUPS=3,PMHIP_GROUP_OFFSET=330" "1 group:PMHIP_GROUPS=1" 2>&1 | grep -v amdgpu.ids | tee "$OUT/groups_100.log"
$P 50 "lockstep start:PMHIP_GROUP_OFFSET=0" "off 100:PMHIP_GROUP_OFFSET=100" "off 160:PMHIP_GROUP_OFFSET=160" "off 250:PMHIP_GROUP_OFFSET=250" "3 groups off 160:PMHIP_GROUPS=3,PMHIP_GROUP_OFFSET=160" 2>&1 | grep -v amdgpu.ids | tee "$OUT/groups_50.log"
$P 25 "lockstep start:PMHIP_GROUP_OFFSET=0" "off 50:PMHIP_GROUP_OFFSET=50" "off 100:PMHIP_GROUP_OFFSET=100" "off 160:PMHIP_GROUP_OFFSET=160" "off 250:PMHIP_GROUP_OFFSET=250" "off 400:PMHIP_GROUP_OFFSET=400" \
   "3 groups off 160:PMHIP_GROUPS=3,PMHIP_GROUP_OFFSET=160" "3 groups off 330:PMHIP_GROUPS=3,PMHIP_GROUP_OFFSET=330" 2>&1 | grep -v amdgpu.ids | tee "$OUT/groups_25.log"
$P 13 "lockstep start:PMHIP_GROUP_OFFSET=0" "off 50:PMHIP_GROUP_OFFSET=50" "off 100:PMHIP_GROUP_OFFSET=100" "off 160:PMHIP_GROUP_OFFSET=160" "off 250:PMHIP_GROUP_OFFSET=250" "off 400:PMHIP_GROUP_OFFSET=400" \
   "3 groups off 160:PMHIP_GROUPS=3,PMHIP_GROUP_OFFSET=160" "3 groups off 330:PMHIP_GROUPS=3,PMHIP_GROUP_OFFSET=330" "1 group:PMHIP_GROUPS=1" 2>&1 | grep -v amdgpu.ids | tee "$OUT/groups_13.log"
timeout 600 python -m pytest tests/test_gpu_patchmatch.py -m gpu -q -x -k "tuning or config2_full_size" 2>&1 | tail -5 | tee "$OUT/parity_tests.log"
