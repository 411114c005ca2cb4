#!/bin/bash
# Round 5, GPU call 3: staggered views (PMHipTuning::viewStagger: view b of a group is `stagger` steps of the schedule behind view b - 1, so a launch mixes anti-diagonals,
# sweeps and levels) at 100 / 50 / 25 / 13 views; and the reference patch from a compact plain anti-diagonal-major image (this tree) against from the quad images
# (openmvs_amd/libpmhip_refq.so = commit c352a63) and against round 4's library.
set -u
OUT=gpurun_out/r05_call3; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PROBE_SPLIT=1
P="timeout 900 python tools/r05/probe_groups.py"
$P 100 "no stagger:" "stagger 5:PMHIP_VIEW_STAGGER=5" "stagger 10:PMHIP_VIEW_STAGGER=10" "stagger 20:PMHIP_VIEW_STAGGER=20" "stagger 30:PMHIP_VIEW_STAGGER=30" "stagger 60:PMHIP_VIEW_STAGGER=60" \
   "stagger 100:PMHIP_VIEW_STAGGER=100" "1 group stagger 15:PMHIP_GROUPS=1,PMHIP_VIEW_STAGGER=15" "1 group stagger 30:PMHIP_GROUPS=1,PMHIP_VIEW_STAGGER=30" \
   "3 groups stagger 45:PMHIP_GROUPS=3,PMHIP_VIEW_STAGGER=45" "stagger 30 sweep2 only:PMHIP_VIEW_STAGGER=30,PMHIP_WIDE_PIXELS=0" 2>&1 | grep -v amdgpu.ids | tee "$OUT/stagger_100.log"
PMHIP_LIB=$PWD/openmvs_amd/libpmhip_refq.so $P 100 "reference patch from quads (c352a63):" 2>&1 | grep -v amdgpu.ids | tee "$OUT/ab_100.log"
PMHIP_LIB=$PWD/openmvs_amd/libpmhip_r04.so $P 100 "round 4 library:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_100.log"
$P 50 "no stagger:" "stagger 20:PMHIP_VIEW_STAGGER=20" "stagger 60:PMHIP_VIEW_STAGGER=60" "stagger 120:PMHIP_VIEW_STAGGER=120" "1 group stagger 30:PMHIP_GROUPS=1,PMHIP_VIEW_STAGGER=30" 2>&1 | grep -v amdgpu.ids | tee "$OUT/stagger_50.log"
$P 25 "no stagger:" "stagger 20:PMHIP_VIEW_STAGGER=20" "stagger 60:PMHIP_VIEW_STAGGER=60" "stagger 120:PMHIP_VIEW_STAGGER=120" "stagger 240:PMHIP_VIEW_STAGGER=240" 2>&1 | grep -v amdgpu.ids | tee "$OUT/stagger_25.log"
$P 13 "no stagger:" "stagger 20:PMHIP_VIEW_STAGGER=20" "stagger 100:PMHIP_VIEW_STAGGER=100" "stagger 400:PMHIP_VIEW_STAGGER=400" 2>&1 | grep -v amdgpu.ids | tee "$OUT/stagger_13.log"
