#!/bin/bash
# Round 5, GPU call 4: on one box -- this tree (whole-pass group streams, reference patch from the compact plain anti-diagonal-major image), the same with the groups meeting before
# every sweep (PMHIP_GROUP_MEET=1: round 4's lockstep), commit c352a63 (reference patch from the quad images) and round 4's library; then the whole gpu suite, which now holds the
# config-5-resolution golden case.
set -u
OUT=gpurun_out/r05_call4; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PROBE_SPLIT=1
P="timeout 600 python tools/r05/probe_groups.py"
for V in 100 13; do
  $P $V "this tree:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_$V.log"
  PMHIP_GROUP_MEET=1 $P $V "this tree, groups meet before every sweep:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_$V.log"
  PMHIP_LIB=$PWD/openmvs_amd/libpmhip_refq.so $P $V "c352a63 (reference patch from quads):" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_$V.log"
  PMHIP_LIB=$PWD/openmvs_amd/libpmhip_r04.so $P $V "round 4 library:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_$V.log"
  $P $V "this tree again:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_$V.log"
  PMHIP_GROUP_MEET=1 $P $V "groups meet, again:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_$V.log"
done
timeout 1700 python -m pytest tests -m gpu -q --durations=10 > "$OUT/gpu_suite.log" 2>&1; echo "suite rc $?"; tail -25 "$OUT/gpu_suite.log"
