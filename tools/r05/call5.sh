#!/bin/bash
# Round 5, GPU call 5: (a) SGM Match with the row-pipelined uniform cost kernel (no scratch, 137 VGPRs, 3 waves per SIMD), the same compiled for 4 waves (128 VGPRs, 32 B scratch)
# and round 4's library; (b) the PatchMatch engine of this tree against round 4's library once more (the guarded redo path reads the quad image with 4-byte loads again:
# pm_sweep2_kernel<4,2> back at 127 VGPRs / 4 waves per SIMD); (c) counters on the benchmark's own workload.
set -u
OUT=gpurun_out/r05_call5; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for lib in libsgmhip.so libsgmhip_w4.so libsgmhip_r04.so libsgmhip.so; do
  echo "== $lib" | tee -a "$OUT/sgm_ab.log"
  SGMHIP_LIB=$PWD/openmvs_amd/$lib timeout 300 python tools/r05/probe_sgm_cost.py 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/sgm_ab.log"
done
export PROBE_SPLIT=1
P="timeout 600 python tools/r05/probe_groups.py"
for V in 100 13; do
  $P $V "this tree:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_$V.log"
  PMHIP_LIB=$PWD/openmvs_amd/libpmhip_r04.so $P $V "round 4 library:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_$V.log"
  $P $V "this tree again:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_$V.log"
done
unset PROBE_SPLIT
bash tools/r05/pmc_bench.sh "$OUT/pmc" 2>&1 | tail -160
