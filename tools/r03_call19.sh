#!/bin/bash
# Round 3, GPU call 19: uniform-range cost kernel with the right-image strip in LDS (sgm_cost_uni_kernel, 3 waves per SIMD, no loads in the walk) against the
# sliding-window kernel; the tSGM sub-group Match with the lane-per-pixel cost kernel.
set -u
OUT=gpurun_out/r03_call19; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_sgm.py tests/test_gpu_sgm_post.py -m gpu -q -x > "$OUT/sgm_suite.log" 2>&1; echo "exit $?" >> "$OUT/sgm_suite.log"; tail -3 "$OUT/sgm_suite.log"
for env in "SGMHIP_COST_UNI=1" "SGMHIP_COST_UNI=0"; do
  echo "$env" | tee -a "$OUT/sgm_probe.log"
  env $env timeout 300 python tools/probe_sgm.py 2>&1 | grep -v "^W2026" | head -9 | tee -a "$OUT/sgm_probe.log"
done
