#!/bin/bash
# Round 3, GPU call 18: atomic-free SGM aggregation with staged 16-byte stores and scalar cost addresses (31 VGPRs, 8 waves per SIMD) against the atomic u16 sums (30 VGPRs now).
set -u
OUT=gpurun_out/r03_call18; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_sgm.py tests/test_gpu_sgm_post.py -m gpu -q -x > "$OUT/sgm_suite.log" 2>&1; echo "exit $?" >> "$OUT/sgm_suite.log"; tail -3 "$OUT/sgm_suite.log"
for env in "SGMHIP_DELTA=1" "SGMHIP_DELTA=0"; do
  echo "$env" | tee -a "$OUT/sgm_probe.log"
  env $env timeout 300 python tools/probe_sgm.py 2>&1 | grep -v "^W2026" | head -4 | tee -a "$OUT/sgm_probe.log"
done
timeout 300 python tools/probe_sgm_size.py 2>&1 | grep -v "^W2026" | tee "$OUT/sgm_size.log"
