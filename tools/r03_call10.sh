#!/bin/bash
# Round 3, GPU call 10: the register-resident SGM path kernel for uniform ranges (sgm_path_uniform_kernel): parity suite of the SGM kernels, then the
# timing probe with it, with 16 / 4 pixels per cost prefetch sub-chunk, and without it (SGMHIP_UNIFORM=0: the general kernel with the new wave minimum).
set -u
OUT=gpurun_out/r03_call10; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_sgm.py -m gpu -q -x > "$OUT/sgm_suite.log" 2>&1; echo "exit $?" >> "$OUT/sgm_suite.log"; tail -5 "$OUT/sgm_suite.log"
for lib in libsgmhip.so libsgmhip_ut16.so libsgmhip_ut4.so; do
  echo "SGMHIP_LIB=$lib" | tee -a "$OUT/sgm_probe.log"
  SGMHIP_LIB=$PWD/openmvs_amd/$lib timeout 300 python tools/probe_sgm.py 2>&1 | grep -v "^W2026" | head -4 | tee -a "$OUT/sgm_probe.log"
done
echo "SGMHIP_UNIFORM=0" | tee -a "$OUT/sgm_probe.log"
SGMHIP_UNIFORM=0 timeout 300 python tools/probe_sgm.py 2>&1 | head -4 | tee -a "$OUT/sgm_probe.log"
