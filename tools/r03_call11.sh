#!/bin/bash
# Round 3, GPU call 11: lane-per-pixel SGM cost kernel (sgm_cost_px_kernel) and 64-bit sum atomics in the uniform path kernel: parity suite, timing
# against the previous kernels (SGMHIP_COST_PX=0, SGMHIP_UNIFORM_ALIGN=2); VALU issue-rate probe (plain vs packed fp32).
set -u
OUT=gpurun_out/r03_call11; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_sgm.py -m gpu -q -x > "$OUT/sgm_suite.log" 2>&1; echo "exit $?" >> "$OUT/sgm_suite.log"; tail -5 "$OUT/sgm_suite.log"
for env in "X=1" "SGMHIP_COST_PX=0" "SGMHIP_UNIFORM_ALIGN=2"; do
  echo "$env" | tee -a "$OUT/sgm_probe.log"
  env $env timeout 300 python tools/probe_sgm.py 2>&1 | grep -v "^W2026" | head -9 | tee -a "$OUT/sgm_probe.log"
done
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-value tools/probes/valu_rate.hip -o /tmp/valu_rate 2>/dev/null && timeout 120 /tmp/valu_rate | tee "$OUT/valu_rate.log"
