#!/bin/bash
# Round 4, GPU call 12: ragged disparity ranges aggregated without atomics (sgm_path_kernel<NK, DELTA>, sgm_path_sub_kernel<LP, MD, DELTA>: one byte store per lane and
# step into the direction's own volume, sums formed by sgm_sum_wta_kernel) against the u16-atomic aggregation (SGMHIP_DELTA=1: the uniform-range kernel only).
set -u
OUT=gpurun_out/r04_call12; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_sgm.py -m gpu -q 2>&1 | tail -3 | tee "$OUT/gpu_sgm_tests.log"
SGMHIP_DELTA=1 timeout 300 python tools/probe_sgm.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/sgm_probe_atomic_ragged.log"
timeout 300 python tools/probe_sgm.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/sgm_probe_delta_ragged.log"
