#!/bin/bash
# Round 3, GPU call 20: the per-diagonal kernel compiled for 4 waves per SIMD now that the visit state is in LDS (127 / 128 VGPRs, 28-52 B of scratch outside the hot loops):
# one view per lane and two views per lane, 100 / 48 / 24 views; SGM parity + timing of the column-major strip.
set -u
OUT=gpurun_out/r03_call20; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1200 python tools/tune.py 100 libpmhip.so:2 libpmhip_mw4.so:2 libpmhip_mw4.so:2:8 libpmhip.so:2:8 libpmhip_mw4.so:3:8 libpmhip_mw4.so:1:8 2>&1 | tee "$OUT/tune100.log"
timeout 600 python tools/tune.py 48 libpmhip.so:1 libpmhip_mw4.so:1 libpmhip_mw4.so:1:8 libpmhip_mw4.so:2:8 2>&1 | tee "$OUT/tune48.log"
timeout 600 python tools/tune.py 24 libpmhip.so:1 libpmhip_mw4.so:1 2>&1 | tee "$OUT/tune24.log"
timeout 300 python -m pytest tests/test_gpu_sgm.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python tools/probe_sgm.py 2>&1 | grep -v "^W2026" | head -3 | tee "$OUT/sgm_probe.log"
