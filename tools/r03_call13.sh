#!/bin/bash
# Round 3, GPU call 13: software-pipelined window-less tap rows, second form (rolled loop over row pairs, A / B register sets; no spilled quads):
# 3 waves per SIMD (default build), 2 waves per SIMD without spills, and the row-at-a-time loop; 100 / 13 views and one depth map.
set -u
OUT=gpurun_out/r03_call13; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_patchmatch.py -m gpu -q -x -k "N4 or variants or views_per_lane" > "$OUT/pm_parity.log" 2>&1; echo "exit $?" >> "$OUT/pm_parity.log"; tail -3 "$OUT/pm_parity.log"
timeout 1200 python tools/tune.py 100 libpmhip.so:2 libpmhip_nopipe.so:2 libpmhip_pipe_mw2.so:2 libpmhip.so:2:8 libpmhip_pipe_mw2.so:3 2>&1 | tee "$OUT/tune100.log"
timeout 600 python tools/tune.py 13 libpmhip.so:1 libpmhip_nopipe.so:1 libpmhip_pipe_mw2.so:1 2>&1 | tee "$OUT/tune13.log"
timeout 300 python tools/small_batch_probe.py 1 > "$OUT/small1.log" 2>&1; grep -v amdgpu.ids "$OUT/small1.log" | tail -2
