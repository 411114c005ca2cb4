#!/bin/bash
# Round 3, GPU call 2: the resident band kernel (pm_band.hip) on the device -- parity first (cross-CU hand-offs cannot be seen by the emulator),
# then timing against the per-diagonal kernels on the same box: 100 views, an 8-GPU shard (13 views), one depth map.
set -u
OUT=gpurun_out/r03_call2; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_patchmatch.py -m gpu -q -x -k "not real_scene" > "$OUT/gpu_patchmatch_band.log" 2>&1; echo "parity rc $?"; tail -5 "$OUT/gpu_patchmatch_band.log"
V="libpmhip.so:2:16:0 libpmhip.so:1:16:1 libpmhip.so:1:4:1 libpmhip_bmw4.so:1:16:1 libpmhip_bmw4.so:1:4:1 libpmhip.so:1:16:1"
VARIANTS="$V" bash tools/gpu_call.sh r03_call2 variants
SMALL_VIEWS=13 SMALL_VARIANTS="libpmhip.so:2:16:0 libpmhip.so:1:16:1 libpmhip_bmw4.so:1:16:1 libpmhip_bmw4.so:1:4:1" bash tools/gpu_call.sh r03_call2 small
timeout 300 python tools/small_batch_probe.py 1 > "$OUT/one_view.log" 2>&1; tail -12 "$OUT/one_view.log"
