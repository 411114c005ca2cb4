#!/bin/bash
# Round 3, GPU call 4: band kernel with (band, chunk) tasks: parity on the device, then chunk width x slack sweep at 100 views, and the small batches.
set -u
OUT=gpurun_out/r03_call4; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_patchmatch.py -m gpu -q -x -k "views_per_lane or N8 or geometric or config2 or mask or batch" > "$OUT/gpu_patchmatch_band.log" 2>&1; echo "parity rc $?"; tail -3 "$OUT/gpu_patchmatch_band.log"
V="libpmhip.so:1:4:1:256:16 libpmhip.so:1:4:1:256:64 libpmhip.so:1:4:1:128:32 libpmhip.so:1:4:1:512:32 libpmhip.so:1:4:1:256:8 libpmhip.so:1:16:1:256:16 libpmhip.so:1:16:1:256:64 libpmhip_bmw4.so:1:4:1:256:32 libpmhip_bmw4.so:1:16:1:256:32 libpmhip.so:1:4:1:4096:16"
VARIANTS="$V" bash tools/gpu_call.sh r03_call4 variants
SMALL_VIEWS=13 SMALL_VARIANTS="libpmhip.so:1:16:1:256:16 libpmhip.so:1:16:1:256:64 libpmhip.so:1:4:1:256:16 libpmhip.so:1:16:1:4096:16" bash tools/gpu_call.sh r03_call4 small
timeout 300 python tools/small_batch_probe.py 1 > "$OUT/one_view.log" 2>&1; tail -4 "$OUT/one_view.log"
