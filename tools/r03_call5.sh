#!/bin/bash
# Round 3, GPU call 5: per-diagonal window-less kernel with more view groups; band kernel with deferred publish; 13-view shard with both.
set -u
OUT=gpurun_out/r03_call5; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
V="libpmhip_nt.so:2:4:0 libpmhip_nt.so:3:4:0 libpmhip_nt.so:4:4:0 libpmhip_nt.so:3:16:0 libpmhip.so:1:4:1:256:16 libpmhip.so:1:4:1:384:24 libpmhip.so:1:16:1:256:16"
VARIANTS="$V" bash tools/gpu_call.sh r03_call5 variants
SMALL_VIEWS=13 SMALL_VARIANTS="libpmhip_nt.so:1:16:0 libpmhip_nt.so:2:16:0 libpmhip_nt.so:2:4:0 libpmhip.so:1:16:1:4096:16 libpmhip.so:1:16:1:1024:16" bash tools/gpu_call.sh r03_call5 small
