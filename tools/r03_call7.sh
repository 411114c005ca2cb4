#!/bin/bash
# Round 3, GPU call 7: the new defaults end to end: bench line (100 views), small batches with the one-wave-per-pixel kernel with and without windows,
# config 5 slice (32 views of 3840x2160 + filters + fuse).
set -u
OUT=gpurun_out/r03_call7; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python tools/small_batch_probe.py 1 2 4 8 13 > "$OUT/small_default.log" 2>&1; cat "$OUT/small_default.log" | grep -v amdgpu.ids
PMHIP_LIB=$PWD/openmvs_amd/libpmhip_wident.so timeout 400 python tools/small_batch_probe.py 1 2 4 8 13 > "$OUT/small_wide_no_tiles.log" 2>&1; grep "WIDE=64" "$OUT/small_wide_no_tiles.log"
timeout 600 python tools/config5_probe.py 32 > "$OUT/config5_32_views_4k.json" 2> "$OUT/config5.err"; tail -c 900 "$OUT/config5_32_views_4k.json"; tail -2 "$OUT/config5.err"
BENCH_ARGS="--steps 2 --warmup 1" bash tools/gpu_call.sh r03_call7 bench
