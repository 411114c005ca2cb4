#!/bin/bash
# Round 4, GPU call 13: ScoreDepthMapTmp (pm_init_kernel) on anti-diagonals with the sweep's optimistic quad rows instead of row-major pixels with guarded rows
# (before: 402 ms of a 4 235 ms step at 100 views, profiles/r04_final2_bench_kernel_stats.csv).
set -u
OUT=gpurun_out/r04_call13; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_patchmatch.py -m gpu -q -x -k "single_view or geometric_round or config2_full or different_sizes or degenerate or many_source" 2>&1 | tail -3 | tee "$OUT/gpu_subset.log"
P="timeout 400 python tools/r04/probe_lanes.py"
$P 100 "diagonal init:" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_100.log"
$P 25 "diagonal init:" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_25.log"
$P 13 "diagonal init:" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_13.log"
timeout 200 python tools/small_batch_probe.py 1 2>&1 | grep -v amdgpu.ids | tee "$OUT/small.log"
