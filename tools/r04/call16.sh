#!/bin/bash
# Round 4, GPU call 16: pm_init_kernel with optimistic tap rows from the level's row-major images (one buffer, two dwordx2 loads per sample) instead of the guarded rows
# (before: 400 ms of init kernels per 100-view step, 108 ms of it not tap rows: profiles/r04_call15_*).
set -u
OUT=gpurun_out/r04_call16; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PROBE_STATS=1
timeout 600 python -m pytest tests/test_gpu_patchmatch.py -m gpu -q -x -k "single_view or geometric_round or config2_full or different_sizes or degenerate or many_source or own_size or non_divisible" 2>&1 | tail -3 | tee "$OUT/gpu_subset.log"
timeout 400 python tools/r04/probe_lanes.py 100 "optimistic init rows:" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_100.log"
timeout 200 python tools/r04/probe_lanes.py 13 "optimistic init rows:" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_13.log"
