#!/bin/bash
# Round 6, GPU call 4: the tree after the resident-32 / wide-trip experiment was taken out again (pm_sweep2_kernel = call 1's kernel + PMStep geometry, tiled sweeps as their own
# instantiation): A/B against the cuts-only library of call 1, the init pass with optimistic rows, tiled sweeps with pixels numbered densely across tiles.
set -u
OUT=gpurun_out/r06_call4; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
step() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/steps.log"; }
step "A/B at 100 views"
TUNE_STEPS='--steps 5 --warmup 2' timeout 900 python tools/tune.py 100 libpmhip_cuts.so:2 libpmhip.so:2 libpmhip_init2.so:2 libpmhip_cuts.so:2 libpmhip.so:2 libpmhip_init2.so:2 > "$OUT/ab_100.log" 2>&1; cat "$OUT/ab_100.log"
step "A/B at 13 views"
TUNE_STEPS='--steps 8 --warmup 2' timeout 400 python tools/tune.py 13 libpmhip_cuts.so:2 libpmhip.so:2 libpmhip_init2.so:2 > "$OUT/ab_13.log" 2>&1; cat "$OUT/ab_13.log"
step "parity subset (golden config 2 / 5, tiled sweeps, mappings)"
timeout 600 python -m pytest tests -m gpu -q -x -k "config2_full_size or config5 or views_per_lane or tiled or golden" > "$OUT/gpu_subset.log" 2>&1; tail -4 "$OUT/gpu_subset.log"
step "tiled sweeps, pixels numbered densely across tiles: 100 / 13 / 1 views"
timeout 600 python tools/r06/probe_tiles.py 100 100 "exact:groups=2" "t128:tw=128,th=128,groups=2" "t64:tw=64,th=64,groups=2" "t32:tw=32,th=32,groups=2" "t32g1:tw=32,th=32,groups=1" "t16:tw=16,th=16,groups=2" > "$OUT/tiles_100.log" 2>&1; cat "$OUT/tiles_100.log"
PROBE_STEPS=3 timeout 400 python tools/r06/probe_tiles.py 13 13 "exact:groups=2" "t128:tw=128,th=128,groups=2" "t64:tw=64,th=64,groups=2" "t32:tw=32,th=32,groups=2" "t32g1:tw=32,th=32,groups=1" "t16:tw=16,th=16,groups=2" "t16g1:tw=16,th=16,groups=1" > "$OUT/tiles_13.log" 2>&1; cat "$OUT/tiles_13.log"
PROBE_STEPS=3 timeout 400 python tools/r06/probe_tiles.py 9 1 "exact:groups=1" "t64:tw=64,th=64,groups=1" "t32:tw=32,th=32,groups=1" "t16:tw=16,th=16,groups=1" "t8:tw=8,th=8,groups=1" > "$OUT/tiles_1.log" 2>&1; cat "$OUT/tiles_1.log"
step done
