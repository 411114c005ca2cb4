#!/bin/bash
# Round 6, GPU call 2: the whole gpu suite on the tree with the wide tail trips and the PMStep launch geometry; A/B against the cuts-only library of call 1; the opt-in tiled
# sweeps at 1 / 13 / 100 views; trip histograms (-DPM_PROFILE).
set -u
OUT=gpurun_out/r06_call2; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
step() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/steps.log"; }
step "gpu suite"
timeout 900 python -m pytest tests -m gpu -q -x --durations=6 > "$OUT/gpu_suite.log" 2>&1; echo "suite rc $?"; tail -12 "$OUT/gpu_suite.log"
step "A/B wide tail trips: cuts-only library (call 1) vs this tree"
TUNE_STEPS='--steps 6 --warmup 2' timeout 500 python tools/tune.py 100 libpmhip_cuts.so:2 libpmhip.so:2 libpmhip_cuts.so:2 libpmhip.so:2 > "$OUT/ab_100.log" 2>&1; cat "$OUT/ab_100.log"
TUNE_STEPS='--steps 8 --warmup 2' timeout 300 python tools/tune.py 50 libpmhip_cuts.so:2 libpmhip.so:2 > "$OUT/ab_50.log" 2>&1; cat "$OUT/ab_50.log"
TUNE_STEPS='--steps 10 --warmup 3' timeout 300 python tools/tune.py 13 libpmhip_cuts.so:2 libpmhip.so:2 > "$OUT/ab_13.log" 2>&1; cat "$OUT/ab_13.log"
step "trip histograms (-DPM_PROFILE), 100 views"
PMHIP_LIB=$R/openmvs_amd/libpmhip_prof.so timeout 400 python tools/phase_prof.py 100 > "$OUT/phase_prof_100.log" 2>&1; cat "$OUT/phase_prof_100.log"
step "tiled sweeps: 100 / 13 / 1 views"
timeout 600 python tools/r06/probe_tiles.py 100 100 "exact:groups=2" "t256:tw=256,th=256,groups=2" "t128:tw=128,th=128,groups=2" "t64:tw=64,th=64,groups=2" "t64g1:tw=64,th=64,groups=1" "t32:tw=32,th=32,groups=2" > "$OUT/tiles_100.log" 2>&1; cat "$OUT/tiles_100.log"
PROBE_STEPS=3 timeout 400 python tools/r06/probe_tiles.py 13 13 "exact:groups=2" "t256:tw=256,th=256,groups=2" "t128:tw=128,th=128,groups=2" "t64:tw=64,th=64,groups=2" "t64g1:tw=64,th=64,groups=1" "t32:tw=32,th=32,groups=2" > "$OUT/tiles_13.log" 2>&1; cat "$OUT/tiles_13.log"
PROBE_STEPS=3 timeout 400 python tools/r06/probe_tiles.py 9 1 "exact:groups=1" "t256:tw=256,th=256,groups=1" "t128:tw=128,th=128,groups=1" "t64:tw=64,th=64,groups=1" "t32:tw=32,th=32,groups=1" "t16:tw=16,th=16,groups=1" > "$OUT/tiles_1.log" 2>&1; cat "$OUT/tiles_1.log"
step done
