#!/bin/bash
# Round 6, GPU call 3: what the wide tail trips / resident-32 waves / PMStep geometry cost or buy on the benchmark (same box): cuts-only library of call 1 against this tree
# with and without them; the init pass with optimistic rows; the fat threshold.
set -u
OUT=gpurun_out/r06_call3; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
step() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/steps.log"; }
step "A/B at 100 views"
TUNE_STEPS='--steps 5 --warmup 2' timeout 900 python tools/tune.py 100 libpmhip_cuts.so:2 libpmhip.so:2 libpmhip_nowide.so:2 libpmhip_nofat.so:2 libpmhip_init2.so:2 libpmhip_cuts.so:2 libpmhip.so:2 > "$OUT/ab_100.log" 2>&1; cat "$OUT/ab_100.log"
step "A/B at 13 views"
TUNE_STEPS='--steps 8 --warmup 2' timeout 400 python tools/tune.py 13 libpmhip_cuts.so:2 libpmhip.so:2 libpmhip_nowide.so:2 libpmhip_init2.so:2 > "$OUT/ab_13.log" 2>&1; cat "$OUT/ab_13.log"
step "parity of the variants that change kernels (golden config 2 through the timed mix)"
timeout 300 python -m pytest tests -m gpu -q -x -k "config2_full_size or views_per_lane" > "$OUT/gpu_subset.log" 2>&1; tail -4 "$OUT/gpu_subset.log"
step done
