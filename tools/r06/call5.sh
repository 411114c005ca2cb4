#!/bin/bash
# Round 6, GPU call 5: the visit's per-pixel work spread over the pixel's lanes (refinement draws prepared at the head of the visit, one Philox per lane; Dir2Normal's two sincos
# on even / odd lanes; the homography's plane part once per trip; InterpolatePixel's ray coordinates at the head) against the tree of profiles/r06_final; issue rates of the
# instructions the cost model needs (integer multiplies, f64, conversions).
set -u
OUT=gpurun_out/r06_call5; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
step() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/steps.log"; }
step "issue rates"
timeout 120 tools/probes/_build/valu_rate > "$OUT/valu_rate.log" 2>&1; cat "$OUT/valu_rate.log"
step "A/B at 100 views"
TUNE_STEPS='--steps 5 --warmup 2' timeout 900 python tools/tune.py 100 $LIBS $LIBS > "$OUT/ab_100.log" 2>&1; cat "$OUT/ab_100.log"
step "A/B at 13 views"
TUNE_STEPS='--steps 8 --warmup 2' timeout 400 python tools/tune.py 13 $LIBS > "$OUT/ab_13.log" 2>&1; cat "$OUT/ab_13.log"
step "parity subset (golden config 2 / 5, tiled sweeps, mappings)"
timeout 600 python -m pytest tests -m gpu -q -x -k "config2_full_size or config5 or views_per_lane or tiled or golden or non_default" > "$OUT/gpu_subset.log" 2>&1; tail -4 "$OUT/gpu_subset.log"
step done
