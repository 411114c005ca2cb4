#!/bin/bash
# Round 6, GPU call 7: the init pass with two source views per lane (pm_init_kernel<4, GEO, 2, 2>) against one view per lane (call 5's tree).
set -u
OUT=gpurun_out/r06_call7; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
step() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/steps.log"; }
step "A/B at 100 views"
TUNE_STEPS='--steps 5 --warmup 2' timeout 900 python tools/tune.py 100 $LIBS $LIBS > "$OUT/ab_100.log" 2>&1; cat "$OUT/ab_100.log"
step "A/B at 13 views"
TUNE_STEPS='--steps 8 --warmup 2' timeout 400 python tools/tune.py 13 $LIBS > "$OUT/ab_13.log" 2>&1; cat "$OUT/ab_13.log"
step "parity subset"
timeout 600 python -m pytest tests -m gpu -q -x -k "config2_full_size or config5 or views_per_lane or golden or non_default or N8 or N4" > "$OUT/gpu_subset.log" 2>&1; tail -4 "$OUT/gpu_subset.log"
step "rocprof kernel stats, 1 + 1 steps"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-shard-rates --no-tiled-leg > /dev/null 2> "$GRAFT_REPO_ROOT/$OUT/rocprof.err" ); echo "rocprof rc $?"
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \; ; head -8 "$OUT/bench_kernel_stats.csv" | cut -c1-200
step done
