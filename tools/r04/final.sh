#!/bin/bash
# Round 4, final evidence: gpu suite, counters of the timed kernel (pm_sweep2_kernel<4,2>, pinned through the environment for the 24-view counter workload), the default
# bench line, rocprof kernel stats of the same command, SGM kernel stats and counters.  Everything that is to be judged is copied to profiles/ by the caller.
set -u
OUT=gpurun_out/r04_final; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
step() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/steps.log"; }
step "gpu suite"
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > "$OUT/gpu_suite.log" 2>&1; echo "suite exit $?" | tee -a "$OUT/gpu_suite.log"; tail -6 "$OUT/gpu_suite.log"
step "counters of pm_sweep2_kernel<4,2> (24 views, one stream)"
PMHIP_WIDE=0 PMHIP_WIDE_PIXELS=0 PMHIP_LANES=4 PMC_TRIES=2 PMC_TIMEOUT=90 bash tools/pmc/run_pmc.sh "$OUT/pmc" 24 libpmhip.so > "$OUT/pmc.log" 2>&1; tail -30 "$OUT/pmc.log"
python tools/pmc/make_traffic.py "$OUT/pmc" pm_sweep2 > "$OUT/traffic.json" 2> "$OUT/traffic.err" && cp profiles/traffic.json "$OUT/traffic_profiles.json"; tail -3 "$OUT/traffic.err"
step "bench"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"; head -c 400 "$OUT/bench.json"; echo; tail -3 "$OUT/bench.err"
step "rocprof kernel stats of bench.py"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python "$R/bench.py" --no-cpu-baseline --no-extras > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/rocprof.err" ); echo "rc $?"
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \; ; rm -rf /tmp/prof_stats
head -6 "$OUT/bench_kernel_stats.csv"; tail -c 300 "$OUT/bench_under_rocprof.json"; echo
step "sgm"
timeout 600 python tools/probe_sgm.py > "$OUT/sgm_probe.log" 2>&1; cat "$OUT/sgm_probe.log" | tail -12
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sgm -o sgm -- python "$R/tools/probe_sgm.py" > "$R/$OUT/sgm_probe_under_rocprof.log" 2> "$R/$OUT/sgm_rocprof.err" ); echo "rc $?"
find /tmp/prof_sgm -name "*kernel_stats.csv" -exec cp {} "$OUT/sgm_kernel_stats.csv" \; ; rm -rf /tmp/prof_sgm
head -10 "$OUT/sgm_kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_sgm_$c -o pmc -- python "$R/tools/probe_sgm.py" > "$R/$OUT/sgm_pmc_${c}_run.log" 2> "$R/$OUT/sgm_pmc_$c.err" )
  csv=$(find /tmp/prof_sgm_$c -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$csv" ]; then python tools/pmc_agg.py "$csv" > "$OUT/sgm_pmc_${c}_per_kernel.txt" 2>&1; head -20 "$OUT/sgm_pmc_${c}_per_kernel.txt"; else echo "sgm pmc $c: no csv"; fi
  rm -rf /tmp/prof_sgm_$c
done
step done
