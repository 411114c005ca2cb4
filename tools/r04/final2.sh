#!/bin/bash
# Round 4, second evidence run (after the per-launch kernel choice became the default and views of different sizes went in): gpu suite, default bench line, rocprof kernel
# stats of the same command.  The counters of pm_sweep2_kernel<4,2> (profiles/traffic.json, r04_final_pmc/) and the SGM evidence are those of tools/r04/final.sh: the kernel
# sources they were measured on are unchanged (traffic.json carries their digest and bench.py checks it).
set -u
OUT=gpurun_out/r04_final2; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
step() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/steps.log"; }
step "gpu suite"
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > "$OUT/gpu_suite.log" 2>&1; echo "suite exit $?" | tee -a "$OUT/gpu_suite.log"; tail -6 "$OUT/gpu_suite.log"
step "smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee "$OUT/smoke.log"
step "bench"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"; head -c 300 "$OUT/bench.json"; echo; tail -3 "$OUT/bench.err"
step "rocprof kernel stats of bench.py"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python "$R/bench.py" --no-cpu-baseline --no-extras > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/rocprof.err" ); echo "rc $?"
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \; ; rm -rf /tmp/prof_stats
head -6 "$OUT/bench_kernel_stats.csv"; tail -c 300 "$OUT/bench_under_rocprof.json"; echo
step "view groups once more, with the per-launch choice"
timeout 300 python tools/r04/probe_lanes.py 100 "default:" "groups3:PMHIP_GROUPS=3" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_100.log"
timeout 200 python tools/r04/probe_lanes.py 13 "default:" "groups3:PMHIP_GROUPS=3" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_13.log"
timeout 200 python tools/small_batch_probe.py 1 2 2>&1 | grep -v amdgpu.ids | tee "$OUT/small.log"
step done
