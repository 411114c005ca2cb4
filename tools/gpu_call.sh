#!/bin/bash
# One gpurun invocation = several measurement steps; pick them by name:
#   gpurun --timeout 1500 -- 'bash tools/gpu_call.sh <tag> suite bench prof pmc variants sgm'
# Everything lands in gpurun_out/<tag>/ (small files only: raw rocprof traces are reduced on the box, the copy-back limit is 64 MiB).
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
step() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/steps.log"; }

pmc_pass() {  # name, counters...  (each pass is its own run: no trace domains next to --pmc; the profiled process imports no torch)
  local name=$1; shift
  [ -f /tmp/pmc_scene.npy ] || python tools/pmc_workload.py gen ${PMC_VIEWS:-100} > "$OUT/pmc_gen.log" 2>&1
  ( cd /tmp && timeout 420 rocprofv3 --pmc "$@" --output-format csv -d "/tmp/prof_pmc_$name" -o pmc -- \
      python "$R/tools/pmc_workload.py" run ${PMC_GEO:-0} > "$R/$OUT/pmc_${name}_run.json" 2> "$R/$OUT/pmc_$name.err" )
  local rc=$?
  local csv=$(find "/tmp/prof_pmc_$name" -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$csv" ]; then python tools/pmc_agg.py "$csv" > "$OUT/pmc_${name}_per_kernel.txt" 2>&1; else echo "pmc pass $name: rc $rc, no counter csv"; tail -5 "$OUT/pmc_$name.err"; fi
  rm -rf "/tmp/prof_pmc_$name"
}

for what in "$@"; do case $what in
suite)
  step "gpu suite"
  timeout 1500 python -m pytest tests -m gpu -q -rxXfE --durations=8 > "$OUT/gpu_suite.log" 2>&1
  echo "exit $?" >> "$OUT/gpu_suite.log"; tail -15 "$OUT/gpu_suite.log" ;;
bench)
  step "bench"
  timeout 900 python bench.py ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "rc $?"; tail -c 1500 "$OUT/bench.json"; tail -3 "$OUT/bench.err" ;;
prof)
  step "rocprof kernel stats of bench.py"
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python "$R/bench.py" --no-cpu-baseline ${BENCH_ARGS:-} \
      > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/rocprof.err" ); echo "rc $?"
  find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \; ; rm -rf /tmp/prof_stats
  head -8 "$OUT/bench_kernel_stats.csv"; tail -c 400 "$OUT/bench_under_rocprof.json" ;;
pmc)
  step "counters"
  pmc_pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
  pmc_pass sq2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD
  pmc_pass fetch FETCH_SIZE
  pmc_pass write WRITE_SIZE
  cat "$OUT"/pmc_*_per_kernel.txt 2>/dev/null | head -80 ;;
variants)
  step "variants"
  timeout 900 python tools/tune.py ${TUNE_VIEWS:-100} ${VARIANTS:-libpmhip.so:2} > "$OUT/variants.log" 2>&1; cat "$OUT/variants.log" ;;
phase)
  step "in-kernel phase profile (-DPM_PROFILE build)"
  for lanes in ${PHASE_LANES:-16}; do for spec in ${PHASE_SPECS:-"100" "9 4"}; do echo "PMHIP_LANES=$lanes" >> "$OUT/phase_prof.log"; PMHIP_LANES=$lanes PMHIP_LIB=$R/openmvs_amd/libpmhip_prof.so timeout 600 python tools/phase_prof.py $spec >> "$OUT/phase_prof.log" 2>&1; done; done
  cat "$OUT/phase_prof.log" ;;
small)
  step "small batches (8-GPU shard size, strong scaling): ${SMALL_VIEWS:-13} views"
  timeout 600 python tools/tune.py ${SMALL_VIEWS:-13} ${SMALL_VARIANTS:-libpmhip.so:1 libpmhip.so:2 libpmhip.so:4 libpmhip.so:13} > "$OUT/small_batches.log" 2>&1; cat "$OUT/small_batches.log" ;;
one)
  step "selected tests: ${ONE_K:-config5}"
  timeout 900 python -m pytest tests -m gpu -q -s -k "${ONE_K:-config5}" > "$OUT/selected_tests.log" 2>&1; tail -12 "$OUT/selected_tests.log" ;;
sgm)
  step "sgm probe"
  timeout 600 python tools/probe_sgm.py > "$OUT/sgm_probe.log" 2>&1; cat "$OUT/sgm_probe.log"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sgm -o sgm -- python "$R/tools/probe_sgm.py" > "$R/$OUT/sgm_probe_under_rocprof.log" 2> "$R/$OUT/sgm_rocprof.err" ); echo "rc $?"
  find /tmp/prof_sgm -name "*kernel_stats.csv" -exec cp {} "$OUT/sgm_kernel_stats.csv" \; ; rm -rf /tmp/prof_sgm
  head -12 "$OUT/sgm_kernel_stats.csv" ;;
*) echo "unknown step $what" ;;
esac; done
step done
