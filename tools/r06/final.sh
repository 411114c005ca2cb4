#!/bin/bash
# Round 6, evidence run on the final tree (one gpurun call): counters over a slice of the benchmark -> profiles/traffic.json (written before bench.py runs, so that the bench line
# carries this tree's traffic and VALU figures), the whole gpu suite, the default bench line, rocprofv3 kernel stats of bench.py, SGM evidence, the N-rank path on one GPU.
set -u
OUT=gpurun_out/r06_final; mkdir -p "$OUT/pmc"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
START=$(date +%s)
step() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/steps.log"; }
step "slice: build, scene, maps"
g++ -std=c++17 -O1 -I"$R/include" "$R/tools/pmc/pmc_slice.cpp" -o /tmp/slice "$R/openmvs_amd/libpmhip.so" -ldl -Wl,-rpath,"$R/openmvs_amd" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 || exit 1
python tools/pmc/make_scene.py 100 1920 1080 /tmp/scene100.bin > "$OUT/pmc/make_scene.log" 2>&1
( cd /tmp && timeout 300 /tmp/slice /tmp/scene100.bin prep /tmp/maps100.bin ) > "$OUT/pmc/prep100.json" 2> "$OUT/pmc/prep100.err"; cat "$OUT/pmc/prep100.json"
for m in photo geo; do for g in 1 2; do ( cd /tmp && timeout 120 /tmp/slice /tmp/scene100.bin $m /tmp/maps100.bin $g ) >> "$OUT/pmc/slice_unprofiled.jsonl" 2>> "$OUT/pmc/slice_unprofiled.err"; done; done
cat "$OUT/pmc/slice_unprofiled.jsonl"
pass() {  # name mode -- counters
  local name=$1 mode=$2; shift 3
  local try
  for try in 1 2; do
    local t0=$(date +%s)
    ( cd /tmp && timeout 150 rocprofv3 --pmc "$@" --output-format csv -d "/tmp/prof_$name" -o pmc -- /tmp/slice /tmp/scene100.bin $mode /tmp/maps100.bin > "$R/$OUT/pmc/pmc_${name}_run.json" 2> "$R/$OUT/pmc/pmc_$name.err" )
    local rc=$?
    local csv=$(find "/tmp/prof_$name" -name "*counter_collection.csv" 2>/dev/null | head -1)
    if [ -n "$csv" ]; then python "$R/tools/pmc_agg.py" "$csv" > "$OUT/pmc/pmc_${name}_per_kernel.txt" 2>&1; echo "pass $name ok (attempt $try, $(( $(date +%s) - t0 )) s)" | tee -a "$OUT/steps.log"; rm -rf "/tmp/prof_$name"; return; fi
    echo "pmc pass $name: attempt $try: rc $rc, no counter csv ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/steps.log"; rm -rf "/tmp/prof_$name"
  done
}
step "counters on the slice"
for m in photo geo; do
  pass sq1_$m $m -- SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  pass sq2_$m $m -- SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT
  pass tcc_$m $m -- TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  pass fetch_$m $m -- FETCH_SIZE
  pass write_$m $m -- WRITE_SIZE
  pass tcp_$m $m -- TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
  pass grbm_$m $m -- GRBM_GUI_ACTIVE GRBM_COUNT
  # (the texture-address unit's own counters -- TA_TA_BUSY_sum, TA_TOTAL_WAVEFRONTS_sum, TA_BUFFER_TOTAL_CYCLES_sum, TA_*_STALLED_BY_TC_CYCLES_sum, TD_TD_BUSY_sum -- hang rocprofv3 on
  # this workload: both attempts of all four passes ran into their 150 s limit, profiles/r06_final/steps.log of 08:05-08:26; the unit's load is therefore priced from
  # SQ_INSTS_VMEM_RD and the probe's cost per wave-load, tools/r06/make_traffic.py)
done
rm -f /tmp/maps100.bin /tmp/scene100.bin
python tools/r06/make_traffic.py "$OUT/pmc" > "$OUT/pmc/make_traffic.log" 2>&1; tail -45 "$OUT/pmc/make_traffic.log"; cp profiles/traffic.json "$OUT/traffic.json"
step "gpu suite"
timeout 900 python -m pytest tests -m gpu -q --durations=6 > "$OUT/gpu_suite.log" 2>&1; echo "suite rc $?"; tail -12 "$OUT/gpu_suite.log"
step "bench"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"; tail -c 800 "$OUT/bench.json"; tail -2 "$OUT/bench.err"
step "rocprof kernel stats of bench.py"
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python "$R/bench.py" --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-shard-rates --no-tiled-leg \
    > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/rocprof.err" ); echo "rocprof rc $?"
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_kernel_stats.csv" \; ; rm -rf /tmp/prof_stats
head -8 "$OUT/bench_kernel_stats.csv"; tail -c 300 "$OUT/bench_under_rocprof.json"
step "sgm: probe, rocprof kernel stats, FETCH / WRITE"
timeout 300 python tools/probe_sgm.py > "$OUT/sgm_probe.log" 2>&1; tail -12 "$OUT/sgm_probe.log"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sgm -o sgm -- python "$R/tools/probe_sgm.py" > "$R/$OUT/sgm_probe_under_rocprof.log" 2> "$R/$OUT/sgm_rocprof.err" ); echo "rc $?"
find /tmp/prof_sgm -name "*kernel_stats.csv" -exec cp {} "$OUT/sgm_kernel_stats.csv" \; ; rm -rf /tmp/prof_sgm; head -10 "$OUT/sgm_kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  if [ $(( $(date +%s) - START )) -gt 2700 ]; then echo "sgm pmc $c skipped: call budget"; continue; fi
  ( cd /tmp && timeout 150 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_sgm_$c -o pmc -- python "$R/tools/probe_sgm.py" > "$R/$OUT/sgm_pmc_${c}_run.log" 2> "$R/$OUT/sgm_pmc_$c.err" )
  csv=$(find /tmp/prof_sgm_$c -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$csv" ]; then python tools/pmc_agg.py "$csv" > "$OUT/sgm_pmc_${c}_per_kernel.txt" 2>&1; head -16 "$OUT/sgm_pmc_${c}_per_kernel.txt"; else echo "sgm pmc $c: no csv"; fi
  rm -rf /tmp/prof_sgm_$c
done
step "N-rank path on one GPU"
sed -e 's#gpurun_out/r05_ranks_on_one_gpu#gpurun_out/r06_final/ranks#g' tools/r05/ranks_on_one_gpu.sh > /tmp/ranks.sh; timeout 400 bash /tmp/ranks.sh 2>&1 | tail -8
step done
