#!/bin/bash
# Round 6, GPU call 1: counters on a SLICE of the benchmark (one level-0 sweep with the benchmark's 100 views resident), the cold-L2 probes, FETCH_SIZE calibration,
# and the A/B of the tap-row instruction cuts.  Everything lands in gpurun_out/r06_call1/.
set -u
OUT=gpurun_out/r06_call1; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-$(pwd)}
START=$(date +%s)
step() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/steps.log"; }
step "counter list"
( cd /tmp && timeout 60 rocprofv3 -L > "$R/$OUT/rocprofv3_counters.txt" 2>&1 )
grep -c . "$OUT/rocprofv3_counters.txt"

step "A/B of the instruction cuts (same box): round-5 library vs this tree, 100 / 13 views"
TUNE_STEPS='--steps 6 --warmup 2' timeout 500 python tools/tune.py 100 libpmhip_r05.so:2 libpmhip.so:2 libpmhip_r05.so:2 libpmhip.so:2 > "$OUT/ab_100.log" 2>&1; cat "$OUT/ab_100.log"
TUNE_STEPS='--steps 10 --warmup 3' timeout 300 python tools/tune.py 13 libpmhip_r05.so:2 libpmhip.so:2 > "$OUT/ab_13.log" 2>&1; cat "$OUT/ab_13.log"

step "golden parity of this tree (config 2 / config 5 golden maps, both kernel families)"
timeout 600 python -m pytest tests -m gpu -q -x -k "golden or config5 or config2 or parity" > "$OUT/gpu_parity_subset.log" 2>&1; tail -5 "$OUT/gpu_parity_subset.log"

step "phase profile + trip histogram (-DPM_PROFILE build), 100 views"
PMHIP_LIB=$R/openmvs_amd/libpmhip_prof.so timeout 400 python tools/phase_prof.py 100 > "$OUT/phase_prof_100.log" 2>&1; cat "$OUT/phase_prof_100.log"

step "slice: build, scene, maps"
for lib in libpmhip.so libpmhip_probes.so libpmhip_inner2.so; do
  g++ -std=c++17 -O1 -I"$R/include" "$R/tools/pmc/pmc_slice.cpp" -o /tmp/slice_${lib%.so} "$R/openmvs_amd/$lib" -ldl -Wl,-rpath,"$R/openmvs_amd" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 || exit 1
done
SL=/tmp/slice_libpmhip
python tools/pmc/make_scene.py 100 1920 1080 /tmp/scene100.bin > "$OUT/make_scene.log" 2>&1
python tools/pmc/make_scene.py 13 1920 1080 /tmp/scene13.bin >> "$OUT/make_scene.log" 2>&1
( cd /tmp && timeout 300 $SL /tmp/scene100.bin prep /tmp/maps100.bin ) > "$OUT/prep100.json" 2> "$OUT/prep100.err"; cat "$OUT/prep100.json"
( cd /tmp && timeout 120 $SL /tmp/scene13.bin prep /tmp/maps13.bin ) > "$OUT/prep13.json" 2> "$OUT/prep13.err"; cat "$OUT/prep13.json"
ls -la /tmp/maps100.bin /tmp/maps13.bin

step "slice unprofiled: 1 / 2 / 3 view groups"
for m in photo geo; do for g in 1 2 3; do ( cd /tmp && timeout 120 $SL /tmp/scene100.bin $m /tmp/maps100.bin $g ) >> "$OUT/slice_unprofiled.jsonl" 2>> "$OUT/slice_unprofiled.err"; done; done
for m in photo geo; do for g in 1 2; do ( cd /tmp && timeout 120 $SL /tmp/scene13.bin $m /tmp/maps13.bin $g ) >> "$OUT/slice_unprofiled.jsonl" 2>> "$OUT/slice_unprofiled.err"; done; done
cat "$OUT/slice_unprofiled.jsonl"

step "warm-cache upper bound: every visit twice inside the kernel (pm_sweep2_kernel only: wide = -1), against once"
for v in 100 13; do for lib in libpmhip libpmhip_inner2; do for m in photo geo; do
  echo -n "$lib views $v: " >> "$OUT/inner_repeat.log"
  ( cd /tmp && timeout 200 /tmp/slice_$lib /tmp/scene$v.bin $m /tmp/maps$v.bin 1 1 -1 ) >> "$OUT/inner_repeat.log" 2>> "$OUT/inner_repeat.err"
done; done; done
cat "$OUT/inner_repeat.log"

pass() {  # name, binary, scene tag, mode, extra args..., then -- counters
  local name=$1 bin=$2 v=$3 mode=$4; shift 4
  local extra=()
  while [ "$1" != "--" ]; do extra+=("$1"); shift; done; shift
  local try
  for try in 1 2; do
    local t0=$(date +%s)
    if [ $(( t0 - START )) -gt ${CALL_BUDGET:-3000} ]; then echo "pass $name skipped: call budget spent" | tee -a "$OUT/steps.log"; return; fi
    ( cd /tmp && timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc "$@" --output-format csv -d "/tmp/prof_$name" -o pmc -- $bin /tmp/scene$v.bin $mode /tmp/maps$v.bin "${extra[@]}" \
        > "$R/$OUT/pmc_${name}_run.json" 2> "$R/$OUT/pmc_$name.err" )
    local rc=$?
    local csv=$(find "/tmp/prof_$name" -name "*counter_collection.csv" 2>/dev/null | head -1)
    if [ -n "$csv" ]; then
      python "$R/tools/pmc_agg.py" "$csv" > "$OUT/pmc_${name}_per_kernel.txt" 2>&1
      python - "$csv" "$OUT/pmc_${name}_dispatches.csv.gz" <<'PY'
import csv, gzip, sys
# per-dispatch rows, reduced: dispatch id, kernel (short), grid size, counter, value
with gzip.open(sys.argv[2], "wt") as o:
    for r in csv.DictReader(open(sys.argv[1])):
        o.write("%s,%s,%s,%s,%s\n" % (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-30:].replace(",", ";"), r.get("Grid_Size", ""), r["Counter_Name"], r["Counter_Value"]))
PY
      [ "${PAIRS:-0}" = 1 ] && python "$R/tools/pmc_agg_alt.py" "$csv" > "$OUT/pmc_${name}_pairs.txt" 2>&1
      echo "pass $name ok (attempt $try, $(( $(date +%s) - t0 )) s)" | tee -a "$OUT/steps.log"; rm -rf "/tmp/prof_$name"; return
    fi
    echo "pmc pass $name: attempt $try: rc $rc, no counter csv ($(( $(date +%s) - t0 )) s)" | tee -a "$OUT/steps.log"; rm -rf "/tmp/prof_$name"
  done
}
step "kernel trace of the slice (per-dispatch durations against grid size): one and two view groups"
for g in 1 2; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_kt$g -o kt -- $SL /tmp/scene100.bin photo /tmp/maps100.bin $g > "$R/$OUT/ktrace_photo_g$g.json" 2> "$R/$OUT/ktrace_g$g.err" )
  csv=$(find /tmp/prof_kt$g -name "*kernel_trace.csv" | head -1)
  [ -n "$csv" ] && python - "$csv" "$OUT/ktrace_photo_g$g.csv.gz" <<'PY'
import csv, gzip, sys
with gzip.open(sys.argv[2], "wt") as o:
    for r in csv.DictReader(open(sys.argv[1])):
        o.write("%s,%s,%s,%s,%s,%s\n" % (r.get("Dispatch_Id", ""), r["Kernel_Name"].split("(")[0][-30:].replace(",", ";"), r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Stream_Id", r.get("Queue_Id", "")), r["Start_Timestamp"], r["End_Timestamp"]))
PY
  rm -rf /tmp/prof_kt$g
done
ls -la "$OUT"/ktrace_*
step "counters on the slice (100 views resident, one level-0 sweep, one view group)"
for m in photo geo; do
  pass sq1_$m $SL 100 $m -- SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
  pass sq2_$m $SL 100 $m -- SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT
  pass tcc_$m $SL 100 $m -- TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
  pass fetch_$m $SL 100 $m -- FETCH_SIZE
  pass write_$m $SL 100 $m -- WRITE_SIZE
  pass tcp_$m $SL 100 $m -- TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
  pass grbm_$m $SL 100 $m -- GRBM_GUI_ACTIVE GRBM_COUNT
done
step "cold-L2 probe: every diagonal launched twice (FETCH_SIZE / TCC hits of the first and of the second launch), 100 and 13 views"
export PAIRS=1
pass fetch2_100 /tmp/slice_libpmhip_probes 100 photo 1 2 -- FETCH_SIZE
pass tcc2_100 /tmp/slice_libpmhip_probes 100 photo 1 2 -- TCC_HIT_sum TCC_MISS_sum
pass fetch2_13 /tmp/slice_libpmhip_probes 13 photo 1 2 -- FETCH_SIZE
pass tcc2_13 /tmp/slice_libpmhip_probes 13 photo 1 2 -- TCC_HIT_sum TCC_MISS_sum
pass fetch2_13_sweep2 /tmp/slice_libpmhip_probes 13 photo 1 2 -1 -- FETCH_SIZE
export PAIRS=0
step "FETCH_SIZE calibration"
( cd /tmp && timeout 120 "$R/tools/probes/_build/fetch_calib" > "$R/$OUT/fetch_calib_unprofiled.log" 2>&1 ); cat "$OUT/fetch_calib_unprofiled.log"
for c in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -c1-8)
  ( cd /tmp && timeout 150 rocprofv3 --pmc $c --output-format csv -d /tmp/prof_calib_$n -o pmc -- "$R/tools/probes/_build/fetch_calib" > "$R/$OUT/fetch_calib_$n.log" 2>&1 )
  csv=$(find /tmp/prof_calib_$n -name "*counter_collection.csv" | head -1); [ -n "$csv" ] && python tools/pmc_agg.py "$csv" > "$OUT/fetch_calib_${n}_per_kernel.txt" 2>&1
  rm -rf /tmp/prof_calib_$n
done
cat "$OUT"/fetch_calib_*_per_kernel.txt
step done
cat "$OUT"/pmc_*_per_kernel.txt 2>/dev/null | head -150
cat "$OUT"/pmc_*_pairs.txt 2>/dev/null | head -60
rm -f /tmp/maps100.bin /tmp/maps13.bin /tmp/scene100.bin /tmp/scene13.bin
