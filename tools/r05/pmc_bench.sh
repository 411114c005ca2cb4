#!/bin/bash
# Counter passes over the BENCHMARK's own workload: 100 views of 1920x1080, photometric pass + 2 geometric rounds -- the stand-alone C++ program tools/pmc/pmc_workload.cpp
# makes the engine calls of one bench.py step (nothing but the C ABI in the profiled process; rocprofv3 --pmc has hung with torch loaded).  ONE view group here: with two
# groups (three HSA queues) every --pmc pass of round 5's first attempt hung at start-up (profiles/r05_call5_pmc_fetch_timeout.err); counters are per dispatch, so the
# grouping does not change them -- a launch covers all 100 views instead of 50, and PMHIP_WIDE_PIXELS is doubled so that the SAME anti-diagonals (<= 400 pixels long) go to
# the two-wide kernel as in the benchmark's two groups of 50: per step the same kernels run on the same pixels.
# One counter group per pass, no trace domains next to --pmc.   bash tools/r05/pmc_bench.sh <out dir>
set -u
OUT=$1; R=${GRAFT_REPO_ROOT:-$(pwd)}
export PMC_GROUPS=1 PMC_GEO=2 PMC_TIMEOUT=${PMC_TIMEOUT:-90} PMC_TRIES=${PMC_TRIES:-2}
export PMHIP_WIDE_PIXELS=40000
rm -f /tmp/pmc_scene.bin
mkdir -p "$OUT"; export TMPDIR=/tmp
g++ -std=c++17 -O1 -I"$R/include" "$R/tools/pmc/pmc_workload.cpp" -o /tmp/pmc_workload "$R/openmvs_amd/libpmhip.so" -Wl,-rpath,"$R/openmvs_amd" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 || exit 1
python "$R/tools/pmc/make_scene.py" 100 1920 1080 /tmp/pmc_scene.bin > "$OUT/make_scene.log" 2>&1
export PMHIP_GROUPS=$PMC_GROUPS
( cd /tmp && timeout 200 /tmp/pmc_workload /tmp/pmc_scene.bin $PMC_GEO > "$R/$OUT/unprofiled_run.json" 2> "$R/$OUT/unprofiled.err" ); echo "unprofiled rc $?"; cat "$OUT/unprofiled_run.json"
pass() {
  local name=$1; shift
  local try
  for try in $(seq 1 $PMC_TRIES); do
    local t0=$(date +%s)
    ( cd /tmp && timeout $PMC_TIMEOUT rocprofv3 --pmc "$@" --output-format csv -d "/tmp/prof_pmc_$name" -o pmc -- /tmp/pmc_workload /tmp/pmc_scene.bin $PMC_GEO \
        > "$R/$OUT/pmc_${name}_run.json" 2> "$R/$OUT/pmc_$name.err" )
    local rc=$?
    local csv=$(find "/tmp/prof_pmc_$name" -name "*counter_collection.csv" 2>/dev/null | head -1)
    if [ -n "$csv" ]; then python "$R/tools/pmc_agg.py" "$csv" > "$OUT/pmc_${name}_per_kernel.txt" 2>&1; echo "pass $name ok (attempt $try, $(( $(date +%s) - t0 )) s)"; rm -rf "/tmp/prof_pmc_$name"; return; fi
    echo "pmc pass $name: attempt $try: rc $rc, no counter csv ($(( $(date +%s) - t0 )) s)"; rm -rf "/tmp/prof_pmc_$name"
  done
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
PMC_TRIES=1 pass sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD
cat "$OUT"/pmc_*_per_kernel.txt 2>/dev/null | head -150
