#!/bin/bash
# Counter passes over the stand-alone C++ workload, one counter group per pass (no trace domains next to --pmc):
#   bash tools/pmc/run_pmc.sh <out dir> [views] [lib]
# Each pass is bounded by its own timeout; a pass that hangs costs at most PMC_TIMEOUT seconds.
set -u
OUT=$1; V=${2:-24}; LIB=${3:-libpmhip.so}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$OUT"; export TMPDIR=/tmp
g++ -std=c++17 -O1 -I"$R/include" "$R/tools/pmc/pmc_workload.cpp" -o /tmp/pmc_workload "$R/openmvs_amd/$LIB" -Wl,-rpath,"$R/openmvs_amd" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 || exit 1
[ -f /tmp/pmc_scene.bin ] || python "$R/tools/pmc/make_scene.py" "$V" 1920 1080 /tmp/pmc_scene.bin > "$OUT/make_scene.log" 2>&1
export PMHIP_GROUPS=${PMC_GROUPS:-1}
( cd /tmp && timeout 120 /tmp/pmc_workload /tmp/pmc_scene.bin ${PMC_GEO:-1} > "$R/$OUT/unprofiled_run.json" 2> "$R/$OUT/unprofiled.err" ); echo "unprofiled rc $?"; cat "$OUT/unprofiled_run.json"
# rocprofv3 --pmc hangs in roughly every second pass on this pool (at start-up or while writing its output; the workload itself takes seconds), so a pass gets a short
# timeout and up to PMC_TRIES attempts instead of one long wait.
pass() {
  local name=$1; shift
  local try
  for try in $(seq 1 ${PMC_TRIES:-3}); do
    ( cd /tmp && timeout ${PMC_TIMEOUT:-100} rocprofv3 --pmc "$@" --output-format csv -d "/tmp/prof_pmc_$name" -o pmc -- /tmp/pmc_workload /tmp/pmc_scene.bin ${PMC_GEO:-1} \
        > "$R/$OUT/pmc_${name}_run.json" 2> "$R/$OUT/pmc_$name.err" )
    local rc=$?
    local csv=$(find "/tmp/prof_pmc_$name" -name "*counter_collection.csv" 2>/dev/null | head -1)
    if [ -n "$csv" ]; then python "$R/tools/pmc_agg.py" "$csv" > "$OUT/pmc_${name}_per_kernel.txt" 2>&1; echo "pass $name ok (attempt $try)"; rm -rf "/tmp/prof_pmc_$name"; return; fi
    echo "pmc pass $name: attempt $try: rc $rc, no counter csv"; rm -rf "/tmp/prof_pmc_$name"
  done
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pass sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum
cat "$OUT"/pmc_*_per_kernel.txt 2>/dev/null | head -120
