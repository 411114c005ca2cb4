#!/bin/bash
# Round 3: the FETCH_SIZE (and first SQ) pass of tools/r03_final.sh again -- rocprofv3 hung in two of the four counter passes of the final call (it does so at random on this pool).
set -u
OUT=gpurun_out/r03_final; mkdir -p "$OUT/pmc"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PMHIP_LANES=4 PMHIP_GROUPS=1
R=$PWD
g++ -std=c++17 -O1 -I"$R/include" "$R/tools/pmc/pmc_workload.cpp" -o /tmp/pmc_workload "$R/openmvs_amd/libpmhip.so" -Wl,-rpath,"$R/openmvs_amd" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 || exit 1
python tools/pmc/make_scene.py 24 1920 1080 /tmp/pmc_scene.bin > /dev/null 2>&1
pass() {
  local name=$1; shift
  ( cd /tmp && timeout ${PMC_TIMEOUT:-120} rocprofv3 --pmc "$@" --output-format csv -d "/tmp/prof_pmc_$name" -o pmc -- /tmp/pmc_workload /tmp/pmc_scene.bin 1 > "$R/$OUT/pmc/pmc_${name}_run.json" 2> "$R/$OUT/pmc/pmc_$name.err" )
  local rc=$?
  local csv=$(find "/tmp/prof_pmc_$name" -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$csv" ]; then python tools/pmc_agg.py "$csv" > "$OUT/pmc/pmc_${name}_per_kernel.txt" 2>&1; echo "pass $name ok"; else echo "pmc pass $name: rc $rc, no counter csv"; fi
  rm -rf "/tmp/prof_pmc_$name"
}
pass fetch FETCH_SIZE
[ -f "$OUT/pmc/pmc_fetch_per_kernel.txt" ] || pass fetch FETCH_SIZE
[ -f "$OUT/pmc/pmc_fetch_per_kernel.txt" ] && pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
head -6 "$OUT"/pmc/pmc_fetch_per_kernel.txt 2>/dev/null; exit 0
