#!/bin/bash
# Round 3, final GPU call on the frozen tree: the whole -m gpu suite, smoke(), the bench line, rocprofv3 kernel stats of bench.py, memory-side and SQ counters of the
# bench's sweep instantiation (pm_sweep2_kernel<4,2,*>, forced at 24 views with PMHIP_LANES=4) for profiles/traffic.json.
set -u
OUT=gpurun_out/r03_final; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_call.sh r03_final suite
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$OUT/smoke.log"
BENCH_ARGS="--steps 3 --warmup 1" bash tools/gpu_call.sh r03_final bench
BENCH_ARGS="--steps 2 --warmup 1 --no-extras" bash tools/gpu_call.sh r03_final prof
export PMHIP_LANES=4
mkdir -p "$OUT/pmc"
R=$PWD
g++ -std=c++17 -O1 -I"$R/include" "$R/tools/pmc/pmc_workload.cpp" -o /tmp/pmc_workload "$R/openmvs_amd/libpmhip.so" -Wl,-rpath,"$R/openmvs_amd" -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 || exit 1
python tools/pmc/make_scene.py 24 1920 1080 /tmp/pmc_scene.bin > "$OUT/pmc/make_scene.log" 2>&1
export PMHIP_GROUPS=1
( cd /tmp && timeout 120 /tmp/pmc_workload /tmp/pmc_scene.bin 1 > "$R/$OUT/pmc/unprofiled_run.json" 2> "$R/$OUT/pmc/unprofiled.err" ); cat "$OUT/pmc/unprofiled_run.json"
pass() {
  local name=$1; shift
  ( cd /tmp && timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc "$@" --output-format csv -d "/tmp/prof_pmc_$name" -o pmc -- /tmp/pmc_workload /tmp/pmc_scene.bin 1 > "$R/$OUT/pmc/pmc_${name}_run.json" 2> "$R/$OUT/pmc/pmc_$name.err" )
  local rc=$?
  local csv=$(find "/tmp/prof_pmc_$name" -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$csv" ]; then python tools/pmc_agg.py "$csv" > "$OUT/pmc/pmc_${name}_per_kernel.txt" 2>&1; echo "pass $name ok"; else echo "pmc pass $name: rc $rc, no counter csv"; fi
  rm -rf "/tmp/prof_pmc_$name"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
[ -f "$OUT/pmc/pmc_write_per_kernel.txt" ] || PMC_TIMEOUT=300 pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pass sq2 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD
head -12 "$OUT"/pmc/pmc_fetch_per_kernel.txt "$OUT"/pmc/pmc_write_per_kernel.txt 2>/dev/null
