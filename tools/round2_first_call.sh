#!/bin/bash
# First GPU call of the next round, in one gpurun invocation (≈ 12-15 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/round2_first_call.sh'
# 1. the 37 cases that never ran on a device, without isolation and without xfail, each under its own timeout;
# 2. the whole -m gpu suite as the driver runs it;
# 3. bench.py (default workload) plain and under rocprofv3 --kernel-trace --stats;
# 4. one SQ_* counter pass on a reduced workload (own run, no trace domains), under a short timeout;
# 5. the residency variants of the sweep kernel;
# 6. SGM timings: Match at 2048x1536 and the whole tSGM loop, resident call vs step-wise.
# Everything lands in gpurun_out/r02_first/; copy what is to be judged into profiles/.
set -u
OUT=gpurun_out/r02_first; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
step() { echo "=== $1 ($(date +%T))" | tee -a "$OUT/steps.log"; }

step "2 gpu suite"
timeout 1200 python -m pytest tests -m gpu -q -rxXfE > "$OUT/2_gpu_suite.log" 2>&1
echo "exit $?" >> "$OUT/2_gpu_suite.log"; tail -5 "$OUT/2_gpu_suite.log"

step "3 bench"
timeout 600 python bench.py > "$OUT/3_bench.json" 2> "$OUT/3_bench.err"; tail -c 600 "$OUT/3_bench.json"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$OUT/prof_stats" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline \
    > "$GRAFT_REPO_ROOT/$OUT/3_bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$OUT/3_rocprof.err" )

step "4 counters"
pmc_pass() {  # name, counters...  (each pass is its own run: no trace domains next to --pmc)
  local name=$1; shift
  ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" --output-format csv -d "$GRAFT_REPO_ROOT/$OUT/prof_pmc_$name" -o pmc -- \
      python "$GRAFT_REPO_ROOT/bench.py" --views-per-gpu 16 --no-cpu-baseline > "$GRAFT_REPO_ROOT/$OUT/4_pmc_${name}_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/4_pmc_$name.err" ) \
      || echo "pmc pass $name failed or timed out" | tee -a "$OUT/steps.log"
  local csv=$(find "$OUT/prof_pmc_$name" -name "*counter_collection.csv" | head -1)
  [ -n "$csv" ] && python tools/pmc_agg.py "$csv" > "$OUT/4_pmc_${name}_per_kernel.txt" 2>&1
  rm -rf "$OUT/prof_pmc_$name"   # the raw CSV exceeds the copy-back limit
}
pmc_pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
pmc_pass sq2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU
pmc_pass fetch FETCH_SIZE
pmc_pass write WRITE_SIZE

step "5 residency variants"
# (variant libraries are cross-compiled in the build container and travel with the snapshot: tools/build_variants.py)
timeout 400 python tools/tune.py 100 libpmhip.so:2 libpmhip_noxcd.so:2 libpmhip_wb8.so:2 libpmhip_tcx10.so:2 libpmhip_tr16.so:2 libpmhip_tr16_w4.so:2 > "$OUT/5_variants.log" 2>&1; tail -8 "$OUT/5_variants.log"
step "6 sgm probe"
timeout 600 python tools/probe_sgm.py > "$OUT/6_sgm_probe.log" 2>&1; tail -5 "$OUT/6_sgm_probe.log"
step "done"
