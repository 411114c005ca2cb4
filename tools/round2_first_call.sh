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

step "1 unrun cases"
OPENMVS_AMD_ISOLATED_CHILD=1 timeout 900 python -m pytest tests -m gpu --runxfail -rxXfE -q --timeout=240 \
    -k "sgm_post or golden_cloud or long_invalid or sub_group or single_call_with_ignore_mask or non_default_options or many_source or degenerate or range_limits or option_sweep" > "$OUT/1_unrun_cases.log" 2>&1
echo "exit $?" >> "$OUT/1_unrun_cases.log"; tail -5 "$OUT/1_unrun_cases.log"

step "2 gpu suite"
timeout 1200 python -m pytest tests -m gpu -q -rxXfE > "$OUT/2_gpu_suite.log" 2>&1
echo "exit $?" >> "$OUT/2_gpu_suite.log"; tail -5 "$OUT/2_gpu_suite.log"

step "3 bench"
timeout 600 python bench.py > "$OUT/3_bench.json" 2> "$OUT/3_bench.err"; tail -c 600 "$OUT/3_bench.json"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$OUT/prof_stats" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --no-cpu-baseline \
    > "$GRAFT_REPO_ROOT/$OUT/3_bench_under_rocprof.json" 2> "$GRAFT_REPO_ROOT/$OUT/3_rocprof.err" )

step "4 counters"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv \
    -d "$GRAFT_REPO_ROOT/$OUT/prof_pmc" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --views-per-gpu 16 --no-cpu-baseline \
    > "$GRAFT_REPO_ROOT/$OUT/4_pmc_bench.json" 2> "$GRAFT_REPO_ROOT/$OUT/4_pmc.err" ) || echo "pmc pass failed or timed out" | tee -a "$OUT/steps.log"
PMC_CSV=$(find "$OUT/prof_pmc" -name "*counter_collection.csv" | head -1)
[ -n "$PMC_CSV" ] && python tools/pmc_agg.py "$PMC_CSV" > "$OUT/4_pmc_per_kernel.txt" 2>&1 && rm -f "$PMC_CSV"   # the raw CSV exceeds the copy-back limit

step "5 residency variants"
python - > "$OUT/5_variants_build.log" 2>&1 <<'PY'
from openmvs_amd.build import build_variant
build_variant("libpmhip.so", "libpmhip_wb1.so", ["-DPM_WINBATCH=1"])   # source windows one view per memory round trip (the timed round-1 kernel did that)
build_variant("libpmhip.so", "libpmhip_wb8.so", ["-DPM_WINBATCH=8"])
build_variant("libpmhip.so", "libpmhip_noxcd.so", ["-DPM_XCD_REMAP=0"])  # dispatch-order block mapping (the timed round-1 kernel)
build_variant("libpmhip.so", "libpmhip_tcx10.so", ["-DPM_TCX=10"])        # 13.3 KB LDS per workgroup: 12 instead of 11 workgroups per CU
build_variant("libpmhip.so", "libpmhip_tr16.so", ["-DPM_TR=16"])
build_variant("libpmhip.so", "libpmhip_tr16_w4.so", ["-DPM_TR=16", "-DPM_MINWAVES=4"])
PY
timeout 1500 python tools/tune.py 100 libpmhip.so:2 libpmhip_noxcd.so:2 libpmhip_wb1.so:2 libpmhip_wb8.so:2 libpmhip_tcx10.so:2 libpmhip_tr16.so:2 libpmhip_tr16_w4.so:2 libpmhip_tr16.so:3 > "$OUT/5_variants.log" 2>&1; tail -8 "$OUT/5_variants.log"
step "6 sgm probe"
timeout 600 python tools/probe_sgm.py > "$OUT/6_sgm_probe.log" 2>&1; tail -5 "$OUT/6_sgm_probe.log"
step "done"
