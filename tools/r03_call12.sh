#!/bin/bash
# Round 3, GPU call 12: (a) SGM: lane-per-pixel cost kernel with the w*(v-mean) column in LDS and 3 waves per SIMD; (b) PatchMatch: software-pipelined
# window-less tap rows (PM_ROW_PIPELINE) against the row-at-a-time loop and against a 2-waves-per-SIMD build without spills; parity first.
set -u
OUT=gpurun_out/r03_call12; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_sgm.py -m gpu -q -x > "$OUT/sgm_suite.log" 2>&1; echo "exit $?" >> "$OUT/sgm_suite.log"; tail -3 "$OUT/sgm_suite.log"
timeout 300 python tools/probe_sgm.py 2>&1 | grep -v "^W2026" | head -9 | tee "$OUT/sgm_probe.log"
timeout 900 python -m pytest tests/test_gpu_patchmatch.py -m gpu -q -x -k "N4 or variants or mixed or views_per_lane or non_default" > "$OUT/pm_parity.log" 2>&1; echo "exit $?" >> "$OUT/pm_parity.log"; tail -3 "$OUT/pm_parity.log"
timeout 1200 python tools/tune.py 100 libpmhip.so:2 libpmhip_nopipe.so:2 libpmhip_pipe_mw2.so:2 libpmhip.so:2:8 2>&1 | tee "$OUT/tune100.log"
timeout 600 python tools/tune.py 13 libpmhip.so:1 libpmhip_nopipe.so:1 libpmhip_pipe_mw2.so:1 2>&1 | tee "$OUT/tune13.log"
timeout 300 python tools/small_batch_probe.py 1 > "$OUT/small1.log" 2>&1; grep -v amdgpu.ids "$OUT/small1.log" | tail -4
PMHIP_LIB=$PWD/openmvs_amd/libpmhip_nopipe.so timeout 300 python tools/small_batch_probe.py 1 > "$OUT/small1_nopipe.log" 2>&1; grep -v amdgpu.ids "$OUT/small1_nopipe.log" | tail -4
