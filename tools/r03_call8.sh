#!/bin/bash
# Round 3, GPU call 8: full gpu suite on the new defaults; SQ counters of the default sweep kernel at 3 and 4 waves per SIMD (48 views: the
# four-lanes-per-pixel mapping); config 5 slice again (speckle filter fix).
set -u
OUT=gpurun_out/r03_call8; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_call.sh r03_call8 suite
rm -f /tmp/pmc_scene.bin
bash tools/pmc/run_pmc_sq.sh "$OUT/pmc_sweep2" 48 libpmhip.so 2>&1 | tail -44
bash tools/pmc/run_pmc_sq.sh "$OUT/pmc_sweep2_mw4" 48 libpmhip_bmw4.so 2>&1 | tail -24
timeout 600 python tools/config5_probe.py 32 > "$OUT/config5_32_views_4k.json" 2> "$OUT/config5.err"; tail -c 700 "$OUT/config5_32_views_4k.json"
