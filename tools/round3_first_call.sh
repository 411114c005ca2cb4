#!/bin/bash
# First GPU call of round 3 (about 5 GPU-minutes):  gpurun --timeout 900 -- 'bash tools/round3_first_call.sh'
# Variants are cross-compiled beforehand in the build container and travel with the snapshot:
#   python tools/build_variants.py sr0=-DPM_SMOOTH_IN_ROW0=1 gfr=-DPM_GLOBAL_FAST_ROW=1 both=-DPM_SMOOTH_IN_ROW0=1,-DPM_GLOBAL_FAST_ROW=1 mw2=-DPM_MINWAVES=2 pNS=-DPM_PROBE_NO_SMOOTH pFB=-DPM_PROBE_NO_FALLBACK
# 1. A/B on ONE box (box-to-box spread is +-3 %, larger than most effects of round 2): product, the two variants prepared in round 2 (bit-exact under the
#    emulator, never timed: smoothness chain next to the first tap row; failed tap rows retried with the optimistic code on the image) and both together,
#    and the two probes that bound what they can gain; the candidates twice, interleaved.
# 2. The product bench line with all legs, and rocprof kernel stats of the plain benchmark.
set -u
OUT=gpurun_out/r03_first; mkdir -p "$OUT"
V="libpmhip.so:2 libpmhip_sr0.so:2 libpmhip_gfr.so:2 libpmhip_both.so:2 libpmhip.so:2 libpmhip_sr0.so:2 libpmhip_gfr.so:2 libpmhip_both.so:2 libpmhip_mw2.so:2 libpmhip_pNS.so:2 libpmhip_pFB.so:2"   # mw2: 187 VGPRs, no scratch, 8 instead of 11 waves per CU
VARIANTS="$V" bash tools/gpu_call.sh r03_first variants
BENCH_ARGS="--steps 2 --warmup 1" bash tools/gpu_call.sh r03_first bench
BENCH_ARGS="--no-extras" bash tools/gpu_call.sh r03_first prof
