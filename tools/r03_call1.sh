#!/bin/bash
# Round 3, GPU call 1: counters on the stand-alone C++ workload, then an A/B of the prepared variants and of the window-less ("nt") builds
# over the lane mappings, all on one box.
set -u
OUT=gpurun_out/r03_call1; mkdir -p "$OUT"
bash tools/pmc/run_pmc.sh "$OUT/pmc" 24 libpmhip.so 2>&1 | tee "$OUT/pmc.log"
V="libpmhip.so:2 libpmhip_nt.so:2 libpmhip_nt.so:2:4 libpmhip_nt.so:2:2 libpmhip_nt.so:2:1 libpmhip_ntmw4.so:2 libpmhip_ntmw4.so:2:4 libpmhip_sr0.so:2 libpmhip_gfr.so:2 libpmhip_both.so:2 libpmhip.so:2"
VARIANTS="$V" bash tools/gpu_call.sh r03_call1 variants
