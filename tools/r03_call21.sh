#!/bin/bash
# Round 3, GPU call 21: the mapping / occupancy choice between the 13-view and the 100-view points (25 and 50 views per GPU are what a 4- and 2-GPU split of the
# 100-view job gives a rank): (4,2) and (8,1), 3 and 4 waves per SIMD, two streams.
set -u
OUT=gpurun_out/r03_call21; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
for v in 25 50 70; do
timeout 900 python tools/tune.py $v libpmhip.so:2:4 libpmhip.so:2:8 libpmhip_mw4.so:2:4 libpmhip_mw4.so:2:8 2>&1 | tee -a "$OUT/tune_mid.log"
done
timeout 300 python tools/tune.py 13 libpmhip_mw4.so:2:8 libpmhip.so:2:8 2>&1 | tee -a "$OUT/tune_mid.log"
