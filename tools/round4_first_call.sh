#!/bin/bash
# First GPU call of round 4 (prepared at the end of round 3, when the budget was spent): what the last change of round 3 -- the two-wide speculative kernel as the
# default for batches of 3-25 reference views -- still lacks on the device.
#   1. the -m gpu suite on that default (it passed under the emulator only; the device has seen the identity check of tools/small_batch_probe.py and the bench's golden leg);
#   2. where the two-wide kernel stops paying: 25 / 35 / 50 / 70 views against one view per lane (PMHIP_DEFAULT_WIDE is 25 because 25 is the largest batch measured);
#   3. the four-waves-per-SIMD build of pm_sweep_widen_kernel (13 views need 3 510 waves for 3 072 slots at three);
#   4. SQ counters of the two-wide kernel at 13 views (tools/pmc/run_pmc_sq.sh with PMHIP_WIDE=25).
set -u
OUT=gpurun_out/r04_call1; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/gpu_call.sh r04_call1 suite
for v in 25 35 50 70; do
  PMHIP_WIDE=64 PMHIP_WIDE_HYPS=2 timeout 300 python tools/tune.py $v libpmhip.so:2 2>&1 | tail -1 | sed 's/^/two-wide   /' | tee -a "$OUT/crossover.log"
  PMHIP_WIDE=0 timeout 300 python tools/tune.py $v libpmhip.so:2:8 2>&1 | tail -1 | sed 's/^/one view per lane /' | tee -a "$OUT/crossover.log"
done
python -c "
from openmvs_amd import build
build.build_variant('libpmhip.so', 'libpmhip_widen4.so', ['-DPM_WIDEN_MINWAVES=4'])"
for lib in libpmhip.so libpmhip_widen4.so; do
  echo "PMHIP_LIB=$lib" | tee -a "$OUT/small.log"
  PMHIP_LIB=$PWD/openmvs_amd/$lib timeout 300 python tools/small_batch_probe.py 4 8 13 2>&1 | grep "WIDE=64" | tee -a "$OUT/small.log"
done
# 5. one stream instead of two view groups for the speculative kernels (a 13-view diagonal is already a single round of the machine)
PMHIP_GROUPS=1 timeout 300 python tools/small_batch_probe.py 8 13 2>&1 | grep "WIDE=64" | sed 's/^/PMHIP_GROUPS=1 /' | tee -a "$OUT/small.log"
PMHIP_WIDE=25 bash tools/pmc/run_pmc_sq.sh "$OUT/pmc_widen2" 13 libpmhip.so 2>&1 | tail -30
