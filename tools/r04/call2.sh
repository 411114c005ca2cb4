#!/bin/bash
# Round 4, GPU call 2: the refactored sweep kernels (branch-free optimistic tap rows, quad images addressed as one buffer by entry index, packed division FMAs,
# no SLP, 4 waves per SIMD) -- gpu suite first (device-only code: inline-asm buffer loads, packed FMAs), then the 100-view / 25 / 13-view timings and occupancy variants.
set -u
OUT=gpurun_out/r04_call2; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > "$OUT/gpu_suite.log" 2>&1; echo "suite exit $?" | tee -a "$OUT/gpu_suite.log"; tail -15 "$OUT/gpu_suite.log"
python -c "
from openmvs_amd import build
build.build_variant('libpmhip.so', 'libpmhip_mw3.so', ['-DPM_BAND_MINWAVES=3'])
build.build_variant('libpmhip.so', 'libpmhip_mw4.so', ['-DPM_BAND_MINWAVES=4'])
build.build_variant('libpmhip.so', 'libpmhip_mw5.so', ['-DPM_BAND_MINWAVES=5'])" 2>&1 | tail -3
for lib in libpmhip_mw3.so libpmhip_mw4.so libpmhip_mw5.so; do
  PMHIP_LIB=$PWD/openmvs_amd/$lib timeout 600 python tools/r04/probe_lanes.py 100 "$lib:" "$lib lanes8:PMHIP_LANES=8" "$lib pointer:PMHIP_QUADBUF=0" "$lib groups3:PMHIP_GROUPS=3" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_100.log"
done
timeout 400 python tools/r04/probe_lanes.py 25 "default_widen2:" "regular:PMHIP_WIDE=0" "regular_lanes4:PMHIP_WIDE=0,PMHIP_LANES=4" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_25.log"
timeout 400 python tools/r04/probe_lanes.py 13 "default_widen2:" "regular:PMHIP_WIDE=0" "widen4:PMHIP_WIDE_HYPS=4" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_13.log"
timeout 300 python tools/r04/probe_lanes.py 1 "default_wide8:" "widen2:PMHIP_WIDE_HYPS=2" "regular:PMHIP_WIDE=0" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_1.log"
