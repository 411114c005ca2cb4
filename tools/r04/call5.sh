#!/bin/bash
# Round 4, GPU call 5: three loads per sample from the plain anti-diagonal-major image (a quarter of the quad image's footprint) against one 16-byte load from the quad image.
set -u
OUT=gpurun_out/r04_call5; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "
from openmvs_amd import build
build.build_variant('libpmhip.so', 'libpmhip_tri.so', ['-DPM_TRILOAD=1'])" 2>&1 | tail -2
PMHIP_LIB=$PWD/openmvs_amd/libpmhip_tri.so timeout 600 python -m pytest tests/test_gpu_patchmatch.py -m gpu -q -x -k "config2_full_size or single_view_parity or geometric_round" > "$OUT/gpu_subset_tri.log" 2>&1; echo "subset exit $?" | tee -a "$OUT/gpu_subset_tri.log"; tail -3 "$OUT/gpu_subset_tri.log"
for lib in libpmhip.so libpmhip_tri.so; do
  PMHIP_LIB=$PWD/openmvs_amd/$lib timeout 600 python tools/r04/probe_lanes.py 100 "$lib:" "$lib lanes8:PMHIP_LANES=8" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_100.log"
  PMHIP_LIB=$PWD/openmvs_amd/$lib timeout 300 python tools/r04/probe_lanes.py 25 "$lib:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_25.log"
  PMHIP_LIB=$PWD/openmvs_amd/$lib timeout 300 python tools/r04/probe_lanes.py 13 "$lib:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_13.log"
done
PMHIP_LIB=$PWD/openmvs_amd/libpmhip_tri.so timeout 300 python tools/small_batch_probe.py 1 2>&1 | grep -v amdgpu.ids | tee "$OUT/small_tri.log"
