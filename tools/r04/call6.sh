#!/bin/bash
# Round 4, GPU call 6: timing probe -- tap samples from a 16 KB LDS array instead of global memory (garbage values): the upper bound of what staging source windows in LDS could give.
set -u
OUT=gpurun_out/r04_call6; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "
from openmvs_amd import build
build.build_variant('libpmhip.so', 'libpmhip_ldsprobe.so', ['-DPM_PROBE_LDS=1'])" 2>&1 | tail -2
PMHIP_LIB=$PWD/openmvs_amd/libpmhip_ldsprobe.so timeout 600 python tools/r04/probe_lanes.py 100 "lds_probe:" "lds_probe lanes8:PMHIP_LANES=8" "lds_probe groups4:PMHIP_GROUPS=4" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_100.log"
PMHIP_LIB=$PWD/openmvs_amd/libpmhip_ldsprobe.so timeout 300 python tools/r04/probe_lanes.py 25 "lds_probe:" "lds_probe regular:PMHIP_WIDE=0" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_25.log"
PMHIP_LIB=$PWD/openmvs_amd/libpmhip_ldsprobe.so timeout 300 python tools/r04/probe_lanes.py 13 "lds_probe:" "lds_probe regular:PMHIP_WIDE=0" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_13.log"
