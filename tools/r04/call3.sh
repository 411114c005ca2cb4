#!/bin/bash
# Round 4, GPU call 3: where does a wave-visit's time go now?  In-kernel phase clock (s_memtime) of the refactored sweep kernel at 100 views, the same-neighbour probe
# (an eighth of the source-image footprint: does the memory system matter at all?), and the full default bench line with the new cpu_baseline (reference code) leg.
set -u
OUT=gpurun_out/r04_call3; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "
from openmvs_amd import build
build.build_variant('libpmhip.so', 'libpmhip_prof.so', ['-DPM_PROFILE'])" 2>&1 | tail -2
PMHIP_LIB=$PWD/openmvs_amd/libpmhip_prof.so timeout 300 python tools/phase_prof.py 100 2>&1 | grep -v amdgpu.ids | tee "$OUT/phase_100.log"
PMHIP_LANES=8 PMHIP_LIB=$PWD/openmvs_amd/libpmhip_prof.so timeout 300 python tools/phase_prof.py 100 2>&1 | grep -v amdgpu.ids | tee "$OUT/phase_100_lanes8.log"
PMHIP_WIDE=0 PMHIP_LIB=$PWD/openmvs_amd/libpmhip_prof.so timeout 300 python tools/phase_prof.py 13 2>&1 | grep -v amdgpu.ids | tee "$OUT/phase_13_regular.log"
PROBE_SAME_NB=1 timeout 400 python tools/r04/probe_lanes.py 100 "same_neighbour:" "same_neighbour_lanes8:PMHIP_LANES=8" 2>&1 | grep -v amdgpu.ids | tee "$OUT/same_nb_100.log"
PROBE_SAME_NB=1 timeout 300 python tools/r04/probe_lanes.py 13 "same_neighbour_widen2:" "same_neighbour_regular:PMHIP_WIDE=0" 2>&1 | grep -v amdgpu.ids | tee "$OUT/same_nb_13.log"
timeout 900 python bench.py --steps 3 --warmup 1 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"; tail -c 3000 "$OUT/bench.json"; tail -5 "$OUT/bench.err"
