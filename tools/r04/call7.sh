#!/bin/bash
# Round 4, GPU call 7: where does the two-wide speculative kernel stop paying?  (PMHIP_WIDE = largest batch that uses it; 25 was simply the largest batch measured in round 3)
set -u
OUT=gpurun_out/r04_call7; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python tools/r04/probe_lanes.py 100 "regular_4_2:" "widen2:PMHIP_WIDE=100" "widen2 groups4:PMHIP_WIDE=100,PMHIP_GROUPS=4" "widen4:PMHIP_WIDE=100,PMHIP_WIDE_HYPS=4" "regular_4_2 groups2 again:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_100.log"
timeout 400 python tools/r04/probe_lanes.py 50 "regular:PMHIP_WIDE=0" "regular lanes4:PMHIP_WIDE=0,PMHIP_LANES=4" "widen2:PMHIP_WIDE=50" "widen2 groups4:PMHIP_WIDE=50,PMHIP_GROUPS=4" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_50.log"
timeout 300 python tools/r04/probe_lanes.py 25 "widen2:" "widen2 groups1:PMHIP_GROUPS=1" "widen2 groups4:PMHIP_GROUPS=4" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_25.log"
timeout 300 python tools/r04/probe_lanes.py 13 "widen2:" "widen2 groups1:PMHIP_GROUPS=1" "widen2 groups4:PMHIP_GROUPS=4" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_13.log"
