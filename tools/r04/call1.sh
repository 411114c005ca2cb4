#!/bin/bash
# Round 4, GPU call 1: does the lane order of pm_sweep2_kernel matter (view-major: the four lanes of a quad read one 64-byte piece of one quad image instead of four images)?
# Plus the over-fetch probe VERDICT r03 item 3 asks for (all sources of a view alias one image: results invalid, time meaningful) and the gpu suite.
set -u
OUT=gpurun_out/r04_call1; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python tools/r04/probe_lanes.py 100 \
  "base:PMHIP_VM=0" "vm:PMHIP_VM=1" "vm_lanes8:PMHIP_VM=1,PMHIP_LANES=8" "lanes8:PMHIP_VM=0,PMHIP_LANES=8" \
  "alias:PMHIP_VM=0,PMHIP_PROBE_ALIAS=1" "alias_vm:PMHIP_VM=1,PMHIP_PROBE_ALIAS=1" \
  "vm_groups1:PMHIP_VM=1,PMHIP_GROUPS=1" "vm_groups4:PMHIP_VM=1,PMHIP_GROUPS=4" "vm_lanes8_groups4:PMHIP_VM=1,PMHIP_LANES=8,PMHIP_GROUPS=4" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_100.log"
timeout 400 python tools/r04/probe_lanes.py 25 \
  "default_widen2:" "regular_vm:PMHIP_WIDE=0,PMHIP_VM=1" "regular:PMHIP_WIDE=0,PMHIP_VM=0" "regular_vm_g1:PMHIP_WIDE=0,PMHIP_VM=1,PMHIP_GROUPS=1" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_25.log"
timeout 400 python tools/r04/probe_lanes.py 13 \
  "default_widen2:" "regular_vm:PMHIP_WIDE=0,PMHIP_VM=1" "regular:PMHIP_WIDE=0,PMHIP_VM=0" "regular_vm_g1:PMHIP_WIDE=0,PMHIP_VM=1,PMHIP_GROUPS=1" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_13.log"
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 > "$OUT/gpu_suite.log" 2>&1; echo "suite exit $?" | tee -a "$OUT/gpu_suite.log"; tail -15 "$OUT/gpu_suite.log"
