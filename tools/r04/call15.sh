#!/bin/bash
# Round 4, GPU call 15: how much of pm_init_kernel (ScoreDepthMapTmp, 9.5 % of a 100-view step) are its guarded tap rows?  Timing probe: a build whose init kernel skips them
# (results invalid; own process: the library is loaded once).  Decides whether optimistic row-major rows for the init kernel are worth building.
set -u
OUT=gpurun_out/r04_call15; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PROBE_STATS=1 PROBE_STEPS=1
PMHIP_LIB=$PWD/openmvs_amd/libpmhip_initnotaps.so timeout 400 python tools/r04/probe_lanes.py 100 "init without taps:" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_100_notaps.log"
