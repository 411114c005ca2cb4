#!/bin/bash
# Round 4, GPU call 17: the init kernel's workgroup size (64 threads = 8 adjacent pixels; 128 / 256 = 16 / 32 adjacent pixels share their source-image lines in one CU's L1)
set -u
OUT=gpurun_out/r04_call17; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 PROBE_STATS=1 PROBE_STEPS=1
for v in b256 b128; do
  PMHIP_LIB=$PWD/openmvs_amd/libpmhip_$v.so timeout 300 python tools/r04/probe_lanes.py 100 "init block $v:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_100.log"
done
