#!/bin/bash
# Round 4, GPU call 4: two-deep software pipeline over the tap rows (ten 16-byte loads in flight per wave instead of five) -- parity at full size, then timings.
set -u
OUT=gpurun_out/r04_call4; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_patchmatch.py -m gpu -q -x -k "config2_full_size or sweep_kernel_variants or single_view_parity or geometric_round or wide_latency or mixed_resolution" > "$OUT/gpu_subset.log" 2>&1; echo "subset exit $?" | tee -a "$OUT/gpu_subset.log"; tail -5 "$OUT/gpu_subset.log"
timeout 600 python tools/r04/probe_lanes.py 100 "pipelined:" "pipelined lanes8:PMHIP_LANES=8" "pipelined groups3:PMHIP_GROUPS=3" "pipelined groups1:PMHIP_GROUPS=1" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/lanes_100.log"
timeout 400 python tools/r04/probe_lanes.py 25 "default_widen2:" "regular:PMHIP_WIDE=0" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_25.log"
timeout 400 python tools/r04/probe_lanes.py 13 "default_widen2:" "regular:PMHIP_WIDE=0" "widen4:PMHIP_WIDE_HYPS=4" 2>&1 | grep -v amdgpu.ids | tee "$OUT/lanes_13.log"
timeout 300 python tools/small_batch_probe.py 1 2 2>&1 | grep -v amdgpu.ids | tee "$OUT/small.log"
