#!/bin/bash
# Round 5, GPU call 2: (a) A/B on one box: this tree's engine against round 4's (openmvs_amd/libpmhip_r04.so, built from `git show e81a3eb:openmvs_amd/csrc`): whole-pass group streams,
# reference patch and guarded redo from the quad images (no plain anti-diagonal-major copies any more); (b) the default bench line with the new legs (shard_rates,
# scaling_model, exchange / wait split, timed-mix parity, full-schedule tolerance); (c) the gpu suite (without the config-5 golden case, whose file is still being generated).
set -u
OUT=gpurun_out/r05_call2; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
P="timeout 600 python tools/r05/probe_groups.py"
for V in 100 13; do
  $P $V "round 5:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_$V.log"
  PMHIP_LIB=$PWD/openmvs_amd/libpmhip_r04.so $P $V "round 4 library:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_$V.log"
  $P $V "round 5 again:" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/ab_$V.log"
done
timeout 1200 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc $?"; tail -c 3000 "$OUT/bench.json"; tail -3 "$OUT/bench.err"
timeout 1500 python -m pytest tests -m gpu -q -x -k "not config5_resolution" --durations=10 > "$OUT/gpu_suite.log" 2>&1; echo "suite rc $?"; tail -25 "$OUT/gpu_suite.log"
