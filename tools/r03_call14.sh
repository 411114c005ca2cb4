#!/bin/bash
# Round 3, GPU call 14: SGM cost kernel with the products in registers + packed dword stores (default) against the LDS-column build; the RCCL plumbing of
# bench.py with the one rank a single-GPU box allows (every collective call of the N-GPU launch executes); the C++ multi-device host on RcclCollective.
set -u
OUT=gpurun_out/r03_call14; mkdir -p "$OUT"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_gpu_sgm.py -m gpu -q -x > "$OUT/sgm_suite.log" 2>&1; echo "exit $?" >> "$OUT/sgm_suite.log"; tail -3 "$OUT/sgm_suite.log"
for lib in libsgmhip.so libsgmhip_tlds.so; do
  echo "SGMHIP_LIB=$lib" | tee -a "$OUT/sgm_probe.log"
  SGMHIP_LIB=$PWD/openmvs_amd/$lib timeout 300 python tools/probe_sgm.py 2>&1 | grep -v "^W2026" | head -5 | tee -a "$OUT/sgm_probe.log"
done
OPENMVS_AMD_FORCE_COLLECTIVES=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 1 --warmup 0 --views-per-gpu 12 --no-extras --no-cpu-baseline > "$OUT/nccl_one_rank_dry_run.log" 2>&1; echo "rc $?" | tee -a "$OUT/nccl_one_rank_dry_run.log"; tail -c 600 "$OUT/nccl_one_rank_dry_run.log"
timeout 900 python -m pytest tests/test_cpp_dense_multi.py -m gpu -q > "$OUT/cpp_multi.log" 2>&1; echo "exit $?" >> "$OUT/cpp_multi.log"; tail -4 "$OUT/cpp_multi.log"
