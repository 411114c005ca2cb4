#!/bin/bash
# Copies what tools/r06/final.sh left under gpurun_out/r06_final into profiles/ (tracked): counter passes -> profiles/r06_final_pmc + profiles/traffic.json, the rest -> profiles/r06_final.
set -eu
cd "$(dirname "$0")/../.."
S=gpurun_out/r06_final
mkdir -p profiles/r06_final_pmc profiles/r06_final
rm -f profiles/r06_final_pmc/* 
cp $S/pmc/*_per_kernel.txt $S/pmc/*.json $S/pmc/*.jsonl $S/pmc/make_traffic.log profiles/r06_final_pmc/
cp $S/traffic.json profiles/traffic.json
cp $S/steps.log profiles/r06_final/steps.log
cp $S/bench.json $S/bench_kernel_stats.csv $S/bench_under_rocprof.json $S/sgm_kernel_stats.csv $S/sgm_pmc_*_per_kernel.txt $S/sgm_probe.log profiles/r06_final/
tail -15 $S/gpu_suite.log > profiles/r06_final/gpu_suite_tail.log
cp $S/ranks/*.json profiles/r06_final/
