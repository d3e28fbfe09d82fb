#!/bin/bash
O=gpurun_out/r02af; mkdir -p $O
rm -rf gpurun_out/pmc_traffic
timeout 400 bash tools/pmc_traffic.sh > $O/pmc_f32.log 2>&1; echo "pmc f32 rc=$?"
cp gpurun_out/pmc_traffic/raw.json $O/pmc_raw_f32.json
rm -rf gpurun_out/pmc_traffic
EMSA_PMC_BENCH_ARGS="--dtype bf16" timeout 400 bash tools/pmc_traffic.sh > $O/pmc_bf16.log 2>&1; echo "pmc bf16 rc=$?"
cp gpurun_out/pmc_traffic/raw.json $O/pmc_raw_bf16.json
rm -rf gpurun_out/pmc_traffic
ls -la $O
