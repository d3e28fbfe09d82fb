#!/bin/bash
# HBM traffic per launch from PMC counters (separate FETCH_SIZE / WRITE_SIZE passes), fp32 and bf16
export EMSA_DUAL_STREAM=0
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic_f32.log 2>&1; cp gpurun_out/pmc_traffic/raw.json gpurun_out/pmc_raw_f32.json; tail -15 gpurun_out/pmc_traffic_f32.log
rm -rf gpurun_out/pmc_traffic
EMSA_PMC_BENCH_ARGS="--dtype bf16" bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic_bf16.log 2>&1; cp gpurun_out/pmc_traffic/raw.json gpurun_out/pmc_raw_bf16.json; tail -8 gpurun_out/pmc_traffic_bf16.log
rm -rf gpurun_out/pmc_traffic
