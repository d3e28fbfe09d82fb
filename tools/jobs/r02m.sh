#!/bin/bash
# full GPU suite + bf16 PMC traffic
O=gpurun_out/r02m; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_gpu.log
rm -rf gpurun_out/pmc_traffic
EMSA_PMC_BENCH_ARGS="--dtype bf16" timeout 1500 bash tools/pmc_traffic.sh > $O/pmc_bf16.log 2>&1; echo "pmc rc=$?"
cp gpurun_out/pmc_traffic/raw.json $O/pmc_raw_bf16.json 2>/dev/null
tail -15 $O/pmc_bf16.log
find gpurun_out/pmc_traffic -name "*.csv" -size +2M -delete
