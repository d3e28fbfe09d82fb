#!/bin/bash
# PMC traffic only (FETCH_SIZE / WRITE_SIZE passes + calibration) on the current kernel sources -> profiles/r06_pmc_traffic*.json
O=gpurun_out/r06pmc; mkdir -p $O
tools/pmc_traffic2.sh f32 > $O/pmc_f32.log 2>&1; python tools/pmc_traffic_json.py gpurun_out/pmc_f32/raw.json $O/r06_pmc_traffic.json r06 > $O/pmc_f32_json.log 2>&1; tail -3 $O/pmc_f32_json.log
tools/pmc_traffic2.sh bf16 --dtype bf16 > $O/pmc_bf16.log 2>&1; EMSA_PMC_BENCH_ARGS="--dtype bf16" python tools/pmc_traffic_json.py gpurun_out/pmc_bf16/raw.json $O/r06_pmc_traffic_bf16.json r06 > $O/pmc_bf16_json.log 2>&1; tail -3 $O/pmc_bf16_json.log
rm -rf gpurun_out/pmc_f32/FETCH_SIZE gpurun_out/pmc_f32/WRITE_SIZE gpurun_out/pmc_bf16/FETCH_SIZE gpurun_out/pmc_bf16/WRITE_SIZE
cp $O/r06_pmc_traffic.json $O/r06_pmc_traffic_bf16.json profiles/ 2>/dev/null
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_f32.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench_driver_cmd_f32.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'])"
