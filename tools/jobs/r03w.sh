#!/bin/bash
O=gpurun_out/r03w; mkdir -p $O
EMSA_DUAL_STREAM=0 timeout 3000 python -m pytest tests -m gpu -x -q > $O/tests_gpu_one_stream.log 2>&1; echo "tests (one stream) rc=$?"; tail -3 $O/tests_gpu_one_stream.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.json 2>$O/driver.err; echo rc=$?; wc -l $O/driver.json; python -c "
import json; d=json.loads(open('$O/driver.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'][:40], d['cpu_baseline']['value'])"
