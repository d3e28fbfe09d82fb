#!/bin/bash
O=gpurun_out/r03u; mkdir -p $O
EMSA_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --batch-size 4 --steps 6 --warmup 2 --no-cpu-baseline > $O/gloo2.json 2> $O/gloo2.err; echo rc=$?; python -c "
import json; d=json.loads(open('$O/gloo2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['n_gpus'], d['roofline']['frac'], d['roofline'].get('measured_over')); print(d['comm'])" || tail -5 $O/gloo2.err
timeout 900 python -m pytest tests/test_parallel_gpu.py -x -q 2>&1 | tail -3
timeout 600 python bench.py --eval --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import sys, json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eval', d['value'], d['roofline']['frac'], d['roofline'].get('measured_over'))"
