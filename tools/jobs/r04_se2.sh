#!/bin/bash
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_model16_gpu.py tests/test_graph_gpu.py -x -q -k "eval or graph or infer or postproc" 2>&1 | tail -3
for i in 1 2 3; do
  timeout 600 python bench.py --eval --graph --batch-size 1 --dtype f16 --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['hipgraph'])"
done
