#!/bin/bash
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops16_gpu.py -x -q -k "se_fusion or pool_se" 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_model16_gpu.py -x -q -k "eval or graph or infer or smoke or train" 2>&1 | tail -3
for i in 1 2 3; do for sp in 1 0; do
  EMSA_SE_PAIR=$sp timeout 600 python bench.py --eval --graph --batch-size 1 --dtype f16 --steps 400 --warmup 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('se_pair=$sp', d['ms_per_step'], d['hipgraph'])"
done; done
