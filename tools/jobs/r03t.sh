#!/bin/bash
O=gpurun_out/r03t; mkdir -p $O
EMSA_WGRAD_STREAM=1 timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "pinned_gradients_small or nbt1d_block" 2>&1 | tail -3
for dt in f32 bf16; do for w in 0 1 0 1; do
  EMSA_WGRAD_STREAM=$w timeout 900 python bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/${dt}_w$w.json 2>$O/${dt}_w$w.err; python -c "
import json; d=json.loads(open('$O/${dt}_w$w.json').read().strip().splitlines()[-1]); print('$dt wgrad-stream=$w', d['value'], d['ms_per_step'], d['peak_hbm_gib'])"
done; done
