#!/bin/bash
O=gpurun_out/r03q; mkdir -p $O
echo "== model tests with the two-stream encoder"
EMSA_DUAL_STREAM=1 timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_model16_gpu.py -x -q -k "pinned_gradients_small or full_model_small or hipgraph_train_step_matches or train_bf16 or bf16_training_step or nbt1d_block_bf16" 2>&1 | grep -v "^  \|^$" | tail -6
for dt in bf16 f32; do for d in 0 1 0 1; do
  EMSA_DUAL_STREAM=$d timeout 900 python bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline > $O/${dt}_dual$d.json 2>$O/${dt}_dual$d.err; python -c "
import json; d=json.loads(open('$O/${dt}_dual$d.json').read().strip().splitlines()[-1]); print('$dt dual=$d', d['value'], d['ms_per_step'])"
done; done
