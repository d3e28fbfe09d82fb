#!/bin/bash
O=gpurun_out/r03r; mkdir -p $O
echo "== all model / parallel / boundary tests with two streams (default)"
timeout 3000 python -m pytest tests/test_model_gpu.py tests/test_model16_gpu.py tests/test_parallel_gpu.py tests/test_boundary_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | grep -v "^  \|^$" | tail -8
for dt in bf16 f32; do for d in 0 1 0 1; do
  EMSA_DUAL_STREAM=$d timeout 900 python bench.py --dtype $dt --steps 20 --warmup 5 --no-cpu-baseline > $O/${dt}_dual$d.json 2>$O/${dt}_dual$d.err; python -c "
import json; d=json.loads(open('$O/${dt}_dual$d.json').read().strip().splitlines()[-1]); print('$dt dual=$d', d['value'], d['ms_per_step'], d['peak_hbm_gib'])"
done; done
timeout 900 python bench.py --graph --steps 20 --warmup 5 --no-cpu-baseline > $O/f32_graph.json 2>$O/f32_graph.err; python -c "
import json; d=json.loads(open('$O/f32_graph.json').read().strip().splitlines()[-1]); print('f32 graph', d['value'], d['ms_per_step'])"
timeout 900 python bench.py --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline > $O/bf16_graph.json 2>$O/bf16_graph.err; python -c "
import json; d=json.loads(open('$O/bf16_graph.json').read().strip().splitlines()[-1]); print('bf16 graph', d['value'], d['ms_per_step'])"
