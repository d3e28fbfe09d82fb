#!/bin/bash
O=gpurun_out/r03m; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "dropout or se_fusion or upsample" 2>&1 | tail -3
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_model16_gpu.py tests/test_parallel_gpu.py tests/test_boundary_gpu.py -x -q 2>&1 | grep -v "^  \|^$" | tail -12
for cfg in "f32" "bf16"; do
timeout 900 python bench.py --dtype $cfg --steps 20 --warmup 5 --no-cpu-baseline > $O/$cfg.json 2>$O/$cfg.err; python -c "
import json; d=json.loads(open('$O/$cfg.json').read().strip().splitlines()[-1]); print('$cfg', d['value'], d['ms_per_step'])"; done
timeout 900 python bench.py --losses --steps 20 --warmup 5 --no-cpu-baseline > $O/f32_losses.json 2>$O/f32_losses.err; python -c "
import json; d=json.loads(open('$O/f32_losses.json').read().strip().splitlines()[-1]); print('f32 losses', d['value'], d['ms_per_step'])"
timeout 900 python bench.py --graph --steps 20 --warmup 5 --no-cpu-baseline > $O/f32_graph.json 2>$O/f32_graph.err; python -c "
import json; d=json.loads(open('$O/f32_graph.json').read().strip().splitlines()[-1]); print('f32 graph', d['value'], d['ms_per_step'])"
