#!/bin/bash
# instance-head backward with the gradient gather inside the kernel + vectorised head activation: tests, A/B
O=gpurun_out/r05h; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
timeout 900 python -m pytest tests/test_ops16_gpu.py tests/test_ops_gpu.py -k "head_act" -m gpu -q > $O/ops.log 2>&1; echo "op tests rc=$?"; tail -3 $O/ops.log
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py tests/test_boundary_gpu.py -m gpu -q -x > $O/model.log 2>&1; echo "model tests rc=$?"; tail -2 $O/model.log
for rep in 1 2; do
run f32_gather_$rep --steps 20 --warmup 5 --no-cpu-baseline
EMSA_HEAD_GATHER=0 run f32_copy_$rep --steps 20 --warmup 5 --no-cpu-baseline
run bf16_graph_gather_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_HEAD_GATHER=0 run bf16_graph_copy_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
done
