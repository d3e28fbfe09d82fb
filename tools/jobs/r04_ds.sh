#!/bin/bash
# batch-1 inference: fp32 graph on one / two streams; 16-bit eager (no graph) with / without twin launches
O=gpurun_out/r04ds; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
EMSA_DUAL_STREAM=1 run b1_f32_graph_ds1 --eval --graph --batch-size 1 --dtype f32 --steps 200 --warmup 20
EMSA_DUAL_STREAM=0 run b1_f32_graph_ds0 --eval --graph --batch-size 1 --dtype f32 --steps 200 --warmup 20
run b1_f16_eager_twin --eval --batch-size 1 --dtype f16 --steps 200 --warmup 20
EMSA_TWIN=0 run b1_f16_eager_twin0 --eval --batch-size 1 --dtype f16 --steps 200 --warmup 20
EMSA_DUAL_STREAM=1 run b2_f32_graph_ds1 --eval --graph --batch-size 2 --dtype f32 --steps 100 --warmup 20
EMSA_DUAL_STREAM=0 run b2_f32_graph_ds0 --eval --graph --batch-size 2 --dtype f32 --steps 100 --warmup 20
