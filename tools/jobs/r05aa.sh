#!/bin/bash
O=gpurun_out/r05aa; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'), d['config'].get('mode'))
except Exception as e:
    print('$name failed', e)
PY
}
B="--steps 20 --warmup 5 --no-cpu-baseline"
for rep in 1 2 3; do
run f32_eager_$rep $B
run f32_graph_$rep $B --graph
done
