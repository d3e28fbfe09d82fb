#!/bin/bash
O=gpurun_out/r05ad; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
PY
}
for rep in 1 2 3; do
run c4_f16_$rep --dtype f16 --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline
run c4_bf16_$rep --dtype bf16 --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline
done
run c4_f16_ref --dtype f16 --eval --graph --batch-size 1 --steps 80 --warmup 20 --protocol reference --no-cpu-baseline
