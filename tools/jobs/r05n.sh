#!/bin/bash
# 32-bit indices in the max-pool / SE / up-sampling kernels: operator tests, step A/B (EMSA_IDX32=0 = before)
O=gpurun_out/r05n; mkdir -p $O
echo skip tests
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
A="--dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline"
B="--steps 20 --warmup 5 --no-cpu-baseline"
for rep in 1 2 3; do
run bf16_idx32_$rep $A
EMSA_IDX32=0 run bf16_idx64_$rep $A
run f32_idx32_$rep $B
EMSA_IDX32=0 run f32_idx64_$rep $B
done
