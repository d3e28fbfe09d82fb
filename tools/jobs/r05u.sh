#!/bin/bash
# same-box A/B: current library vs the previous commit's (build/prev, EMSA_LIB)
O=gpurun_out/r05u; mkdir -p $O
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
C="--dtype f16 --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline"
for rep in 1 2 3; do
EMSA_LIB=$PWD/build/prev/libemsanet_hip.so run bf16_prev_$rep $A
run bf16_new_$rep $A
EMSA_LIB=$PWD/build/prev/libemsanet_hip.so run f32_prev_$rep $B
run f32_new_$rep $B
done
EMSA_LIB=$PWD/build/prev/libemsanet_hip.so run c4_prev $C
run c4_new $C
EMSA_LIB=$PWD/build/prev/libemsanet_hip.so run c4_prev_2 $C
run c4_new_2 $C
