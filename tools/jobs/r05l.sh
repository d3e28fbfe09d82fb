#!/bin/bash
# fast BatchNorm passes: parity (bit-identity vs the general loops, operator tests), micro-benchmark A/B, step A/B
O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests/test_ops16_gpu.py tests/test_ops_gpu.py -k "bn" -m gpu -q -x > $O/bn_tests.log 2>&1; echo "bn tests rc=$?"; tail -3 $O/bn_tests.log
python tools/pointwise_bench16.py bf16 f32 > $O/pointwise_bench16_fast.txt 2>&1; cat $O/pointwise_bench16_fast.txt
EMSA_BN_FAST=0 python tools/pointwise_bench16.py bf16 f32 > $O/pointwise_bench16_general.txt 2>&1; cat $O/pointwise_bench16_general.txt
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
for rep in 1 2; do
run bf16_fast_$rep $A
EMSA_BN_FAST=0 run bf16_general_$rep $A
run f32_fast_$rep $B
EMSA_BN_FAST=0 run f32_general_$rep $B
done
