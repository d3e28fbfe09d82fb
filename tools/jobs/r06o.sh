#!/bin/bash
# defaults re-checked on the final sources: fold thresholds (fp32 / bf16), one box, two rounds
O=gpurun_out/r06o; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
B="--dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing"
F="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing"
for rep in 1 2; do
run bf16_default_$rep $B
EMSA_BN1_FOLD16_MIN_MB=12 run bf16_fold16_min12_$rep $B
EMSA_BN1_FOLD16_MIN_MB=48 run bf16_fold16_min48_$rep $B
run f32_default_$rep $F
EMSA_BN1_FOLD_MIN_MB=12 run f32_fold_min12_$rep $F
EMSA_BN1_FOLD_MIN_MB=48 run f32_fold_min48_$rep $F
EMSA_BN1_FOLD=0 run f32_fold_off_$rep $F
done
