#!/bin/bash
O=gpurun_out/r05s; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'])
except Exception as e:
    print('$name failed', e)
PY
}
A="--dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline"
B="--steps 20 --warmup 5 --no-cpu-baseline"
run base_bf16_1 $A
for v in 512 768 1536 2048; do EMSA_BN_APPLY_WGS=$v run apply_${v}_bf16 $A; done
run base_bf16_2 $A
for v in 256 512 768; do EMSA_BN_REDUCE_ROWS=$v run rows_${v}_bf16 $A; done
run base_bf16_3 $A
run base_f32_1 $B
for v in 512 2048; do EMSA_BN_APPLY_WGS=$v run apply_${v}_f32 $B; done
for v in 512; do EMSA_BN_REDUCE_ROWS=$v run rows_${v}_f32 $B; done
run base_f32_2 $B
