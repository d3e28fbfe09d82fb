#!/bin/bash
O=gpurun_out/r03v; mkdir -p $O
b() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/$name.json 2>$O/$name.err; python -c "
import json; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['value'], d['ms_per_step'])"; }
for i in 1 2; do
b default$i X=1
b nofold$i EMSA_BN1_FOLD=0
b fold64mb$i EMSA_BN1_FOLD_MIN_MB=64
b foldall$i EMSA_BN1_FOLD=1
done
