#!/bin/bash
O=gpurun_out/r06k; mkdir -p $O
timeout 900 python -m pytest tests/test_model16_gpu.py -m gpu -q -x -k "bn1_fold16 or nbt1d_block_bf16" > $O/fold.log 2>&1; echo "fold rc=$?"; grep -n "^E  \|passed\|failed\|block train" $O/fold.log | cut -c1-300 | head -30
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e); print(open('$O/$name.err').read()[-1500:])
PY
}
A="--dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing"
for rep in 1 2 3; do
EMSA_BN1_FOLD16=0 run bf16_nofold_$rep $A
EMSA_BN1_FOLD16=1 run bf16_fold_$rep $A
done
