#!/bin/bash
O=gpurun_out/r06c; mkdir -p $O
timeout 1500 python -m pytest tests/test_timed_size_gpu.py tests/test_sample_pair.py -m gpu -q -s > $O/new_tests.log 2>&1; echo "new tests rc=$?"
grep -n "rel-L2\|^E  \|passed\|failed\|bs-32 vs\|train bs 32\|real sample" $O/new_tests.log | cut -c1-400 | head -50
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
A="--dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline"
run bf16_base_1 $A
for c in 128 160 192 224; do EMSA_RS_CUS=$c run bf16_cus${c} $A; done
run bf16_base_2 $A
EMSA_RS_CUS=128 run bf16_cus128_2 $A
