#!/bin/bash
O=gpurun_out/r06j; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/suite.log 2>&1; echo "suite rc=$?"; tail -6 $O/suite.log | cut -c1-300
grep -n "bs-32 vs\|train bs 32\|real sample" $O/suite.log | cut -c1-400
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
run f32_1 --steps 20 --warmup 5 --no-cpu-baseline
run bf16_1 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
