#!/bin/bash
O=gpurun_out/r05ab2; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('$name', d['value'], d['ms_per_step'], r.get('frac'), r.get('traffic'), (d.get('hipgraph') or {}).get('nodes'), d['config'].get('mode'), (d.get('cpu_baseline') or {}).get('value'))
except Exception as e:
    print('$name failed', e); print(open('$O/$name.err').read()[-1500:])
PY
}
run driver_cmd --gpus 1 --steps 20 --warmup 5
run bf16_default --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
run bf16_eager --dtype bf16 --eager --steps 20 --warmup 5 --no-cpu-baseline
