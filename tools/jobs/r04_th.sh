#!/bin/bash
# twin-launch threshold: batch 3 at 640x480 and batch 1 at 960x736 (R101), twin on / off
O=gpurun_out/r04th; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
EMSA_TWIN=1 run b3_f16_twin1 --eval --graph --batch-size 3 --dtype f16 --steps 200 --warmup 20
EMSA_TWIN=0 run b3_f16_twin0 --eval --graph --batch-size 3 --dtype f16 --steps 200 --warmup 20
EMSA_TWIN=1 run r101_b1_f16_twin1 --eval --graph --batch-size 1 --dtype f16 --backbone resnet101 --height 736 --width 960 --steps 100 --warmup 20
EMSA_TWIN=0 run r101_b1_f16_twin0 --eval --graph --batch-size 1 --dtype f16 --backbone resnet101 --height 736 --width 960 --steps 100 --warmup 20
EMSA_TWIN=1 run r101_b2_f16_twin1 --eval --graph --batch-size 2 --dtype f16 --backbone resnet101 --height 736 --width 960 --steps 100 --warmup 20
EMSA_TWIN=0 run r101_b2_f16_twin0 --eval --graph --batch-size 2 --dtype f16 --backbone resnet101 --height 736 --width 960 --steps 100 --warmup 20
