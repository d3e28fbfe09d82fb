#!/bin/bash
# 16-bit stem on 32-channel K steps: tests + A/B
O=gpurun_out/r05i; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
timeout 900 python -m pytest tests/test_ops16_gpu.py -m gpu -q -x > $O/ops16.log 2>&1; echo "ops16 rc=$?"; tail -2 $O/ops16.log
timeout 900 python -m pytest tests/test_model16_gpu.py -k "eval_16bit or hipgraph or twin" -m gpu -q -x > $O/m16.log 2>&1; echo "m16 rc=$?"; tail -2 $O/m16.log
for rep in 1 2; do
run bf16_graph_stem32_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_CONVH_STEM32=0 run bf16_graph_stem64_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
done
run c4_f16_stem32 --dtype f16 --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline
EMSA_CONVH_STEM32=0 run c4_f16_stem64 --dtype f16 --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline
python tools/conv_bench16.py 2>&1 | tail -30 > $O/conv_bench16.txt; head -40 $O/conv_bench16.txt
