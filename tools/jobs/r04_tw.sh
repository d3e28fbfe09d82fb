#!/bin/bash
# twin launches (emsa_conv1d_rs_pair_t): parity tests, batch-1 graph and batch-32 eval A/B
O=gpurun_out/r04tw; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_rs_gpu.py -m gpu -x -q -k "pair" > $O/tests_pair.log 2>&1; echo "tests pair rc=$?"; tail -3 $O/tests_pair.log
timeout 900 python -m pytest tests/test_model16_gpu.py -m gpu -x -q -k "twin or hipgraph_inference" > $O/tests_twin.log 2>&1; echo "tests twin rc=$?"; tail -5 $O/tests_twin.log
run() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
for dt in f16 bf16; do
EMSA_TWIN=1 EMSA_DUAL_STREAM=1 run b1_${dt}_twin1_ds1 --eval --graph --batch-size 1 --dtype $dt --steps 200 --warmup 20
EMSA_TWIN=0 EMSA_DUAL_STREAM=1 run b1_${dt}_twin0_ds1 --eval --graph --batch-size 1 --dtype $dt --steps 200 --warmup 20
EMSA_TWIN=1 EMSA_DUAL_STREAM=0 run b1_${dt}_twin1_ds0 --eval --graph --batch-size 1 --dtype $dt --steps 200 --warmup 20
EMSA_TWIN=0 EMSA_DUAL_STREAM=0 run b1_${dt}_twin0_ds0 --eval --graph --batch-size 1 --dtype $dt --steps 200 --warmup 20
done
EMSA_TWIN=1 run b32_bf16_twin1 --eval --batch-size 32 --dtype bf16 --steps 20 --warmup 5
EMSA_TWIN=0 run b32_bf16_twin0 --eval --batch-size 32 --dtype bf16 --steps 20 --warmup 5
EMSA_TWIN=1 EMSA_DUAL_STREAM=0 run b32_bf16_twin1_ds0 --eval --batch-size 32 --dtype bf16 --steps 20 --warmup 5
EMSA_TWIN=0 EMSA_DUAL_STREAM=0 run b32_bf16_twin0_ds0 --eval --batch-size 32 --dtype bf16 --steps 20 --warmup 5
