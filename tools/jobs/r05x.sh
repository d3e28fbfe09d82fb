#!/bin/bash
# double-buffered LDS in the tr weight-gradient kernel: tests, conv_bench16 wgrad rows, same-box A/B (build/prev = HEAD)
O=gpurun_out/r05x; mkdir -p $O
timeout 900 python -m pytest tests/test_ops16_gpu.py -k "wgrad or conv16" -m gpu -q -x > $O/ops16.log 2>&1; echo "ops16 rc=$?"; tail -2 $O/ops16.log
python tools/conv_bench16.py 2>&1 | grep "wgrad" | head -12 | tee $O/conv_bench16_new.txt
EMSA_LIB=$PWD/build/prev/libemsanet_hip.so python tools/conv_bench16.py 2>&1 | grep "wgrad" | head -12 | tee $O/conv_bench16_prev.txt
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
A="--dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline"
for rep in 1 2 3; do
run bf16_new_$rep $A
EMSA_LIB=$PWD/build/prev/libemsanet_hip.so run bf16_prev_$rep $A
done
run bf16_new_4 $A
