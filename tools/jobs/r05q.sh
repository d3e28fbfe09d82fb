#!/bin/bash
O=gpurun_out/r05q; mkdir -p $O
timeout 600 python -m pytest tests/test_ops16_gpu.py tests/test_ops_gpu.py -k "up or ppm" -m gpu -q -x > $O/up.log 2>&1; echo "up tests rc=$?"; tail -2 $O/up.log
python tools/up2x_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/up2x_bwd.txt
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
B="--steps 20 --warmup 5 --no-cpu-baseline"
for rep in 1 2; do
run bf16_new_$rep $A
EMSA_UP2X_BWD_WGS=1024 run bf16_1024_$rep $A
run f32_new_$rep $B
EMSA_UP2X_BWD_WGS=1024 run f32_1024_$rep $B
done
