#!/bin/bash
O=gpurun_out/r03j; mkdir -p $O
R=$GRAFT_REPO_ROOT
echo "=== op tests"
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_ops16_gpu.py -x -q 2>&1 | tail -4
echo "=== model / parallel tests"
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_parallel_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -4
echo "=== pointwise bench"
timeout 600 python tools/pointwise_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/pointwise_bench.txt | grep "up2x"
echo "--- quad form"; EMSA_UP2X_ROWS=0 timeout 600 python tools/pointwise_bench.py 2>&1 | grep "up2x"
echo "=== 3x3 512 shapes, pinned vs not"
EMSA_BENCH_SHAPE="512" timeout 300 python tools/conv_bench.py wino 2>&1 | grep -v amdgpu.ids
EMSA_WINO_PIN=0 EMSA_BENCH_SHAPE="512" timeout 300 python tools/conv_bench.py wino 2>&1 | grep -v amdgpu.ids
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('$name', d['value'], d['ms_per_step'], r.get('frac'))
except Exception as e:
    print('$name failed', e)
PY
}
run f32_all --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
EMSA_UP2X_ROWS=0 run f32_quad_up2x --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
EMSA_WINO_PIN=0 run f32_nopin --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
run f32_all_b --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
run bf16_all --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
