#!/bin/bash
# bf16: fused BatchNorm-backward reduction in the conv_rs data gradient -- at which channel counts does it pay?
O=gpurun_out/r05f; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
timeout 600 python -m pytest "tests/test_model16_gpu.py::test_train_bf16_pinned_gradients" -m gpu -q -x > $O/test.log 2>&1; echo "test rc=$?"; tail -2 $O/test.log
for rep in 1 2; do
run bf16_graph_fuse_all_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_BN_FUSE_MIN_C=256 run bf16_graph_fuse_c256_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_BN_FUSE_MIN_C=128 run bf16_graph_fuse_c128_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_BN_FUSE=0 run bf16_graph_fuse_none_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
done
