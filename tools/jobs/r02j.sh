#!/bin/bash
O=gpurun_out/r02j; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "fused_bn_backward" > $O/tests_bnb.log 2>&1; echo "bnb op tests rc=$?"; tail -4 $O/tests_bnb.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_model16_gpu.py -m gpu -x -q -k "nbt1d or pinned_gradients_small or full_model_small or bf16_pinned or hipgraph_train" > $O/tests_model.log 2>&1; echo "model tests rc=$?"; tail -4 $O/tests_model.log
for dt in f32 bf16; do
  for v in fuse nofuse; do
    case $v in
      fuse) env_="" ;;
      nofuse) env_="EMSA_BN_FUSE=0" ;;
    esac
    env $env_ timeout 600 python bench.py --dtype $dt --no-cpu-baseline > $O/bench_${dt}_$v.json 2> $O/bench_${dt}_$v.err
    python - <<PY
import json
try:
    d = json.loads(open('$O/bench_${dt}_$v.json').read().strip().splitlines()[-1])
    print('$dt $v', d['value'], d['ms_per_step'])
except Exception as e:
    print('$dt $v failed', e)
PY
  done
done
