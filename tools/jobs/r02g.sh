#!/bin/bash
# GPU job (run as: gpurun -- bash tools/jobs/r02g.sh): new kernels first, then measurements
O=gpurun_out/r02g; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops16_gpu.py tests/test_staging.py -m gpu -x -q -k "upsample or up2x or stager or staged or maxpool" > $O/tests_new.log 2>&1; echo "new tests rc=$?"; tail -3 $O/tests_new.log
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "wgrad_side_stream or hipgraph_train" > $O/tests_side.log 2>&1; echo "side tests rc=$?"; tail -3 $O/tests_side.log
for dt in f32 bf16; do
  for v in base fusedoff quad side; do
    case $v in
      base) env_="" ; fl="" ;;
      fusedoff) env_="EMSA_UP2X_FUSED=0"; fl="" ;;
      quad) env_="EMSA_UP2X_FWD=quad"; fl="" ;;
      side) env_=""; fl="--wgrad-stream" ;;
    esac
    env $env_ timeout 600 python bench.py --dtype $dt --no-cpu-baseline $fl > $O/bench_${dt}_$v.json 2> $O/bench_${dt}_$v.err
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
timeout 600 python tools/torch_ops_profile.py --stacks > $O/torch_ops.txt 2>&1; echo "torch_ops rc=$?"
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --h2d --steps 6 > $O/bench_bf16_h2d.json 2> $O/bench_bf16_h2d.err; tail -c 600 $O/bench_bf16_h2d.json
