#!/bin/bash
O=gpurun_out/r02v; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "dgrad" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "nbt1d or pinned_gradients_small or hipgraph_train" > $O/tests_model.log 2>&1; echo "model tests rc=$?"; tail -3 $O/tests_model.log
for v in phases single phases2 single2; do
  case $v in
    phases*) env_="" ;;
    single*) env_="EMSA_DGRAD_PHASES=0" ;;
  esac
  env $env_ timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d = json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], [(k['kernel'][:22], k['avg_us'], k['launches']) for k in d['conv_kernels'][2:3]])
PY
done
