#!/bin/bash
O=gpurun_out/r02w; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops16_gpu.py -m gpu -x -q -k "conv16" > $O/tests16.log 2>&1; echo "tests16 rc=$?"; tail -3 $O/tests16.log
timeout 1200 python -m pytest tests/test_model16_gpu.py -m gpu -x -q > $O/tests_model16.log 2>&1; echo "model16 rc=$?"; tail -3 $O/tests_model16.log
for v in phases single phases2 single2; do
  case $v in
    phases*) env_="" ;;
    single*) env_="EMSA_DGRAD_PHASES=0" ;;
  esac
  env $env_ timeout 600 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d = json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'])
PY
done
