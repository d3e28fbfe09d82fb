#!/bin/bash
O=gpurun_out/r02u; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops16_gpu.py tests/test_model16_gpu.py -m gpu -x -q > $O/tests16.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests16.log
for v in short long short2 long2; do
  case $v in
    short*) env_="" ;;
    long*) env_="EMSA_CONVH_SHORTK=0" ;;
  esac
  env $env_ timeout 600 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d = json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], [(k['kernel'][:14], k['avg_us']) for k in d['conv_kernels'][:1]])
PY
done
for dt in bf16 f16; do
timeout 600 python bench.py --dtype $dt --eval --graph --batch-size 1 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$dt bs1 graph', d['value'], d['ms_per_step'])"
done
