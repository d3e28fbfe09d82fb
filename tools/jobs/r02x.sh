#!/bin/bash
O=gpurun_out/r02x; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops16_gpu.py -m gpu -x -q -k "stem" > $O/tests.log 2>&1; echo "stem tests rc=$?"; tail -3 $O/tests.log
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_model16_gpu.py tests/test_golden_gpu.py -m gpu -x -q -k "full_model_small or pinned_gradients_small or full_res_eval or eval_16bit or golden or hipgraph" > $O/tests_model.log 2>&1; echo "model tests rc=$?"; tail -3 $O/tests_model.log
for v in rows generic rows2 generic2; do
  case $v in
    rows*) env_="" ;;
    generic*) env_="EMSA_STEM_ROWS=0" ;;
  esac
  env $env_ timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d = json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'])
PY
done
EMSA_STEM_ROWS=1 timeout 600 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 rows', d['value'], d['ms_per_step'])"
EMSA_STEM_ROWS=0 timeout 600 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16 generic', d['value'], d['ms_per_step'])"
