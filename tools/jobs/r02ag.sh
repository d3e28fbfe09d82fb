#!/bin/bash
O=gpurun_out/r02ag; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops16_gpu.py -m gpu -x -q -k "se_ or pool_se or channel" > $O/tests.log 2>&1; echo "se tests rc=$?"; tail -2 $O/tests.log
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_model16_gpu.py tests/test_golden_gpu.py tests/test_parallel_gpu.py -m gpu -x -q -p no:cacheprovider > $O/tests_model.log 2>&1; echo "model tests rc=$?"; tail -2 $O/tests_model.log
timeout 300 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'])"
