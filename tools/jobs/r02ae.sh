#!/bin/bash
O=gpurun_out/r02ae; mkdir -p $O
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_model16_gpu.py -m gpu -x -q -k "hipgraph or full_model_small or eval_16bit_vs_fp32" -p no:cacheprovider > $O/t.log 2>&1; echo "rc=$?"; tail -2 $O/t.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f32', d['value'], d['ms_per_step'])"
