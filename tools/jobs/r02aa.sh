#!/bin/bash
O=gpurun_out/r02aa; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops16_gpu.py -m gpu -x -q -k "se_ or pool_se or channel" > $O/tests.log 2>&1; echo "se tests rc=$?"; tail -3 $O/tests.log
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_model16_gpu.py -m gpu -x -q -k "full_model_small or eval_16bit_vs_fp32 or bf16_training_step" > $O/tests_model.log 2>&1; echo "model tests rc=$?"; tail -3 $O/tests_model.log
timeout 600 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bf16', d['value'], d['ms_per_step'])"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bf16 -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/$O/prof_bf16.log 2>&1
cd $R; find $O -name "*kernel_trace*" -delete
grep -i "channel_dot\|se_scale" $O/prof_bf16/p_kernel_stats.csv | cut -c1-140
