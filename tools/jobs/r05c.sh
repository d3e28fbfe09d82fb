#!/bin/bash
# round 5, third GPU call: whole suite with the oracle thread cap (time!), wide statistics merge A/B,
# bf16 one-stream kernel table with the multi-job weight gradients, default bench line
O=gpurun_out/r05c; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('$name', d['value'], d['ms_per_step'], r.get('frac'), (d.get('whole_step') or {}).get('hbm_frac'), (d.get('hipgraph') or {}).get('nodes'), (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('cores'))
except Exception as e:
    print('$name failed', e)
PY
}
( time timeout 2400 python -m pytest tests -m gpu -q --durations=25 ) > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests_gpu.log
grep -h "eval-BN\|gradient-norm gain" $O/tests_gpu.log | head
timeout 600 python -m pytest "tests/test_model16_gpu.py::test_eval_bn_bf16_pinned_gradients_baseline_resolution" "tests/test_model16_gpu.py::test_train_bf16_pinned_gradients" -m gpu -q -s > $O/bf16_grad_tests.log 2>&1; echo "bf16 grad tests rc=$?"; grep -h "eval-BN\|gradient-norm gain\|passed\|failed" $O/bf16_grad_tests.log | cut -c1-900
run f32_driver_cmd --gpus 1 --steps 20 --warmup 5
EMSA_BN_FINALIZE_WIDE=0 run f32_finalize_two_launches --steps 20 --warmup 5 --no-cpu-baseline
run f32_finalize_wide --steps 20 --warmup 5 --no-cpu-baseline
EMSA_BN_FINALIZE_WIDE=0 run f32_finalize_two_launches_b --steps 20 --warmup 5 --no-cpu-baseline
run f32_finalize_wide_b --steps 20 --warmup 5 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bf16_one_stream -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_bf16_one.log 2>&1; echo "prof bf16 one stream rc=$?"
cd $R
find $O -name "*kernel_trace*" -delete
python tools/stats_csv_to_md.py $(ls $O/prof_bf16_one_stream/*kernel_stats.csv | head -1) 25 "r05_c: EMSA_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline (bf16 storage, ONE stream; multi-job weight gradients)" > $O/bf16_one_stream_kernel_stats.md
rm -rf $O/prof_*/
head -40 $O/bf16_one_stream_kernel_stats.md | cut -c1-150
