#!/bin/bash
# end-of-round evidence: tests, smoke, bench lines of every BASELINE config, profiles
O=gpurun_out/r03z; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('$name', d['value'], d['ms_per_step'], r.get('frac'), (r.get('in_timed_region') or {}).get('frac'), (d.get('h2d_staged') or {}).get('value'), (d.get('comm') or {}).get('exposed_comm_ms_per_step'))
except Exception as e:
    print('$name failed', e)
PY
}
timeout 3000 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests_gpu.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
run bench_driver_cmd_f32 --gpus 1 --steps 20 --warmup 5
EMSA_DUAL_STREAM=0 run bench_f32_one_stream --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
run bench_bf16 --dtype bf16 --steps 20 --warmup 5 --h2d --no-cpu-baseline
run bench_bf16_forcedist_bf16grads --dtype bf16 --grad-dtype bf16 --force-dist --steps 20 --warmup 5 --no-cpu-baseline
run bench_bf16_forcedist_segmented_graph --dtype bf16 --force-dist --graph --steps 20 --warmup 5 --no-cpu-baseline
run bench_f32_forcedist --force-dist --steps 20 --warmup 5 --no-cpu-baseline
run bench_f32_losses --losses --steps 40 --warmup 5 --no-cpu-baseline
run bench_bf16_losses --dtype bf16 --losses --steps 40 --warmup 5 --no-cpu-baseline
run bench_f32_graph --graph --steps 20 --warmup 5 --no-cpu-baseline
run bench_bf16_graph --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
run config3_r101_960x736_bs16_f32 --backbone resnet101 --height 736 --width 960 --batch-size 16 --steps 6 --warmup 2 --no-cpu-baseline
run config3_r101_960x736_bs16_bf16 --dtype bf16 --backbone resnet101 --height 736 --width 960 --batch-size 16 --steps 6 --warmup 2 --no-cpu-baseline
for dt in f32 bf16 f16; do
  run config4_eval_graph_bs1_$dt --dtype $dt --eval --graph --batch-size 1 --steps 200 --warmup 20 --no-cpu-baseline
done
run eval_bs32_f32 --eval --steps 20 --warmup 5 --no-cpu-baseline
run eval_bs32_bf16 --dtype bf16 --eval --steps 20 --warmup 5 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_f32 -o p --output-format csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_f32.log 2>&1; echo "prof f32 rc=$?"
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_f32_one_stream -o p --output-format csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_f32_one.log 2>&1; echo "prof f32 one stream rc=$?"
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bf16_one_stream -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_bf16_one.log 2>&1; echo "prof bf16 one stream rc=$?"
cd $R; find $O -name "*kernel_trace*" -delete
timeout 600 python tools/conv_bench16.py all > $O/conv_bench16.txt 2>&1
timeout 900 python tools/conv_bench.py > $O/conv_bench_f32.txt 2>&1
timeout 600 python tools/pointwise_bench.py > $O/pointwise_bench.txt 2>&1
