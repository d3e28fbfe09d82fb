#!/bin/bash
# half-block kernel iteration: tests + configs[4] A/B + kernel trace of the bs-1 graph
O=gpurun_out/r05e; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('$name', d['value'], d['ms_per_step'], r.get('frac'), (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
timeout 900 python -m pytest tests/test_conv_rs_gpu.py -k "half_block" -m gpu -q -x > $O/hb_ops.log 2>&1; echo "half-block op tests rc=$?"; tail -2 $O/hb_ops.log
timeout 900 python -m pytest tests/test_model16_gpu.py -k "half_blocks or eval_bn" -m gpu -q -x -s > $O/hb_model.log 2>&1; echo "model tests rc=$?"; grep -h "eval-BN\|passed\|failed" $O/hb_model.log | cut -c1-700
for rep in 1 2; do
for dt in f16 bf16; do
  run config4_${dt}_half_block_$rep --dtype $dt --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline
  EMSA_HALF_BLOCK=0 run config4_${dt}_half_block_off_$rep --dtype $dt --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline
done
done
run eval_b2_f16_hb_on --dtype f16 --eval --graph --batch-size 2 --steps 200 --warmup 20 --no-cpu-baseline
EMSA_HALF_BLOCK=0 run eval_b2_f16_hb_off --dtype f16 --eval --graph --batch-size 2 --steps 200 --warmup 20 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_eval_bs1_f16 -o p --output-format csv -- python $R/bench.py --eval --graph --batch-size 1 --dtype f16 --steps 100 --warmup 10 --no-cpu-baseline > $R/$O/prof_eval.log 2>&1; echo "prof eval rc=$?"
cd $R
f=$(ls $O/prof_eval_bs1_f16/*kernel_trace.csv 2>/dev/null | head -1)
python tools/graph_timeline.py $f 136 2 > $O/eval_bs1_f16_timeline.txt 2>&1; grep -i "half_block" $O/eval_bs1_f16_timeline.txt | head -25; tail -3 $O/eval_bs1_f16_timeline.txt
python tools/stats_csv_to_md.py $(ls $O/prof_eval_bs1_f16/*kernel_stats.csv | head -1) 110 "r05_e: rocprofv3 --kernel-trace --stats -- python bench.py --eval --graph --batch-size 1 --dtype f16 --steps 100 --warmup 10 (configs[4]: whole-model hipGraph, twin launches + fused half-blocks, per forward)" > $O/eval_bs1_f16_kernel_stats.md
rm -rf $O/prof_*/
