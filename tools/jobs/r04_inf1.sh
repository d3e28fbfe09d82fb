#!/bin/bash
# inference bs=1 (configs[4]): latency + kernel stats of the graph replay
O=gpurun_out/r04_inf1; mkdir -p $O; R=$PWD
for dt in f16 bf16 f32; do
timeout 600 python bench.py --eval --graph --batch-size 1 --dtype $dt --steps 200 --warmup 20 --no-cpu-baseline > $O/eval_$dt.json 2>$O/eval_$dt.err; python -c "
import json; d=json.loads(open('$O/eval_$dt.json').read().strip().splitlines()[-1]); print('$dt graph bs1', d['value'], d['ms_per_step'])"
done
EMSA_DUAL_STREAM=0 timeout 600 python bench.py --eval --graph --batch-size 1 --dtype f16 --steps 200 --warmup 20 --no-cpu-baseline > $O/eval_f16_1s.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/eval_f16_1s.json').read().strip().splitlines()[-1]); print('f16 graph bs1 one stream', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p --output-format csv -- python $R/bench.py --eval --graph --batch-size 1 --dtype f16 --steps 100 --warmup 10 --no-cpu-baseline > $R/$O/prof.log 2>&1; echo "prof rc=$?"
cd $R
f=$(ls $O/prof/*kernel_stats.csv | head -1); python tools/stats_csv_to_md.py $f 110 "eval bs=1 fp16 graph (100 steps + 10 warm-up)" > $O/eval_f16_kernel_stats.md; head -40 $O/eval_f16_kernel_stats.md | cut -c1-180
