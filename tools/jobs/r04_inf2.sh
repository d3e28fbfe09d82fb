#!/bin/bash
O=gpurun_out/r04_inf2; mkdir -p $O; R=$PWD
timeout 600 tools/bin/conv_rs_probe 32 check > $O/probe_check.txt 2>&1; echo "probe rc=$?"; tail -1 $O/probe_check.txt
timeout 1500 python -m pytest tests/test_conv_rs_gpu.py -x -q > $O/t_rs.log 2>&1; echo "rs tests rc=$?"; tail -2 $O/t_rs.log
timeout 3000 python -m pytest tests/test_model16_gpu.py -x -q > $O/t_m16.log 2>&1; echo "model16 rc=$?"; tail -2 $O/t_m16.log
timeout 2400 python -m pytest tests/test_model_gpu.py -x -q -k "hipgraph or eval or golden" > $O/t_m.log 2>&1; echo "model rc=$?"; tail -2 $O/t_m.log
for dt in f16 bf16 f32; do
timeout 600 python bench.py --eval --graph --batch-size 1 --dtype $dt --steps 200 --warmup 20 --no-cpu-baseline > $O/eval_$dt.json 2>$O/eval_$dt.err; python -c "
import json; d=json.loads(open('$O/eval_$dt.json').read().strip().splitlines()[-1]); print('$dt graph bs1', d['value'], d['ms_per_step'])"
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p --output-format csv -- python $R/bench.py --eval --graph --batch-size 1 --dtype f16 --steps 100 --warmup 10 --no-cpu-baseline > $R/$O/prof.log 2>&1; echo "prof rc=$?"
cd $R
f=$(ls $O/prof/*kernel_stats.csv | head -1); python tools/stats_csv_to_md.py $f 110 "eval bs=1 fp16 graph (100 steps + 10 warm-up)" > $O/eval_f16_kernel_stats.md; head -24 $O/eval_f16_kernel_stats.md | cut -c1-180
