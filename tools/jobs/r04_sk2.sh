#!/bin/bash
O=gpurun_out/r04_sk2; mkdir -p $O; R=$PWD
for i in 1 2 3; do
for sk in 1 0; do
EMSA_CONVH_SPLITK=$sk timeout 600 python bench.py --eval --graph --batch-size 1 --dtype f16 --steps 300 --warmup 30 --no-cpu-baseline > $O/e.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/e.json').read().strip().splitlines()[-1]); print('f16 graph bs1 splitk=$sk', d['value'], d['ms_per_step'])"
done; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p --output-format csv -- python $R/bench.py --eval --graph --batch-size 1 --dtype f16 --steps 100 --warmup 10 --no-cpu-baseline > $R/$O/prof.log 2>&1
cd $R
python tools/stats_csv_to_md.py $(ls $O/prof/*kernel_stats.csv | head -1) 110 "x" | grep "conv_h\|splitk\|avgpool\|channel_dot\|total" | cut -c1-170
