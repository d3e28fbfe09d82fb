#!/bin/bash
# rocprofv3 kernel stats of the bf16 step with conv_rs: one stream and two streams
O=gpurun_out/r04_rs5; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_one -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $R/$O/prof_one.log 2>&1; echo "one stream rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_two -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $R/$O/prof_two.log 2>&1; echo "two streams rc=$?"
cd $R
for d in one two; do f=$(ls $O/prof_$d/*kernel_stats.csv 2>/dev/null | head -1); echo "== $d $f"; python tools/stats_csv_to_md.py $f 13 > $O/${d}_kernel_stats.md 2>/dev/null; head -30 $O/${d}_kernel_stats.md | cut -c1-200; done
tail -2 $O/prof_one.log | cut -c1-300
