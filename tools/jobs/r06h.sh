#!/bin/bash
# concurrency inside the bf16 training-step graph: default vs conv_rs on half the CUs
O=gpurun_out/r06h; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in default cus128; do
  if [ $v = cus128 ]; then export EMSA_RS_CUS=128; else unset EMSA_RS_CUS; fi
  timeout 900 rocprofv3 --kernel-trace -d $R/$O/prof_$v -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > $R/$O/prof_$v.log 2>&1; echo "prof $v rc=$?"
  f=$(ls $R/$O/prof_$v/*kernel_trace.csv 2>/dev/null | head -1)
  python $R/tools/overlap_stats.py $f 1498 1 > $R/$O/overlap_$v.txt 2>&1; cat $R/$O/overlap_$v.txt | cut -c1-150
done
find $R/$O -name "*kernel_trace*" -delete
