#!/bin/bash
O=$PWD/gpurun_out/r04_bn1; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
EMSA_DUAL_STREAM=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o p -- python $R/bench.py --dtype bf16 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --roofline-steps 0 > $O/log.txt 2>&1
python $R/tools/bn_trace.py $(find $O/tr -name "*kernel_trace.csv" | head -1)
head -1 $(find $O/tr -name "*kernel_trace.csv" | head -1)
rm -rf $O/tr
