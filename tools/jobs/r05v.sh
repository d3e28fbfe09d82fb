#!/bin/bash
O=gpurun_out/r05v; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops16_gpu.py -k "pool" -m gpu -q -x > $O/ops.log 2>&1; echo "ops rc=$?"; tail -2 $O/ops.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in new prev; do
  if [ $lib = prev ]; then export EMSA_LIB=$R/build/prev/libemsanet_hip.so; else unset EMSA_LIB; fi
  EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$lib -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $R/$O/prof_$lib.log 2>&1
  grep -h "maxpool" $R/$O/prof_$lib/*kernel_stats.csv | cut -c1-40,60-200 | head -4
done
find $R/$O -name "*kernel_trace*" -delete
