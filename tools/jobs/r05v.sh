#!/bin/bash
O=gpurun_out/r05v; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops16_gpu.py -k "se or pool or channel" -m gpu -q -x > $O/ops.log 2>&1; echo "ops rc=$?"; tail -2 $O/ops.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for dt in bf16 f32; do
if [ $dt = bf16 ]; then D="--dtype bf16"; else D=""; fi
for lib in new prev; do
  if [ $lib = prev ]; then export EMSA_LIB=$R/build/prev/libemsanet_hip.so; else unset EMSA_LIB; fi
  EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_${dt}_$lib -o p --output-format csv -- python $R/bench.py $D --steps 10 --warmup 3 --no-cpu-baseline > $R/$O/prof_$lib.log 2>&1
  echo "$dt $lib"; grep -h "channel_dot_kernel" $R/$O/prof_${dt}_$lib/*kernel_stats.csv | sed 's/"[^"]*"//' | head -3
done; done
find $R/$O -name "*kernel_trace*" -delete
