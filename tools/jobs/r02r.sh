#!/bin/bash
O=gpurun_out/r02r; mkdir -p $O
for i in 1 2; do
  timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_gpu_$i.log 2>&1; echo "run $i rc=$?"; tail -2 $O/tests_gpu_$i.log
done
