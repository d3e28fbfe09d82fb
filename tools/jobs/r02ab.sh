#!/bin/bash
O=gpurun_out/r02ab; mkdir -p $O
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "hipgraph_train" -p no:cacheprovider > $O/t$i.log 2>&1; echo "run $i rc=$?"; grep -E "AssertionError: \(|passed|failed" $O/t$i.log | head -3
done
