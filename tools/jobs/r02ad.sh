#!/bin/bash
O=gpurun_out/r02ad; mkdir -p $O
for v in A B C; do
  EMSA_LIB=$GRAFT_REPO_ROOT/emsanet_amd/lib/var_se$v/libemsanet_hip.so timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "hipgraph_train" -p no:cacheprovider > $O/t$v.log 2>&1; echo "variant $v rc=$?"; grep -E "AssertionError: \(|passed|failed" $O/t$v.log | head -2
done
