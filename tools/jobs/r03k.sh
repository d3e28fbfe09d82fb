#!/bin/bash
O=gpurun_out/r03k; mkdir -p $O
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "train_step_variants" 2>&1 | grep -v "^  \|^$" | tail -30
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "upsample" 2>&1 | tail -3
echo "=== up2x fwd, default stores"; timeout 600 python tools/pointwise_bench.py 2>&1 | grep "up2x" | tee $O/pointwise_default.txt
echo "=== up2x fwd, non-temporal stores"; EMSA_UP2X_NT=1 timeout 600 python tools/pointwise_bench.py 2>&1 | grep "up2x" | tee $O/pointwise_nt.txt
