#!/bin/bash
O=gpurun_out/r05y; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --durations=10 ) > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests_gpu.log
python tools/pointwise_bench16.py bf16 f32 2>&1 | grep -v amdgpu.ids > $O/pointwise_bench16.txt; cat $O/pointwise_bench16.txt
python tools/up2x_bwd_bench.py 2>&1 | grep -v amdgpu.ids > $O/up2x_bwd.txt; tail -1 $O/up2x_bwd.txt
