#!/bin/bash
O=gpurun_out/r05q2; mkdir -p $O
timeout 600 python -m pytest tests/test_ops16_gpu.py tests/test_ops_gpu.py -k "up or ppm" -m gpu -q -x > $O/up.log 2>&1; echo "up tests rc=$?"; tail -2 $O/up.log
python tools/up2x_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/up2x_bwd.txt
