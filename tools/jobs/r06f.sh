#!/bin/bash
O=gpurun_out/r06f; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_ops16_gpu.py -m gpu -q -x -k "ppm or bilinear" > $O/ppm.log 2>&1; echo "ppm rc=$?"; tail -3 $O/ppm.log
EMSA_DUAL_STREAM=0 python tools/grad_repeat_probe.py bf16 2>&1 | tail -8
python tools/grad_repeat_probe.py bf16 2>&1 | tail -8
timeout 1500 python -m pytest tests/test_timed_size_gpu.py -m gpu -q -s -k "linear" > $O/new_tests.log 2>&1; echo "new tests rc=$?"
grep -n "rel-L2\|^E  \|passed\|failed\|bs-32 vs\|train bs 32\|real sample" $O/new_tests.log | cut -c1-330 | head -40
