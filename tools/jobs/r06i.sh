#!/bin/bash
O=gpurun_out/r06i; mkdir -p $O
timeout 1500 python -m pytest tests/test_timed_size_gpu.py tests/test_sample_pair.py -m gpu -q -s > $O/new_tests.log 2>&1; echo "new tests rc=$?"
grep -n "^E  \|passed\|failed\|bs-32 vs\|train bs 32\|real sample" $O/new_tests.log | cut -c1-400 | head -30
timeout 1500 python -m pytest tests/test_model16_gpu.py tests/test_ops16_gpu.py -m gpu -q -s -k "pinned or wgrad_multi" > $O/m16.log 2>&1; echo "m16 rc=$?"; grep -n "^E  \|passed\|failed\|norm ratio\|gradient-norm gain" $O/m16.log | cut -c1-500 | head
