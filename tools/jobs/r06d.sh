#!/bin/bash
O=gpurun_out/r06d; mkdir -p $O
timeout 1500 python -m pytest tests/test_timed_size_gpu.py tests/test_sample_pair.py -m gpu -q -s > $O/new_tests.log 2>&1; echo "new tests rc=$?"
grep -n "rel-L2\|^E  \|passed\|failed\|bs-32 vs\|train bs 32\|real sample" $O/new_tests.log | cut -c1-330 | head -40
