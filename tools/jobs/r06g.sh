#!/bin/bash
O=gpurun_out/r06g; mkdir -p $O
timeout 1500 python -m pytest tests/test_timed_size_gpu.py tests/test_sample_pair.py -m gpu -q -s > $O/new_tests.log 2>&1; echo "new tests rc=$?"
grep -n "^E  \|passed\|failed\|bs-32 vs\|train bs 32\|real sample" $O/new_tests.log | cut -c1-400 | head -30
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_timed_size_gpu.py --deselect tests/test_sample_pair.py > $O/suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/suite.log | cut -c1-300
