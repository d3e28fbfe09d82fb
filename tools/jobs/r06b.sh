#!/bin/bash
O=gpurun_out/r06b; mkdir -p $O
timeout 1500 python -m pytest tests/test_timed_size_gpu.py tests/test_sample_pair.py -m gpu -q -s > $O/new_tests.log 2>&1; echo "new tests rc=$?"; grep -v "^$" $O/new_tests.log | tail -60
