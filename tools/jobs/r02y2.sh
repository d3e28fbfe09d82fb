#!/bin/bash
O=gpurun_out/r02y; mkdir -p $O
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_model16_gpu.py tests/test_golden_gpu.py -m gpu -q -s -p no:cacheprovider > $O/tests_model_s.log 2>&1; echo "rc=$?"
grep -E "pinned parity|bf16 train|rel-L2|worst|max rel|agreement|err " $O/tests_model_s.log | cut -c1-220 | head -60
