#!/bin/bash
O=gpurun_out/r02y; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests_gpu.log; grep -n "^FAILED" $O/tests_gpu.log | head
