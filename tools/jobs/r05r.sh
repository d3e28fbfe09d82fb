#!/bin/bash
O=gpurun_out/r05r; mkdir -p $O
timeout 1400 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_gpu.log
