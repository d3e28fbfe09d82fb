#!/bin/bash
O=gpurun_out/r05w; mkdir -p $O
timeout 1500 python -m pytest tests/test_model16_gpu.py -m gpu -x -q > $O/m16.log 2>&1; echo "m16 rc=$?"; tail -3 $O/m16.log
