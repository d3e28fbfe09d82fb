#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -k "rgbd or normal or pinned or resnet18 or config1" > $O/t1.log 2>&1; tail -3 $O/t1.log
timeout 900 python -m pytest tests/test_boundary_gpu.py -x -q > $O/t2.log 2>&1; tail -2 $O/t2.log
