#!/bin/bash
O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -k "other_resnet or rgbd or normal" > $O/t1.log 2>&1; tail -25 $O/t1.log
