#!/bin/bash
O=gpurun_out/r04_wh2; mkdir -p $O
echo base; timeout 600 python tools/conv_bench16.py wgrad 2>&1 | grep "1x3\|3x1\|1x1" 
for e in 1 2 4 8 9 6; do echo "exp $e"; EMSA_LIB=$PWD/tools/bin/whe$e/libemsanet_hip.so timeout 600 python tools/conv_bench16.py wgrad 2>&1 | grep "1x3\|3x1\|1x1"; done
