#!/bin/bash
O=gpurun_out/r04_wh1; mkdir -p $O
timeout 900 python -m pytest tests/test_ops16_gpu.py -x -q -k "wgrad" > $O/t.log 2>&1; echo "tests rc=$?"; tail -2 $O/t.log
timeout 600 python tools/conv_bench16.py wgrad > $O/occ2.txt 2>&1; cat $O/occ2.txt
EMSA_LIB=$PWD/tools/bin/wh3/libemsanet_hip.so timeout 600 python tools/conv_bench16.py wgrad > $O/occ3.txt 2>&1; cat $O/occ3.txt
for b in 512 1024; do echo "blocks $b"; EMSA_W1D_BLOCKS=$b timeout 600 python tools/conv_bench16.py wgrad 2>&1 | grep -v "^lib\|^shape"; done
