#!/bin/bash
O=gpurun_out/r04_wg1; mkdir -p $O
for b in 768 512 384 256; do
echo "== EMSA_W1D_BLOCKS=$b"; EMSA_W1D_BLOCKS=$b timeout 600 python tools/conv_bench16.py wgrad 2>/dev/null | grep -v "^lib\|3x3\|1x1\|s2" 
done
