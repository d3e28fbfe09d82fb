#!/bin/bash
O=gpurun_out/r02q; mkdir -p $O
for b in 256 384 512 768 1024; do
  EMSA_W1D_BLOCKS=$b timeout 600 python tools/conv_bench16.py wgrad > $O/cb16_wgrad_b$b.txt 2>&1
  EMSA_W1D_BLOCKS=$b timeout 900 python tools/conv_bench.py wgrad > $O/cb32_wgrad_b$b.txt 2>&1
done
for b in 256 384 512 768 1024; do echo "== bf16 blocks $b"; sed -n 4,12p $O/cb16_wgrad_b$b.txt | awk '{printf "%s %s %s %s\n", $1,$2,$3,$5}'; done
for b in 256 384 512 768 1024; do echo "== f32 blocks $b"; grep -i "wgrad\|c64\|c128\|c256\|c512" $O/cb32_wgrad_b$b.txt | head -12 | cut -c1-150; done
