#!/bin/bash
O=gpurun_out/r02o; mkdir -p $O
for t in 0 1 2; do
  EMSA_CONVH_TILE=$t timeout 600 python tools/conv_bench16.py fwd > $O/cb16_fwd_tile$t.txt 2>&1
  EMSA_CONVH_TILE=$t timeout 600 python tools/conv_bench16.py dgrad > $O/cb16_dgrad_tile$t.txt 2>&1
done
timeout 600 python tools/conv_bench16.py all > $O/cb16_all_auto.txt 2>&1
for t in 0 1 2; do echo "== fwd tile $t"; sed -n 4,20p $O/cb16_fwd_tile$t.txt | awk '{printf "%s %s %s %s\n", $1,$2,$3,$5}' ; done
