#!/bin/bash
O=gpurun_out/r04_w8; mkdir -p $O
echo "== default (4-wave workgroups, two per CU)"; timeout 600 tools/bin/conv_rs_probe 32 time 2>&1 | grep -v "^arch\|ALL OK" | grep "c64\|c128\|c256\|shape"
echo "== EMSA_RS_W8=7 (8-wave workgroups, one per CU)"; LD_LIBRARY_PATH=tools/bin/w8 timeout 600 tools/bin/conv_rs_probe 32 all 2>&1 | grep -v "^arch" | grep "c64 \|c128 \|c256 \|shape\|ALL\|FAIL" | grep -v " ok$"
