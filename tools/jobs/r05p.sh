#!/bin/bash
O=gpurun_out/r05p; mkdir -p $O
for g in 128 256 384 512; do echo "== EMSA_UP2X_BWD_WGS=$g"; EMSA_UP2X_BWD_WGS=$g python tools/up2x_bwd_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/up2x_bwd_$g.txt; done
