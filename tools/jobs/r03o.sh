#!/bin/bash
R=$GRAFT_REPO_ROOT
echo "== wino default stores"; timeout 600 python tools/conv_bench.py wino 2>&1 | grep -v amdgpu | head -8
echo "== wino non-temporal stores"; EMSA_LIB=$R/emsanet_amd/lib/var_nt/libemsanet_hip.so timeout 600 python tools/conv_bench.py wino 2>&1 | grep -v amdgpu | head -8
echo "== c64 at batch 8: cold operands vs one buffer set (Infinity-Cache resident)"
EMSA_BENCH_N=8 EMSA_BENCH_SHAPE="c64" timeout 300 python tools/conv_bench.py wino 2>&1 | grep -v amdgpu
EMSA_BENCH_N=8 EMSA_BENCH_HOT=1 EMSA_BENCH_SHAPE="c64" timeout 300 python tools/conv_bench.py wino 2>&1 | grep -v amdgpu
