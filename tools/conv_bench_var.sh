#!/bin/bash
# usage: conv_bench_var.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...   (GPU box)
set -e
cd "$(dirname "$0")/.."
i=0
for extra in "$@"; do
  i=$((i+1))
  make -s -C emsanet_amd/csrc OUT=/tmp/var$i/libemsanet_hip.so OBJDIR=/tmp/var$i EXTRA="$extra" >/dev/null
  echo "== $extra"
  EMSA_LIB=/tmp/var$i/libemsanet_hip.so python tools/conv_bench.py ${WHAT:-all} -1 2>&1 | grep -v amdgpu | cut -c1-190
done
