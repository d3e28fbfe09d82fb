#!/bin/bash
set -e
cd "$(dirname "$0")/.."
for sb in 0 1 2 3; do
  make -s -C emsanet_amd/csrc OUT=/tmp/sb$sb/libemsanet_hip.so OBJDIR=/tmp/sb$sb EXTRA=-DEMSA_SB=$sb >/dev/null
  echo "== EMSA_SB=$sb"
  EMSA_LIB=/tmp/sb$sb/libemsanet_hip.so python tools/conv_bench.py all -1 2>&1 | grep -v amdgpu | cut -c1-190
done
