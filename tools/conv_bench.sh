#!/bin/bash
# GPU box: baseline + ablation builds of the conv kernels (EMSA_ABL: 1 no loads, 2 no stores, 4 no MFMA)
set -e
cd "$(dirname "$0")/.."
what=${1:-fwd}
python tools/conv_bench.py $what -1
for abl in 1 2 3 4 7; do
  make -s -C emsanet_amd/csrc OUT=/tmp/abl$abl/libemsanet_hip.so OBJDIR=/tmp/abl$abl EXTRA=-DEMSA_ABL=$abl >/dev/null
  echo "== ablation $abl"
  EMSA_LIB=/tmp/abl$abl/libemsanet_hip.so python tools/conv_bench.py $what -1
done
