#!/bin/bash
# builds the product library, a debug library (phase timers) and the stand-alone conv_rs probe
set -e
cd "$(dirname "$0")/.."
make -C emsanet_amd/csrc -j8 2>&1 | grep -E "error|Error" || true
mkdir -p tools/bin/dbg
make -C emsanet_amd/csrc -j8 OUT=$PWD/tools/bin/dbg/libemsanet_hip.so OBJDIR=$PWD/tools/bin/dbg EXTRA=-DEMSA_RS_DBG=1 2>&1 | grep -E "error|Error" || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Iinclude tools/conv_rs_probe.hip -Lemsanet_amd/lib -lemsanet_hip -ldl \
  -Wl,-rpath,'$ORIGIN/../../emsanet_amd/lib' -o tools/bin/conv_rs_probe 2>&1 | grep -E "error" || true
ls -la tools/bin/conv_rs_probe tools/bin/dbg/libemsanet_hip.so emsanet_amd/lib/libemsanet_hip.so
