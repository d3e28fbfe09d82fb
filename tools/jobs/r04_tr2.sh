#!/bin/bash
for s in "1x3 c128" "1x3 c512"; do EMSA_LIB=$PWD/tools/bin/whdbg/libemsanet_hip.so python tools/wgrad_phases.py "$s" 2>&1 | grep -v amdgpu.ids; done
for b in 768 1024 512 768; do echo "blocks $b"; EMSA_W1D_BLOCKS=$b bash tools/jobs/r04_wh3.sh base | tail -2; done
