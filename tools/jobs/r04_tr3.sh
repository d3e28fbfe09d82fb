#!/bin/bash
timeout 900 python -m pytest tests/test_ops16_gpu.py -x -q -k "wgrad" 2>&1 | tail -3
for s in "1x3 c128"; do EMSA_LIB=$PWD/tools/bin/whdbg/libemsanet_hip.so python tools/wgrad_phases.py "$s" 2>&1 | grep -v amdgpu.ids; done
bash tools/jobs/r04_wh3.sh base | tail -2
EMSA_WGRAD16_TR=0 bash tools/jobs/r04_wh3.sh base | tail -2
bash tools/jobs/r04_wh3.sh base | tail -2
