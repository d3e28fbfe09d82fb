#!/bin/bash
timeout 900 python -m pytest tests/test_ops16_gpu.py -x -q -k "wgrad" 2>&1 | tail -3
bash tools/jobs/r04_wh3.sh head pk64 base head pk64 base
EMSA_LIB=$PWD/tools/bin/whdbg/libemsanet_hip.so python tools/wgrad_phases.py "1x3 c128" 2>&1 | grep -v amdgpu.ids
