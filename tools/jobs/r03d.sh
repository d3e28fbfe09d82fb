#!/bin/bash
F='^/opt\|amdgpu.ids\|UserWarning\|Consider using\|print(exp'
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  echo "=== restore PRE_SD=$1 POST_SD=$2"; HOOKS=prealloc PRE_SD=$1 POST_SD=$2 timeout 300 python tools/graph_step_debug.py restore 2>&1 | grep -v "$F" | grep -v "output [0-9]* .*max diff 0.0" | head -40
done
echo "=== the test itself"
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -k "hipgraph_train" 2>&1 | tail -5
