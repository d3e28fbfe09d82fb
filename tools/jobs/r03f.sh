#!/bin/bash
F='^/opt\|amdgpu.ids\|UserWarning\|Consider using\|print(exp'
echo "=== restore PRE_SD=1 POST_SD=1"; HOOKS=none PRE_SD=1 POST_SD=1 timeout 300 python tools/graph_step_debug.py restore 2>&1 | grep -v "$F" | grep -v "output [0-9]* .*max diff 0.0" | head -40
echo "=== restore PRE_SD=0 POST_SD=0"; HOOKS=none timeout 300 python tools/graph_step_debug.py restore 2>&1 | grep -v "$F" | grep -v "output [0-9]* .*max diff 0.0" | head -40
