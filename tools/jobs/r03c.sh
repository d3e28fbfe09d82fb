#!/bin/bash
for h in none clone prealloc; do
  echo "=== restore, HOOKS=$h"; HOOKS=$h timeout 300 python tools/graph_step_debug.py restore 2>&1 | grep -v "^/opt\|amdgpu.ids\|UserWarning\|Consider using\|print(exp" | grep -v "output [0-9]* .*max diff 0.0" | head -40
done
echo "=== prealloc depth 4"; HOOKS=prealloc HOOK_DEPTH=4 timeout 300 python tools/graph_step_debug.py restore 2>&1 | grep -v "^/opt\|amdgpu.ids\|UserWarning\|Consider using\|print(exp" | grep -v "output [0-9]* .*max diff 0.0" | head -60
echo "=== keep, HOOKS=none"; HOOKS=none timeout 300 python tools/graph_step_debug.py keep 2>&1 | grep "loss graph"
