#!/bin/bash
# hipGraph train step vs eager twin: which part of the warm-up restore exposes the discrepancy
for e in keep restore no_buf no_drop no_par; do
  echo "=== $e"; timeout 300 python tools/graph_step_debug.py $e 2>&1 | grep -v "^/opt\|amdgpu.ids" | head -60
done
echo "=== op tests of the folded BatchNorm + layout kernel"
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "folded or winograd" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "nbt1d_block or pinned_gradients_small" 2>&1 | tail -5
timeout 600 python tools/conv_bench.py inbn 2>&1 | grep -v "amdgpu.ids"
