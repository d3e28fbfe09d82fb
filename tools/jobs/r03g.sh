#!/bin/bash
echo "== torch-level reduction probe"; timeout 300 python tools/graph_torch_reduce_repro.py 2>&1 | grep -v "amdgpu.ids"
