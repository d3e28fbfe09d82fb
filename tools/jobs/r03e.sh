#!/bin/bash
echo "== HIP-level memset-node probe"; timeout 120 tools/bin/repro_memset 2>&1 | tail -25
echo "== torch-level reduction probe"; timeout 300 python tools/graph_torch_reduce_repro.py 2>&1 | grep -v "amdgpu.ids"
