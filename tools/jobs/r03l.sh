#!/bin/bash
O=gpurun_out/r03l; mkdir -p $O
timeout 900 python -m pytest tests/test_ops16_gpu.py -x -q 2>&1 | grep -v "^  \|^$" | tail -8
timeout 900 python -m pytest tests/test_model16_gpu.py tests/test_model_gpu.py -x -q -k "bf16 or 16bit or train_step_variants or inference_with_postprocessing" 2>&1 | grep -v "^  \|^$" | tail -12
echo "=== wgrad 16-bit shapes, new modes vs EMSA_WGRAD16_MODES=0"
timeout 600 python tools/conv_bench16.py wgrad 2>&1 | grep -v amdgpu | tee $O/conv_bench16_wgrad.txt | grep -i "1x1\|s2\|wgrad" | head -30
EMSA_WGRAD16_MODES=0 timeout 600 python tools/conv_bench16.py wgrad 2>&1 | grep -v amdgpu | grep -i "1x1\|s2" | head
for m in 1 0; do EMSA_WGRAD16_MODES=$m timeout 900 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bf16_modes$m.json 2>$O/bf16_modes$m.err; python -c "
import json; d=json.loads(open('$O/bf16_modes$m.json').read().strip().splitlines()[-1]); print('modes=$m', d['value'], d['ms_per_step'])"; done
