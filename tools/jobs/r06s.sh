#!/bin/bash
# fp32 one-GPU step: eager (the default so far) vs the whole step replayed from ONE hipGraph, alternating x3 on one box
O=gpurun_out/r06s; mkdir -p $O
for i in 1 2 3; do
  for m in eager graph; do
    timeout 600 python bench.py --gpus 1 --$m --steps 20 --warmup 5 --no-cpu-baseline > $O/f32_${m}_$i.json 2> $O/f32_${m}_$i.err
    python -c "
import json; d=json.loads(open('$O/f32_${m}_$i.json').read().strip().splitlines()[-1]); print('$m', $i, d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'), (d.get('roofline') or {}).get('frac'))"
  done
done
