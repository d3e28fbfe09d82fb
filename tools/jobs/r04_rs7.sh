#!/bin/bash
O=gpurun_out/r04_rs7; mkdir -p $O
for pc in 2 1 2 1; do
EMSA_RS_PER_CU=$pc timeout 900 python bench.py --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 > $O/bf16g_pc$pc.json 2>$O/err.txt; python -c "
import json; d=json.loads(open('$O/bf16g_pc$pc.json').read().strip().splitlines()[-1]); print('per_cu=$pc graph', d['value'], d['ms_per_step'])"
done
