#!/bin/bash
O=gpurun_out/r04_rs8; mkdir -p $O
for rs in 0 1 0 1; do
EMSA_CONV_RS=$rs timeout 900 python bench.py --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 > $O/bf16g_rs$rs.json 2>$O/err.txt; python -c "
import json; d=json.loads(open('$O/bf16g_rs$rs.json').read().strip().splitlines()[-1]); print('rs=$rs graph', d['value'], d['ms_per_step'])"
done
timeout 600 tools/bin/conv_rs_probe 32 time 2>&1 | tail -18
