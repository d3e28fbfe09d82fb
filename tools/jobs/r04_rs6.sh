#!/bin/bash
O=gpurun_out/r04_rs6; mkdir -p $O
timeout 600 tools/bin/conv_rs_probe 32 all > $O/probe.txt 2>&1; echo "probe rc=$?"; tail -19 $O/probe.txt
timeout 1500 python -m pytest tests/test_conv_rs_gpu.py -x -q > $O/t_rs.log 2>&1; echo "rs tests rc=$?"; tail -2 $O/t_rs.log
timeout 2400 python -m pytest tests/test_model16_gpu.py -x -q > $O/t_m16.log 2>&1; echo "model16 rc=$?"; tail -2 $O/t_m16.log
for rs in 0 1; do
EMSA_CONV_RS=$rs timeout 900 python bench.py --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline > $O/bf16g_rs$rs.json 2>$O/bf16g_rs$rs.err; python -c "
import json; d=json.loads(open('$O/bf16g_rs$rs.json').read().strip().splitlines()[-1]); print('rs=$rs graph', d['value'], d['ms_per_step'])"
done
EMSA_CONV_RS=1 timeout 900 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bf16_rs1.json 2>$O/bf16_rs1.err; python -c "
import json; d=json.loads(open('$O/bf16_rs1.json').read().strip().splitlines()[-1]); r=d['roofline']; print('rs=1 eager', d['value'], d['ms_per_step'], r['frac'])"
