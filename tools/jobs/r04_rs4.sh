#!/bin/bash
# conv_rs inside the model: new op tests, the 16-bit model tests, bf16 bench A/B on one box
O=gpurun_out/r04_rs4; mkdir -p $O
timeout 1500 python -m pytest tests/test_conv_rs_gpu.py -x -q > $O/t_rs.log 2>&1; echo "rs tests rc=$?"; tail -3 $O/t_rs.log
timeout 2400 python -m pytest tests/test_model16_gpu.py -x -q > $O/t_m16.log 2>&1; echo "model16 rc=$?"; tail -3 $O/t_m16.log
for rs in 0 1; do
EMSA_CONV_RS=$rs timeout 900 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bf16_rs$rs.json 2>$O/bf16_rs$rs.err; python -c "
import json; d=json.loads(open('$O/bf16_rs$rs.json').read().strip().splitlines()[-1]); r=d['roofline']; print('rs=$rs eager', d['value'], d['ms_per_step'], r['frac'])"
EMSA_CONV_RS=$rs timeout 900 python bench.py --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline > $O/bf16g_rs$rs.json 2>$O/bf16g_rs$rs.err; python -c "
import json; d=json.loads(open('$O/bf16g_rs$rs.json').read().strip().splitlines()[-1]); print('rs=$rs graph', d['value'], d['ms_per_step'])"
done
