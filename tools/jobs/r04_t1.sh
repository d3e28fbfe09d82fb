#!/bin/bash
# new / changed tests of round 4 + driver-style bench lines
O=gpurun_out/r04_t1; mkdir -p $O
timeout 600 tools/bin/conv_rs_probe 32 check > $O/probe_check.txt 2>&1; echo "probe rc=$?"; tail -1 $O/probe_check.txt
timeout 1500 python -m pytest tests/test_conv_rs_gpu.py -x -q > $O/t_rs.log 2>&1; echo "rs tests rc=$?"; tail -2 $O/t_rs.log
timeout 3000 python -m pytest tests/test_model16_gpu.py -x -q -k "pinned or full_size" > $O/t_m16.log 2>&1; echo "model16 rc=$?"; tail -2 $O/t_m16.log; grep "bf16 train:" $O/t_m16.log | cut -c1-250
timeout 3000 python -m pytest tests/test_parallel_gpu.py -x -q > $O/t_par.log 2>&1; echo "parallel rc=$?"; tail -2 $O/t_par.log
timeout 1500 python -m pytest tests/test_ops_gpu.py -x -q -k "folded or bf16_mfma" > $O/t_ops.log 2>&1; echo "ops rc=$?"; tail -2 $O/t_ops.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.json 2>$O/driver.err; python -c "
import json; d=json.loads(open('$O/driver.json').read().strip().splitlines()[-1]); r=d['roofline']; print('f32', d['value'], d['ms_per_step'], r['frac'], d['cpu_baseline']['value'])"
