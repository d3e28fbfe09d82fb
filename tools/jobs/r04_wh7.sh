#!/bin/bash
O=gpurun_out/r04_wh7; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops16_gpu.py tests/test_model16_gpu.py -x -q > $O/t.log 2>&1; echo "tests rc=$?"; tail -2 $O/t.log
for i in 1 2; do
for lib in head new; do
  L=$PWD/emsanet_amd/lib/libemsanet_hip.so; [ $lib = head ] && L=$PWD/tools/bin/head/libemsanet_hip.so
  EMSA_LIB=$L timeout 600 python bench.py --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --roofline-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib graph', d['value'], d['ms_per_step'])"
done; done
