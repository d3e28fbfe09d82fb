#!/bin/bash
O=gpurun_out/r04_tr5; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops16_gpu.py tests/test_model16_gpu.py -x -q > $O/t.log 2>&1; echo "tests rc=$?"; tail -2 $O/t.log
bash tools/jobs/r04_wh3.sh base | tail -2
EMSA_WGRAD16_TR=0 bash tools/jobs/r04_wh3.sh base | tail -2
for i in 1 2; do
for tr in 1 0; do
  EMSA_WGRAD16_TR=$tr timeout 600 python bench.py --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --roofline-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tr=$tr graph', d['value'], d['ms_per_step'])"
done; done
