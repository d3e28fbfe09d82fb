#!/bin/bash
O=gpurun_out/r04_sk1; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops16_gpu.py -x -q -k "tap_split or ppm or pool" > $O/t1.log 2>&1; echo "ops16 rc=$?"; tail -3 $O/t1.log
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -k "ppm" > $O/t2.log 2>&1; echo "ops ppm rc=$?"; tail -2 $O/t2.log
timeout 2400 python -m pytest tests/test_model16_gpu.py tests/test_model_gpu.py -x -q -k "hipgraph or eval" > $O/t3.log 2>&1; echo "model eval rc=$?"; tail -2 $O/t3.log
for dt in f16 bf16 f32; do
timeout 600 python bench.py --eval --graph --batch-size 1 --dtype $dt --steps 200 --warmup 20 --no-cpu-baseline > $O/eval_$dt.json 2>$O/eval_$dt.err; python -c "
import json; d=json.loads(open('$O/eval_$dt.json').read().strip().splitlines()[-1]); print('$dt graph bs1', d['value'], d['ms_per_step'])"
done
EMSA_CONVH_SPLITK=0 timeout 600 python bench.py --eval --graph --batch-size 1 --dtype f16 --steps 200 --warmup 20 --no-cpu-baseline > $O/eval_f16_nosplit.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/eval_f16_nosplit.json').read().strip().splitlines()[-1]); print('f16 graph bs1 no split', d['value'], d['ms_per_step'])"
