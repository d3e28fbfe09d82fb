#!/bin/bash
O=gpurun_out/r02l; mkdir -p $O
EMSA_CONVH_PF=0 timeout 900 python -m pytest tests/test_ops16_gpu.py -m gpu -x -q -k "conv16 or stem16" > $O/tests_pf0.log 2>&1; echo "pf0 tests rc=$?"; tail -4 $O/tests_pf0.log
for pf in 1 0; do
  EMSA_CONVH_PF=$pf timeout 600 python tools/conv_bench16.py fwd > $O/cb16_fwd_pf$pf.txt 2>&1
  EMSA_CONVH_PF=$pf timeout 600 python tools/conv_bench16.py dgrad > $O/cb16_dgrad_pf$pf.txt 2>&1
done
paste <(awk '{print $1,$2,$3,$4}' $O/cb16_fwd_pf1.txt) <(awk '{print $4}' $O/cb16_fwd_pf0.txt) | column -t | head -40
for pf in 1 0; do
  EMSA_CONVH_PF=$pf timeout 600 python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_bf16_pf$pf.json 2> $O/bench_bf16_pf$pf.err
  python - <<PY
import json
d = json.loads(open('$O/bench_bf16_pf$pf.json').read().strip().splitlines()[-1])
print('pf$pf', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_us'])
PY
done
