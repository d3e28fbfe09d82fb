#!/bin/bash
O=gpurun_out/r02p; mkdir -p $O
for v in new old new_pf1 old_pf1 new2 old2; do
  case $v in
    new|new2) env_="" ;;
    old|old2) env_="EMSA_CONVH_RULE=old" ;;
    new_pf1) env_="EMSA_CONVH_PF=1" ;;
    old_pf1) env_="EMSA_CONVH_RULE=old EMSA_CONVH_PF=1" ;;
  esac
  env $env_ timeout 600 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
d = json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], [(k['kernel'][:14], k['avg_us'], k['launches']) for k in d['conv_kernels'][:2]])
PY
done
