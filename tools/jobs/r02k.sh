#!/bin/bash
O=gpurun_out/r02k; mkdir -p $O
for v in fuse nofuse fuse2 nofuse2; do
  case $v in
    fuse*) env_="" ;;
    nofuse*) env_="EMSA_BN_FUSE=0" ;;
  esac
  env $env_ timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_f32_$v.json 2> $O/bench_f32_$v.err
  python - <<PY
import json
d = json.loads(open('$O/bench_f32_$v.json').read().strip().splitlines()[-1])
print('$v', d['value'], d['ms_per_step'], [ (k['kernel'][:20], k['avg_us']) for k in d['conv_kernels'][:2]])
PY
done
