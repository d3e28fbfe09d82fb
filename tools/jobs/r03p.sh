#!/bin/bash
O=gpurun_out/r03p; mkdir -p $O; R=$GRAFT_REPO_ROOT
for i in 1 2; do
for v in default nt; do
  if [ $v = nt ]; then export EMSA_LIB=$R/emsanet_amd/lib/var_nt/libemsanet_hip.so; else unset EMSA_LIB; fi
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/f32_$v$i.json 2>$O/f32_$v$i.err; python -c "
import json; d=json.loads(open('$O/f32_$v$i.json').read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"
done; done
