#!/bin/bash
# env sweeps on one box: CU budget of the persistent conv_rs grid, workgroup budget of the multi-job
# weight gradients (bf16 graph-replayed step)
O=gpurun_out/r05k; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
A="--dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline"
run base_1 $A
for c in 240 224 208 192 160 128; do EMSA_RS_CUS=$c run rs_cus_$c $A; done
run base_2 $A
for b in 512 1024 1536; do EMSA_WGRAD_MULTI_WGS=$b run wgm_$b $A; done
EMSA_RS_PER_CU=1 run rs_per_cu_1 $A
run base_3 $A
for c in 224 192; do EMSA_RS_CUS=$c run rs_cus_${c}_b $A; done
