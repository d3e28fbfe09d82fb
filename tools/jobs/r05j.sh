#!/bin/bash
# decoder-internal cut of the segmented step: tests + single-rank RCCL rehearsal
O=gpurun_out/r05j; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    c = d.get('comm') or {}
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'), c.get('bucket_launch_ms_before_backward_end'), c.get('exposed_comm_ms_per_step'), c.get('bucket_bytes'))
except Exception as e:
    print('$name failed', e)
PY
}
timeout 1200 python -m pytest tests/test_parallel_gpu.py -m gpu -q -x > $O/par.log 2>&1; echo "parallel tests rc=$?"; tail -4 $O/par.log; grep -h "segmented graph step" $O/par.log | cut -c1-400
run f32_forcedist_segmented --force-dist --graph --steps 20 --warmup 5 --no-cpu-baseline
run bf16_forcedist_segmented --dtype bf16 --force-dist --graph --steps 20 --warmup 5 --no-cpu-baseline
tail -3 $O/f32_forcedist_segmented.err
