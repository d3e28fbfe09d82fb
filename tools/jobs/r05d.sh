#!/bin/bash
# round 5, fourth GPU call: fused half-block (tests, configs[4] A/B), eval-BN gradient localisation
O=gpurun_out/r05d; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('$name', d['value'], d['ms_per_step'], r.get('frac'), (d.get('whole_step') or {}).get('hbm_frac'), (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
timeout 900 python -m pytest tests/test_conv_rs_gpu.py -k "half_block" -m gpu -q -x > $O/hb_ops.log 2>&1; echo "half-block op tests rc=$?"; tail -5 $O/hb_ops.log
timeout 900 python -m pytest tests/test_model16_gpu.py -k "half_blocks or twin_launches or hipgraph_inference" -m gpu -q -x > $O/hb_model.log 2>&1; echo "half-block model tests rc=$?"; tail -5 $O/hb_model.log
for dt in f16 bf16; do
  run config4_${dt}_half_block --dtype $dt --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline
  EMSA_HALF_BLOCK=0 run config4_${dt}_half_block_off --dtype $dt --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline
done
for bsz in 2 3; do
  run eval_b${bsz}_f16_hb_default --dtype f16 --eval --graph --batch-size $bsz --steps 200 --warmup 20 --no-cpu-baseline
  EMSA_HALF_BLOCK=1 run eval_b${bsz}_f16_hb_on --dtype f16 --eval --graph --batch-size $bsz --steps 200 --warmup 20 --no-cpu-baseline
  EMSA_HALF_BLOCK=0 run eval_b${bsz}_f16_hb_off --dtype f16 --eval --graph --batch-size $bsz --steps 200 --warmup 20 --no-cpu-baseline
done
