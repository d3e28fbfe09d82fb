#!/bin/bash
O=gpurun_out/r06l; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
    c = d.get('comm') or {}
    print('   comm:', {k: c.get(k) for k in ('bucket_launch_ms_before_backward_end', 'conv_rs_cu_budget', 'exposed_ms_per_step', 'buckets')})
    rp = d.get('reference_protocol')
    if rp:
        for k in ('as_reference', 'pinned_raw', 'pinned_compact'):
            print('  ', k, {q: rp[k].get(q) for q in ('fps_mean', 'fps_std', 'ms_mean', 'reps_fps_mean', 'device_to_host_bytes')})
    cb = d.get('cpu_baseline')
    if cb: print('   cpu_baseline', cb.get('value'), cb.get('cores'), cb.get('host_saturating'))
except Exception as e:
    print('$name failed', e); print(open('$O/$name.err').read()[-2000:])
PY
}
export EMSA_DIST_BACKEND=gloo
run bf16_forcedist_cuts321 --dtype bf16 --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bf16_forcedist_cuts21 --dtype bf16 --force-dist --cut-stages 2,1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run f32_forcedist_cuts321 --force-dist --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
unset EMSA_DIST_BACKEND
run config4_f16_reference --eval --graph --batch-size 1 --dtype f16 --protocol reference --protocol-reps 5 --steps 80 --warmup 20 --no-cpu-baseline
run f32_driver --gpus 1 --steps 20 --warmup 5
