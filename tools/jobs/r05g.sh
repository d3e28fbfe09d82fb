#!/bin/bash
# bf16 defaults re-examined after conv_rs + multi-job weight gradients: streams, workgroups per CU, dgrad phases
O=gpurun_out/r05g; mkdir -p $O
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
timeout 1200 python -m pytest tests/test_model16_gpu.py tests/test_parallel_gpu.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for rep in 1 2; do
run bf16_graph_default_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_DUAL_STREAM=0 run bf16_graph_one_stream_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_RS_PER_CU=1 run bf16_graph_rs_per_cu1_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_DGRAD_PHASES=1 run bf16_graph_dgrad_phases_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_UP2X_FUSED=0 run bf16_graph_up2x_two_pass_$rep --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
done
run bf16_eager --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
run bf16_forcedist_segmented --dtype bf16 --force-dist --graph --steps 20 --warmup 5 --no-cpu-baseline
