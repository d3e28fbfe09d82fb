#!/bin/bash
# round 5, second GPU call: multi-job weight gradients (tests + A/B), low-priority weight-gradient
# stream (A/B), BatchNorm statistics engine vs oracle (forward side of the bf16 gradient gain)
O=gpurun_out/r05b; mkdir -p $O
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
timeout 900 python -m pytest "tests/test_model_gpu.py::test_graph_memset_repair_handles_chained_memsets" tests/test_ops16_gpu.py::test_conv16_wgrad_multi -m gpu -q -x > $O/new_tests.log 2>&1; echo "new tests rc=$?"; tail -5 $O/new_tests.log
timeout 900 python -m pytest tests/test_model16_gpu.py tests/test_parallel_gpu.py -m gpu -q -x > $O/model16_tests.log 2>&1; echo "model16 tests rc=$?"; tail -3 $O/model16_tests.log
timeout 600 python tools/bn_stats_compare.py 256 320 8 --out $O/bn_stats_bf16_emul.txt > $O/bn_stats1.log 2>&1; echo "bn stats emul rc=$?"; head -30 $O/bn_stats_bf16_emul.txt
timeout 600 python tools/bn_stats_compare.py 256 320 8 --plain --out $O/bn_stats_bf16_plain.txt > $O/bn_stats2.log 2>&1; echo "bn stats plain rc=$?"
timeout 600 python tools/bn_stats_compare.py 256 320 8 --f32 --out $O/bn_stats_f32.txt > $O/bn_stats3.log 2>&1; echo "bn stats f32 rc=$?"
run bf16_graph_multi1 --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_WGRAD_MULTI=0 run bf16_graph_multi0 --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
run bf16_graph_multi1_b --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_WGRAD_STREAM=1 run bf16_graph_wstream1 --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_WGRAD_STREAM=2 run bf16_graph_wstream2 --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
run f32_default --steps 20 --warmup 5 --no-cpu-baseline
EMSA_WGRAD_STREAM=1 run f32_wstream1 --steps 20 --warmup 5 --no-cpu-baseline
EMSA_WGRAD_STREAM=2 run f32_wstream2 --steps 20 --warmup 5 --no-cpu-baseline
run f32_graph --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_WGRAD_STREAM=1 run f32_graph_wstream1 --graph --steps 20 --warmup 5 --no-cpu-baseline
run bf16_eager_multi1 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
timeout 900 python tools/oracle_threads.py > $O/oracle_threads.txt 2>&1; tail -20 $O/oracle_threads.txt
