#!/bin/bash
# round 6, first call: new tests (timed-size backward, sample pair, pack-plan freeze) + baseline bench lines
O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python -m pytest tests/test_timed_size_gpu.py tests/test_sample_pair.py -m gpu -q -x -s > $O/new_tests.log 2>&1; echo "new tests rc=$?"; tail -15 $O/new_tests.log
timeout 600 python -m pytest tests/test_model16_gpu.py -m gpu -q -x -k "pack_table or lean_pack" > $O/pack.log 2>&1; echo "pack rc=$?"; tail -5 $O/pack.log
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
run f32_1 --steps 20 --warmup 5 --no-cpu-baseline
run bf16_1 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
run f32_2 --steps 20 --warmup 5 --no-cpu-baseline
python tools/conv_bench16.py > $O/conv_bench16.txt 2>&1; tail -30 $O/conv_bench16.txt
python tools/conv_bench.py > $O/conv_bench.txt 2>&1; tail -30 $O/conv_bench.txt
