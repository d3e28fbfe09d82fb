#!/bin/bash
# end-of-round check of the final code: full GPU suite, smoke, the driver's bench line, bf16 lines
O=gpurun_out/r04x; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.json 2>$O/driver.err; python -c "
import json; d=json.loads(open('$O/driver.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['in_timed_region']['frac'], d['cpu_baseline']['value'])"
timeout 900 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bf16.json 2>$O/bf16.err; python -c "
import json; d=json.loads(open('$O/bf16.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['in_timed_region']['frac'])"
timeout 900 python bench.py --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline > $O/bf16_graph.json 2>$O/bf16g.err; python -c "
import json; d=json.loads(open('$O/bf16_graph.json').read().strip().splitlines()[-1]); print('bf16 graph', d['value'], d['ms_per_step'])"
