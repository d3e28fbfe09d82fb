#!/bin/bash
O=gpurun_out/r03s; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/f32_driver.json 2>$O/f32_driver.err; python -c "
import json; d=json.loads(open('$O/f32_driver.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'], indent=1)); print(json.dumps(d['roofline_by_class'])); print(d['cpu_baseline'])"
timeout 900 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bf16.json 2>$O/bf16.err; python -c "
import json; d=json.loads(open('$O/bf16.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(json.dumps(d['roofline'], indent=1))"
timeout 900 python bench.py --force-dist --steps 10 --warmup 3 --no-cpu-baseline > $O/f32_fd.json 2>$O/f32_fd.err; python -c "
import json; d=json.loads(open('$O/f32_fd.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('measured_over'))"
