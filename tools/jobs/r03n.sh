#!/bin/bash
O=gpurun_out/r03n; mkdir -p $O
timeout 900 python -m pytest tests/test_ops16_gpu.py -x -q 2>&1 | grep -v "^  \|^$" | tail -6
timeout 1200 python -m pytest tests/test_model16_gpu.py -x -q 2>&1 | grep -v "^  \|^$" | tail -6
for m in 1 0; do EMSA_WGRAD16_MODES=$m timeout 900 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bf16_modes$m.json 2>$O/bf16_modes$m.err; python -c "
import json; d=json.loads(open('$O/bf16_modes$m.json').read().strip().splitlines()[-1]); print('modes=$m', d['value'], d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bf16 -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_bf16.log 2>&1; echo "prof bf16 rc=$?"
cd $R; find $O -name "*kernel_trace*" -delete
