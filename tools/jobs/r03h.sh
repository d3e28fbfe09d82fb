#!/bin/bash
# memset-node surgery + folded bn1 + boundary tests; first bench / profile of the round
O=gpurun_out/r03h; mkdir -p $O
R=$GRAFT_REPO_ROOT
F='^/opt\|amdgpu.ids\|UserWarning\|Consider using\|print(exp\|parts_static ='
echo "=== graph debug PRE_SD=1 POST_SD=1 (memset nodes replaced)"
HOOKS=none PRE_SD=1 POST_SD=1 timeout 300 python tools/graph_step_debug.py restore 2>&1 | grep -v "$F" | grep -v "output [0-9]* .*max diff 0.0" | head -20
echo "=== graph tests"
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "hipgraph" 2>&1 | tail -4
echo "=== boundary tests"
timeout 1500 python -m pytest tests/test_boundary_gpu.py -x -q 2>&1 | tail -6
echo "=== model tests (folded bn1 default)"
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_golden_gpu.py -x -q -k "not hipgraph" 2>&1 | tail -4
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('$name', d['value'], d['ms_per_step'], r.get('frac'), json.dumps(d.get('roofline_by_class')))
except Exception as e:
    print('$name failed', e)
PY
}
run bench_f32_fold --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
EMSA_BN1_FOLD=0 run bench_f32_nofold --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
run bench_f32_fold_b --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_f32 -o p --output-format csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_f32.log 2>&1; echo "prof f32 rc=$?"
cd $R; find $O -name "*kernel_trace*" -delete; ls $O/prof_f32 | head
