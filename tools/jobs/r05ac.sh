#!/bin/bash
# lean 16-bit pack table (no [tap][n][k] operand for conv_rs convs unless somebody reads it): model tests + A/B
O=gpurun_out/r05ac; mkdir -p $O
timeout 1500 python -m pytest tests/test_model16_gpu.py tests/test_conv_rs_gpu.py -m gpu -x -q > $O/m16.log 2>&1; echo "m16 rc=$?"; tail -3 $O/m16.log
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e); print(open('$O/$name.err').read()[-800:])
PY
}
A="--dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline"
for rep in 1 2 3; do
run bf16_lean_$rep $A
EMSA_PACK_LEAN=0 run bf16_full_$rep $A
done
run bf16_eager_lean --dtype bf16 --eager --steps 20 --warmup 5 --no-cpu-baseline
run c4_f16 --dtype f16 --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p --output-format csv -- python $R/bench.py --dtype bf16 --eager --steps 10 --warmup 3 --no-cpu-baseline > $R/$O/prof.log 2>&1
grep -h "pack_batch" $R/$O/prof/*kernel_stats.csv | sed 's/"[^"]*"//' | head -3
find $R/$O -name "*kernel_trace*" -delete
