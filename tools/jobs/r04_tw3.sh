#!/bin/bash
# twin launches, third step (up-sampling pairs, SE MLP rewrite): tests + batch-1 numbers + timeline
O=gpurun_out/r04tw3; mkdir -p $O
timeout 900 python -m pytest tests/test_conv_rs_gpu.py tests/test_postprocessing.py -m gpu -x -q -k "pair or softmax" > $O/tests_rs.log 2>&1; echo "tests pair rc=$?"; tail -3 $O/tests_rs.log
timeout 900 python -m pytest tests/test_model16_gpu.py -m gpu -x -q -k "twin or hipgraph_inference or eval_16bit" > $O/tests_twin.log 2>&1; echo "tests twin rc=$?"; tail -5 $O/tests_twin.log
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_ops16_gpu.py -m gpu -x -q -k "se or SE or channel or ppm" > $O/tests_se.log 2>&1; echo "tests se rc=$?"; tail -3 $O/tests_se.log
run() { name=$1; shift; timeout 600 python bench.py "$@" --no-cpu-baseline > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
for dt in f16 bf16; do
run b1_${dt}_default --eval --graph --batch-size 1 --dtype $dt --steps 300 --warmup 30
EMSA_TWIN=0 run b1_${dt}_twin0 --eval --graph --batch-size 1 --dtype $dt --steps 300 --warmup 30
done
run b1_f32_default --eval --graph --batch-size 1 --dtype f32 --steps 200 --warmup 30
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr -o p -- python $R/bench.py --eval --graph --batch-size 1 --dtype f16 --steps 60 --warmup 10 --no-cpu-baseline > $R/$O/bench_tr.json 2> $R/$O/bench_tr.err
cd $R
f=$(ls $O/tr/*kernel_trace.csv 2>/dev/null | head -1)
n=$(python -c "import json; d=json.loads(open('$O/bench_tr.json').read().strip().splitlines()[-1]); n=(d.get('hipgraph') or {}).get('nodes', 177); print(n[0] if isinstance(n, list) else n)")
python tools/graph_timeline.py $f $n 2 > $O/timeline_twin.txt 2>&1; tail -3 $O/timeline_twin.txt
rm -rf $O/tr
