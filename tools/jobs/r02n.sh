#!/bin/bash
O=gpurun_out/r02n; mkdir -p $O
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_model16_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "nbt1d or pinned_gradients_small or hipgraph or wgrad" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for dt in f32 bf16; do
  timeout 600 python bench.py --dtype $dt --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_$dt.json 2> $O/bench_$dt.err
  python - <<PY
import json
d = json.loads(open('$O/bench_$dt.json').read().strip().splitlines()[-1])
print('$dt', d['value'], d['ms_per_step'])
PY
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_f32 -o p --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/$O/prof_f32.log 2>&1
cd $R; find $O -name "*kernel_trace*" -delete
grep -i "reduce" $O/prof_f32/p_kernel_stats.csv | cut -c1-160
