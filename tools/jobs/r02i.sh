#!/bin/bash
O=gpurun_out/r02i; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops16_gpu.py -m gpu -x -q -k "upsample" > $O/tests_up.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests_up.log
for dt in f32 bf16; do
  timeout 600 python bench.py --dtype $dt --no-cpu-baseline > $O/bench_$dt.json 2> $O/bench_$dt.err
  python - <<PY
import json
try:
    d = json.loads(open('$O/bench_$dt.json').read().strip().splitlines()[-1])
    print('$dt', d['value'], d['ms_per_step'])
except Exception as e:
    print('$dt failed', e)
PY
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bf16 -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/$O/prof_bf16.log 2>&1
cd $R; find $O -name "*kernel_trace*" -delete
grep -i "up2x" $O/prof_bf16/p_kernel_stats.csv | cut -c1-200
