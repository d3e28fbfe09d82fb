#!/bin/bash
# up2x forward with hoisted loads: operator tests, one-stream kernel tables, step numbers
R=$(pwd); O=gpurun_out/r05t; mkdir -p $O
timeout 900 python -m pytest tests/test_ops16_gpu.py tests/test_ops_gpu.py tests/test_model16_gpu.py -k "up or ppm or twin or decoder" -m gpu -q -x > $O/up.log 2>&1; echo "up rc=$?"; tail -2 $O/up.log
cd /tmp && export TMPDIR=/tmp
for dt in bf16 f32; do
  if [ $dt = bf16 ]; then D="--dtype bf16"; else D=""; fi
  EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_$dt -o p --output-format csv -- python $R/bench.py $D --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_$dt.log 2>&1; echo "prof $dt rc=$?"
done
cd $R
for dt in bf16 f32; do
python tools/stats_csv_to_md.py $(ls $O/prof_$dt/*kernel_stats.csv | head -1) 25 "r05_t: EMSA_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py ($dt) --steps 20 --warmup 5 --no-cpu-baseline (ONE stream)" > $O/${dt}_one_stream_kernel_stats.md
head -4 $O/${dt}_one_stream_kernel_stats.md | tail -1 | cut -c1-150; grep "up2x" $O/${dt}_one_stream_kernel_stats.md | cut -c1-70,100-170
done
find $O -name "*kernel_trace*" -delete
run() { name=$1; shift; timeout 600 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    print('$name', d['value'], d['ms_per_step'], (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
run bf16_graph --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
run f32 --steps 20 --warmup 5 --no-cpu-baseline
run c4_f16 --dtype f16 --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline
