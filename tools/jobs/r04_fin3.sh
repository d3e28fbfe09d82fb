#!/bin/bash
# last evidence pass on the final sources: full GPU tests, smoke, PMC traffic, the driver's bench line, inference lines
O=gpurun_out/r04_fin3; mkdir -p $O; R=$PWD
timeout 2700 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests_gpu.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
tools/pmc_traffic2.sh f32 > $O/pmc_f32.log 2>&1; python tools/pmc_traffic_json.py gpurun_out/pmc_f32/raw.json $O/r04_pmc_traffic.json r04 > $O/pmc_f32_json.log 2>&1; tail -2 $O/pmc_f32_json.log
tools/pmc_traffic2.sh bf16 --dtype bf16 > $O/pmc_bf16.log 2>&1; EMSA_PMC_BENCH_ARGS="--dtype bf16" python tools/pmc_traffic_json.py gpurun_out/pmc_bf16/raw.json $O/r04_pmc_traffic_bf16.json r04 > $O/pmc_bf16_json.log 2>&1; tail -2 $O/pmc_bf16_json.log
cp $O/r04_pmc_traffic.json $O/r04_pmc_traffic_bf16.json profiles/
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('$name', d['value'], d['ms_per_step'], r.get('frac'), r.get('traffic'), (d.get('hipgraph') or {}).get('nodes'))
except Exception as e:
    print('$name failed', e)
PY
}
run bench_driver_cmd_f32 --gpus 1 --steps 20 --warmup 5
run bench_bf16_graph --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
for dt in f32 bf16 f16; do
  run config4_eval_graph_bs1_$dt --dtype $dt --eval --graph --batch-size 1 --steps 400 --warmup 40 --no-cpu-baseline
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_eval_bs1_f16 -o p --output-format csv -- python $R/bench.py --eval --graph --batch-size 1 --dtype f16 --steps 100 --warmup 10 --no-cpu-baseline > $R/$O/prof_eval.log 2>&1; echo "prof eval rc=$?"
cd $R; find $O -name "*kernel_trace*" -delete
python tools/stats_csv_to_md.py $(ls $O/prof_eval_bs1_f16/*kernel_stats.csv | head -1) 110 "r04_z: rocprofv3 --kernel-trace --stats -- python bench.py --eval --graph --batch-size 1 --dtype f16 --steps 100 --warmup 10 (configs[4]: whole-model hipGraph of 271 nodes, per forward; final sources)" > $O/eval_bs1_f16_kernel_stats.md
rm -rf gpurun_out/pmc_f32/FETCH_SIZE gpurun_out/pmc_f32/WRITE_SIZE gpurun_out/pmc_bf16/FETCH_SIZE gpurun_out/pmc_bf16/WRITE_SIZE $O/prof_*/ gpurun_out/pmc_*/calib_*SIZE
