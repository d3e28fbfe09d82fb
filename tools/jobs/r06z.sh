#!/bin/bash
# end-of-round evidence on the final sources: tests, smoke, PMC traffic (+ calibration), bench lines of
# every BASELINE config, rocprofv3 tables
O=gpurun_out/r06z; mkdir -p $O   # (run three times this round: the last run is what profiles/r06_z_* hold)
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('$name', d['value'], d['ms_per_step'], r.get('frac'), r.get('traffic'), (d.get('whole_step') or {}).get('hbm_frac'), (d.get('hipgraph') or {}).get('nodes'), (d.get('cpu_baseline') or {}).get('value'))
except Exception as e:
    print('$name failed', e)
PY
}
if [ "$1" != "notests" ]; then
( time timeout 2400 python -m pytest tests -m gpu -q --durations=15 ) > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests_gpu.log
grep -h "eval-BN\|gradient-norm gain" $O/tests_gpu.log | cut -c1-600
fi
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
tools/pmc_traffic2.sh f32 > $O/pmc_f32.log 2>&1; python tools/pmc_traffic_json.py gpurun_out/pmc_f32/raw.json $O/r06_pmc_traffic.json r06 > $O/pmc_f32_json.log 2>&1; tail -4 $O/pmc_f32_json.log
tools/pmc_traffic2.sh bf16 --dtype bf16 > $O/pmc_bf16.log 2>&1; EMSA_PMC_BENCH_ARGS="--dtype bf16" python tools/pmc_traffic_json.py gpurun_out/pmc_bf16/raw.json $O/r06_pmc_traffic_bf16.json r06 > $O/pmc_bf16_json.log 2>&1; tail -4 $O/pmc_bf16_json.log
rm -rf gpurun_out/pmc_f32/FETCH_SIZE gpurun_out/pmc_f32/WRITE_SIZE gpurun_out/pmc_bf16/FETCH_SIZE gpurun_out/pmc_bf16/WRITE_SIZE
cp $O/r06_pmc_traffic.json $O/r06_pmc_traffic_bf16.json profiles/ 2>/dev/null
run bench_driver_cmd_f32 --gpus 1 --steps 20 --warmup 5
run bench_f32_eager --gpus 1 --eager --steps 20 --warmup 5 --no-cpu-baseline
run bench_bf16 --dtype bf16 --eager --steps 20 --warmup 5 --no-cpu-baseline
run bench_bf16_graph --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_BN1_FOLD16=0 run bench_bf16_graph_bn1_fold16_off --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_BN1_FOLD16=1 run bench_bf16_graph_bn1_fold16_everywhere --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
run bench_bf16_graph_2 --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_DIST_BACKEND=gloo run bench_f32_forcedist_segmented_graph --force-dist --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_DIST_BACKEND=gloo run bench_bf16_forcedist_segmented_graph --dtype bf16 --force-dist --graph --steps 20 --warmup 5 --no-cpu-baseline
for dt in f16 bf16 f32; do
  run config4_eval_graph_bs1_$dt --dtype $dt --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline
done
run config4_eval_graph_bs1_f16_reference_protocol --dtype f16 --eval --graph --batch-size 1 --steps 80 --warmup 20 --protocol reference --protocol-reps 5 --no-cpu-baseline
run config3_r101_960x736_bs16_f32 --backbone resnet101 --height 736 --width 960 --batch-size 16 --steps 10 --warmup 3 --no-cpu-baseline
run config3_r101_960x736_bs16_bf16 --dtype bf16 --backbone resnet101 --height 736 --width 960 --batch-size 16 --steps 10 --warmup 3 --no-cpu-baseline
run eval_bs32_f32 --eval --steps 20 --warmup 5 --no-cpu-baseline
run eval_bs32_bf16 --dtype bf16 --eval --steps 20 --warmup 5 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_f32_one_stream -o p --output-format csv -- python $R/bench.py --gpus 1 --eager --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_f32_one.log 2>&1; echo "prof f32 one stream rc=$?"
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bf16_one_stream -o p --output-format csv -- python $R/bench.py --dtype bf16 --eager --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_bf16_one.log 2>&1; echo "prof bf16 one stream rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_eval_bs1_f16 -o p --output-format csv -- python $R/bench.py --eval --graph --batch-size 1 --dtype f16 --steps 100 --warmup 10 --no-cpu-baseline > $R/$O/prof_eval.log 2>&1; echo "prof eval rc=$?"
cd $R
f=$(ls $O/prof_eval_bs1_f16/*kernel_trace.csv 2>/dev/null | head -1)
python tools/graph_timeline.py $f 155 2 > $O/eval_bs1_f16_twin_timeline.txt 2>&1; tail -3 $O/eval_bs1_f16_twin_timeline.txt
find $O -name "*kernel_trace*" -delete
python tools/stats_csv_to_md.py $(ls $O/prof_f32_one_stream/*kernel_stats.csv | head -1) 25 "r06_z: EMSA_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline (fp32, ONE stream: a launch's duration is its own; final sources of round 6)" > $O/f32_one_stream_kernel_stats.md
python tools/stats_csv_to_md.py $(ls $O/prof_bf16_one_stream/*kernel_stats.csv | head -1) 25 "r06_z: EMSA_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline (bf16 storage, ONE stream; final sources of round 6)" > $O/bf16_one_stream_kernel_stats.md
python tools/stats_csv_to_md.py $(ls $O/prof_eval_bs1_f16/*kernel_stats.csv | head -1) 110 "r06_z: rocprofv3 --kernel-trace --stats -- python bench.py --eval --graph --batch-size 1 --dtype f16 --steps 100 --warmup 10 (configs[4]: whole-model hipGraph with twin launches, per forward)" > $O/eval_bs1_f16_kernel_stats.md
rm -rf $O/prof_*/
