#!/bin/bash
# end-of-round evidence, second pass (after the BatchNorm-backward fusion default changed): tests, bf16 PMC
# traffic, bench lines, bf16 / f32 rocprofv3 tables
O=gpurun_out/r05zz; mkdir -p $O
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
( time timeout 2400 python -m pytest tests -m gpu -q --durations=15 ) > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests_gpu.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
tools/pmc_traffic2.sh bf16 --dtype bf16 > $O/pmc_bf16.log 2>&1; EMSA_PMC_BENCH_ARGS="--dtype bf16" python tools/pmc_traffic_json.py gpurun_out/pmc_bf16/raw.json $O/r05_pmc_traffic_bf16.json r05 > $O/pmc_bf16_json.log 2>&1; tail -3 $O/pmc_bf16_json.log
rm -rf gpurun_out/pmc_bf16/FETCH_SIZE gpurun_out/pmc_bf16/WRITE_SIZE
cp $O/r05_pmc_traffic_bf16.json profiles/ 2>/dev/null
run bench_driver_cmd_f32 --gpus 1 --steps 20 --warmup 5
run bench_bf16 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
run bench_bf16_graph --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_WGRAD_MULTI=0 run bench_bf16_graph_wgrad_multi_off --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_BN_FUSE=1 run bench_bf16_graph_bn_fuse_on --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
run bench_bf16_forcedist_segmented_graph --dtype bf16 --force-dist --graph --steps 20 --warmup 5 --no-cpu-baseline
run config3_r101_960x736_bs16_bf16 --dtype bf16 --backbone resnet101 --height 736 --width 960 --batch-size 16 --steps 10 --warmup 3 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bf16_one_stream -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_bf16_one.log 2>&1; echo "prof bf16 one stream rc=$?"
cd $R
find $O -name "*kernel_trace*" -delete
python tools/stats_csv_to_md.py $(ls $O/prof_bf16_one_stream/*kernel_stats.csv | head -1) 25 "r05_zz: EMSA_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline (bf16 storage, ONE stream; final sources and defaults of round 5)" > $O/bf16_one_stream_kernel_stats.md
rm -rf $O/prof_*/
head -16 $O/bf16_one_stream_kernel_stats.md | cut -c1-130
