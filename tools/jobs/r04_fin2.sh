#!/bin/bash
# final sources: driver bench line, bf16 lines, one-stream kernel tables, PMC traffic
O=gpurun_out/r04_fin2; mkdir -p $O; R=$PWD
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
run bench_bf16_graph --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
EMSA_WGRAD16_TR=0 run bench_bf16_graph_wgrad_tr_off --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline
run bench_bf16 --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
run config3_r101_960x736_bs16_bf16 --dtype bf16 --backbone resnet101 --height 736 --width 960 --batch-size 16 --steps 6 --warmup 2 --no-cpu-baseline
for dt in bf16 f16; do
  run config4_eval_graph_bs1_$dt --dtype $dt --eval --graph --batch-size 1 --steps 300 --warmup 30 --no-cpu-baseline
done
cd /tmp && export TMPDIR=/tmp
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bf16_one_stream -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_bf16_one.log 2>&1; echo "prof bf16 one stream rc=$?"
cd $R; find $O -name "*kernel_trace*" -delete
python tools/stats_csv_to_md.py $(ls $O/prof_bf16_one_stream/*kernel_stats.csv | head -1) 25 "r04_z: EMSA_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline (bf16 storage, ONE stream; final sources of the round)" > $O/bf16_one_stream_kernel_stats.md
bash tools/jobs/r04_wh3.sh base > $O/wgrad16_kernel_trace.txt 2>&1
EMSA_WGRAD16_TR=0 bash tools/jobs/r04_wh3.sh base | tail -2 >> $O/wgrad16_kernel_trace.txt 2>&1
EMSA_LIB=$PWD/tools/bin/whdbg/libemsanet_hip.so python tools/wgrad_phases.py "1x3 c128" 2>&1 | grep -v amdgpu.ids > $O/wgrad16_phases.txt
EMSA_WGRAD16_TR=0 EMSA_LIB=$PWD/tools/bin/whdbg/libemsanet_hip.so python tools/wgrad_phases.py "1x3 c128" 2>&1 | grep -v amdgpu.ids >> $O/wgrad16_phases.txt
tools/pmc_traffic2.sh f32 > $O/pmc_f32.log 2>&1; python tools/pmc_traffic_json.py gpurun_out/pmc_f32/raw.json $O/r04_pmc_traffic.json r04 > $O/pmc_f32_json.log 2>&1; tail -3 $O/pmc_f32_json.log
tools/pmc_traffic2.sh bf16 --dtype bf16 > $O/pmc_bf16.log 2>&1; EMSA_PMC_BENCH_ARGS="--dtype bf16" python tools/pmc_traffic_json.py gpurun_out/pmc_bf16/raw.json $O/r04_pmc_traffic_bf16.json r04 > $O/pmc_bf16_json.log 2>&1; tail -8 $O/pmc_bf16_json.log
cp $O/r04_pmc_traffic.json $O/r04_pmc_traffic_bf16.json profiles/
run bench_driver_cmd_f32 --gpus 1 --steps 20 --warmup 5
run bench_bf16_with_traffic --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline
rm -rf gpurun_out/pmc_f32/FETCH_SIZE gpurun_out/pmc_f32/WRITE_SIZE gpurun_out/pmc_bf16/FETCH_SIZE gpurun_out/pmc_bf16/WRITE_SIZE $O/prof_*/ gpurun_out/pmc_*/calib_*SIZE
