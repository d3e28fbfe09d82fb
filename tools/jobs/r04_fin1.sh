#!/bin/bash
O=gpurun_out/r04_fin1; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bf16_one_stream -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_bf16_one.log 2>&1; echo "prof bf16 one stream rc=$?"
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_f32_one_stream -o p --output-format csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_f32_one.log 2>&1; echo "prof f32 one stream rc=$?"
cd $R; find $O -name "*kernel_trace*" -delete
python tools/stats_csv_to_md.py $(ls $O/prof_f32_one_stream/*kernel_stats.csv | head -1) 25 "r04_z: EMSA_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline (fp32, ONE stream: a launch's duration is its own)" > $O/f32_one_stream_kernel_stats.md
python tools/stats_csv_to_md.py $(ls $O/prof_bf16_one_stream/*kernel_stats.csv | head -1) 25 "r04_z: EMSA_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline (bf16 storage, ONE stream)" > $O/bf16_one_stream_kernel_stats.md
grep "bn_\|total GPU" $O/bf16_one_stream_kernel_stats.md | cut -c1-160
grep "bn_\|total GPU\|channel_dot" $O/f32_one_stream_kernel_stats.md | cut -c1-160
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver.json 2>$O/driver.err; python -c "
import json; d=json.loads(open('$O/driver.json').read().strip().splitlines()[-1]); r=d['roofline']; print('f32', d['value'], d['ms_per_step'], r['frac'], r.get('traffic'), d['cpu_baseline']['value'])"
tools/pmc_traffic2.sh f32 > $O/pmc_f32.log 2>&1; python tools/pmc_traffic_json.py gpurun_out/pmc_f32/raw.json $O/r04_pmc_traffic.json r04 > $O/pmc_f32_json.log 2>&1; tail -3 $O/pmc_f32_json.log
tools/pmc_traffic2.sh bf16 --dtype bf16 > $O/pmc_bf16.log 2>&1; EMSA_PMC_BENCH_ARGS="--dtype bf16" python tools/pmc_traffic_json.py gpurun_out/pmc_bf16/raw.json $O/r04_pmc_traffic_bf16.json r04 > $O/pmc_bf16_json.log 2>&1; tail -3 $O/pmc_bf16_json.log
rm -rf gpurun_out/pmc_f32/FETCH_SIZE gpurun_out/pmc_f32/WRITE_SIZE gpurun_out/pmc_bf16/FETCH_SIZE gpurun_out/pmc_bf16/WRITE_SIZE $O/prof_*/ gpurun_out/pmc_*/calib_*SIZE
