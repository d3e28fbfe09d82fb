#!/bin/bash
# end-of-round evidence, second part: bf16 one-stream kernel table, configs[3] bench lines, segmented-graph single-rank lines
O=gpurun_out/r04zz2; mkdir -p $O
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 900 python bench.py "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d = json.loads(open('$O/$name.json').read().strip().splitlines()[-1])
    r = d.get('roofline') or {}
    print('$name', d['value'], d['ms_per_step'], r.get('frac'), (d.get('whole_step') or {}).get('hbm_frac'), (d.get('comm') or {}).get('path'))
except Exception as e:
    print('$name failed', e)
PY
}
run config3_r101_960x736_bs16_f32 --backbone resnet101 --height 736 --width 960 --batch-size 16 --steps 6 --warmup 2 --no-cpu-baseline
run config3_r101_960x736_bs16_bf16 --dtype bf16 --backbone resnet101 --height 736 --width 960 --batch-size 16 --steps 6 --warmup 2 --no-cpu-baseline
run bench_bf16_forcedist_segmented_graph --dtype bf16 --force-dist --graph --steps 20 --warmup 5 --no-cpu-baseline
run bench_f32_forcedist_segmented_graph --force-dist --graph --steps 20 --warmup 5 --no-cpu-baseline
cd /tmp && export TMPDIR=/tmp
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bf16_one_stream -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_bf16_one.log 2>&1; echo "prof bf16 one stream rc=$?"
cd $R; find $O -name "*kernel_trace*" -delete
python tools/stats_csv_to_md.py $(ls $O/prof_bf16_one_stream/*kernel_stats.csv | head -1) 26 "r04_zz: EMSA_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline (bf16 storage, ONE stream; final sources of round 4; 26 passes = 20 timed + 5 warm-up + the byte-count pass)" > $O/bf16_one_stream_kernel_stats.md
rm -rf $O/prof_*/
