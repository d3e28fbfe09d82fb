#!/bin/bash
# bf16 one-stream kernel table with the fast BatchNorm passes + per-launch-shape breakdown of the BN kernels
R=$(pwd); O=gpurun_out/r05m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
EMSA_DUAL_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bf16 -o p --output-format csv rocpd -- python $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_bf16.log 2>&1; echo "prof bf16 rc=$?"
cd $R
python tools/stats_csv_to_md.py $(ls $O/prof_bf16/*kernel_stats.csv | head -1) 25 "r05_m: EMSA_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -- python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline (bf16, ONE stream; fast BatchNorm passes)" > $O/bf16_one_stream_kernel_stats.md
head -40 $O/bf16_one_stream_kernel_stats.md | cut -c1-150
DB=$(ls $O/prof_bf16/*.db | head -1)
for k in bn_act_fwd_fast bn_bwd_reduce_fast bn_bwd_apply_fast bn_bwd_sum bn_finalize_rows; do python tools/rocpd_by_grid.py $DB $k $O/by_grid_$k.md | head -30; done
rm -rf $O/prof_bf16/*.db
