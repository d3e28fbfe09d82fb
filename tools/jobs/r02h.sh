#!/bin/bash
# GPU job: full -m gpu suite, smoke, the driver's bench command, profiles of the fp32 and bf16 steps
O=gpurun_out/r02h; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests_gpu.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests_gpu.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_f32_driver.json 2> $O/bench_f32_driver.err; tail -c 400 $O/bench_f32_driver.json; echo
timeout 600 python bench.py --dtype bf16 --steps 20 --warmup 5 --h2d > $O/bench_bf16.json 2> $O/bench_bf16.err; tail -c 400 $O/bench_bf16.json; echo
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_f32 -o p --output-format csv -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_f32.log 2>&1; echo "prof f32 rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bf16 -o p --output-format csv -- python $R/bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $R/$O/prof_bf16.log 2>&1; echo "prof bf16 rc=$?"
cd $R
rm -f $O/prof_*/p_kernel_trace.csv $O/prof_*/*/p_kernel_trace.csv
find $O -name "*kernel_trace*" -delete
ls -R $O | head -30
