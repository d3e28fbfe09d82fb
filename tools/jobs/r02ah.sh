#!/bin/bash
O=gpurun_out/r02ah; mkdir -p $O
timeout 100 python bench.py --dtype bf16 --steps 20 --warmup 5 --h2d --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; tail -c 300 $O/bench_bf16.json; echo
timeout 150 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd_f32.json 2> $O/bench_f32.err; head -c 200 $O/bench_driver_cmd_f32.json; echo
