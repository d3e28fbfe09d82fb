#!/bin/bash
O=gpurun_out/r03i; mkdir -p $O
echo "=== segmented graph tests"
timeout 1200 python -m pytest tests/test_parallel_gpu.py -x -q -s 2>&1 | grep -v "amdgpu.ids" | tail -15
echo "=== bench: 2 ranks on one GPU through gloo, eager vs segmented graphs (bf16, small batch: host-bound)"
for mode in "" "--graph"; do
EMSA_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --dtype bf16 --batch-size 4 --steps 10 --warmup 3 --no-cpu-baseline $mode > $O/bench_gloo2$mode.json 2> $O/bench_gloo2$mode.err; tail -c 1500 $O/bench_gloo2$mode.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['comm']))" || tail -5 $O/bench_gloo2$mode.err
done
echo "=== single-rank RCCL, segmented graph vs eager (bf16 bs32)"
for mode in "" "--graph"; do
timeout 900 python bench.py --force-dist --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline $mode > $O/bench_fd$mode.json 2> $O/bench_fd$mode.err; python -c "
import sys, json
d = json.loads(open('$O/bench_fd$mode.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], json.dumps(d['comm']))" || tail -5 $O/bench_fd$mode.err
done
