#!/bin/bash
# multi-rank flow of bench.py as the driver launches it (gloo rehearsal: two ranks on ONE GPU), and
# the single-rank RCCL path, final code
O=gpurun_out/r04g; mkdir -p $O
EMSA_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 2 --warmup 1 --batch-size 2 --no-cpu-baseline --roofline-steps 1 > $O/two_rank_gloo.json 2>$O/two_rank.err; echo "rc=$?"; tail -c 600 $O/two_rank_gloo.json; echo
timeout 900 python bench.py --force-dist --steps 10 --warmup 3 --no-cpu-baseline > $O/forcedist.json 2>$O/fd.err; python -c "
import json; d=json.loads(open('$O/forcedist.json').read().strip().splitlines()[-1]); print('force-dist f32', d['value'], d['ms_per_step'], d['comm']['bucket_order'], d['comm']['exposed_comm_ms_per_step'])"
