#!/bin/bash
# micro-timings: BatchNorm finalize (one-launch rows form vs two-launch form), channel means, then the remaining GPU tests
O=gpurun_out/r04_bn1; mkdir -p $O
python - > $O/micro.txt 2>&1 <<'PY'
import sys, time, torch
sys.path.insert(0, '.')
from emsanet_amd import functional as Fn
dev = 'cuda:0'
def timeit(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for c in (64, 128, 256, 512):
    for rows in (150, 512, 1024, 1025, 4800):
        st = torch.rand(3, rows, c, device=dev) + 1
        g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        t = timeit(lambda: Fn.bn_finalize(st, rows * 100, g, b, 1e-3, 0.1, rm, rv))
        print(f"bn_finalize c={c:4d} rows={rows:5d}: {t:7.1f} us (host-paired)")
for dt in (torch.float32, torch.bfloat16):
    for (c, h, w) in ((64, 240, 320), (64, 120, 160), (128, 60, 80), (512, 15, 20)):
        x = Fn.act_empty(32, c, h, w, dev, dtype=dt).normal_()
        t = timeit(lambda: Fn.channel_mean(x), 50)
        print(f"channel_mean {dt} c={c} {h}x{w}: {t:7.1f} us  {x.numel() * x.element_size() / t / 1e6:5.2f} TB/s")
PY
cat $O/micro.txt
timeout 3000 python -m pytest tests/test_model16_gpu.py tests/test_conv_rs_gpu.py -x -q > $O/t1.log 2>&1; echo "model16+rs rc=$?"; tail -2 $O/t1.log
timeout 3300 python -m pytest tests -m gpu -x -q --deselect tests/test_model16_gpu.py --deselect tests/test_conv_rs_gpu.py -k "not test_boundary and not test_ops16 and not test_loss and not test_optim and not test_golden" > $O/t2.log 2>&1; echo "rest rc=$?"; tail -2 $O/t2.log
timeout 900 python bench.py --dtype bf16 --force-dist --graph --steps 20 --warmup 5 --no-cpu-baseline > $O/bf16_fd_graph.json 2>$O/bf16_fd_graph.err; echo "forcedist graph rc=$?"; python -c "
import json; d=json.loads(open('$O/bf16_fd_graph.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['comm']['path'], d['comm']['backend'], (d['roofline'] or {}).get('frac'))"
for i in 1 2; do timeout 900 python bench.py --dtype bf16 --graph --steps 20 --warmup 5 --no-cpu-baseline --roofline-steps 0 > $O/bf16g.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bf16g.json').read().strip().splitlines()[-1]); print('bf16 graph', d['value'], d['ms_per_step'])"; done
