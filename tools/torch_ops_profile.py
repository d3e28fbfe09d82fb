"""Which ATen ops (not engine kernels) still run inside one training step, by call count and
shape: finds stray copies / fills / adds issued by the Python glue.  Run on the GPU box:
    python tools/torch_ops_profile.py [--batch-size 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch-size', type=int, default=32)
    ap.add_argument('--stacks', action='store_true')
    a = ap.parse_args()
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets
    dev = torch.device('cuda', 0)
    model = EMSANet(full_args(), nyuv2_config())
    bench.deterministic_init_(model)
    model.to(dev).train()
    batch = bench.synthetic_batch_device(a.batch_size, 480, 640, 1234, dev)
    params = [p for p in model.parameters() if p.requires_grad]
    buckets = GradientBuckets(params)
    opt = FusedSGD(buckets, lr=1e-5)
    cots = None

    def step():
        nonlocal cots
        buckets.reset()
        flat = bench.flatten_outputs(model(batch))
        if cots is None:
            cots = [torch.randn_like(t) * 1e-3 for t in flat]
        torch.autograd.backward(flat, cots)
        buckets.finish()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True,
                 with_stack=a.stacks) as prof:
        step()
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=6 if a.stacks else 0)
    rows = [e for e in ka if e.key.startswith('aten::') and e.device_time_total > 0]
    rows.sort(key=lambda e: -e.count)
    print(f"{'op':34s} {'calls':>6s} {'dev us':>9s}  shapes")
    for e in rows[:40]:
        print(f"{e.key:34s} {e.count:6d} {e.device_time_total:9.0f}  {str(e.input_shapes)[:90]}")
        if a.stacks:
            for ln in e.stack[:6]:
                print('      ', ln)
    print('\nmemcpy / memset events:')
    n = {}
    for ev in prof.events():
        if 'Memcpy' in ev.name or 'Memset' in ev.name or 'copyBuffer' in ev.name \
                or 'fillBuffer' in ev.name:
            k = ev.name
            c, t = n.get(k, (0, 0.0))
            n[k] = (c + 1, t + ev.device_time)
    for k, (c, t) in sorted(n.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:50s} {c:6d} {t:9.0f} us")


if __name__ == '__main__':
    main()
