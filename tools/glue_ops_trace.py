"""Which lines of emsanet_amd issue torch glue ops (zeros / cat / clone / copy_ / zero_ / add) inside one
training step (GPU box):  python tools/glue_ops_trace.py [bf16|f32]"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

COUNTS = collections.Counter()
ON = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'emsanet_amd' in fr.filename:
            return f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:70]}"
    return 'outside emsanet_amd'


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        if ON[0]:
            COUNTS[(name, site())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets
    dev = torch.device('cuda', 0)
    model = EMSANet(full_args(), nyuv2_config())
    bench.deterministic_init_(model)
    model.to(dev).train()
    if kind == 'bf16':
        model.set_compute_dtype(torch.bfloat16)
    batch = bench.synthetic_batch_device(32, 480, 640, 1234, dev)
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
    for owner, names in ((torch, ('zeros', 'zeros_like', 'cat', 'empty_like')),
                         (torch.Tensor, ('zero_', 'new_zeros', 'clone', 'copy_', 'contiguous', 'fill_', 'float', 'to'))):
        for n in names:
            wrap(owner, n)
    ON[0] = True
    step()
    ON[0] = False
    torch.cuda.synchronize()
    for (name, where), c in sorted(COUNTS.items(), key=lambda kv: -kv[1]):
        print(f"{c:4d} {name:12s} {where}")


if __name__ == '__main__':
    main()
