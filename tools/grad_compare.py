"""Debug tool (GPU box): gradient error of engine and fp32 oracle against the fp64 oracle,
max-norm and L2-norm, for random and smooth cotangents."""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from emsanet_amd import full_args, nyuv2_config            # noqa: E402
from emsanet_amd.model import EMSANet                       # noqa: E402
from oracle.emsanet_oracle import (EMSANetOracle, deterministic_state_dict,   # noqa: E402
                                   synthetic_batch)
from util import rnd                                        # noqa: E402


def flatten(outs):
    flat = []
    for o, sides in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            flat += list(s) if isinstance(s, tuple) else [s]
    return flat


def main():
    mode, cot = sys.argv[1], sys.argv[2]
    h, w, bs = 96, 128, 4
    args = full_args(input_height=h, input_width=w)
    cfg = nyuv2_config()
    o32 = EMSANetOracle(args, cfg)
    sd = deterministic_state_dict(o32, 0)
    o32.load_state_dict(sd)
    o64 = copy.deepcopy(o32).double()
    model = EMSANet(args, cfg)
    model.load_state_dict(sd)
    model.to('cuda:0')
    batch = synthetic_batch(bs, h, w)
    for m in (model, o32, o64):
        m.train(mode == 'train')
        m.dropout_seed = 1234
    outs = [flatten(o32(batch)), flatten(o64({k: v.double() for k, v in batch.items()})),
            flatten(model({k: v.to('cuda:0') for k, v in batch.items()}))]
    if cot == 'random':
        cots = [rnd(*t.shape, seed=100 + i, scale=1e-1) for i, t in enumerate(outs[0])]
        for f, dev, dt in zip(outs, ('cpu', 'cpu', 'cuda:0'), (torch.float32, torch.float64, torch.float32)):
            torch.autograd.backward(f, [c.to(dev, dt) for c in cots])
    else:
        for f in outs:
            sum((t * t).mean() for t in f).backward()
    p32, p64 = dict(o32.named_parameters()), dict(o64.named_parameters())
    rows = []
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        r = p64[k].grad
        g, c = p.grad.detach().cpu().double(), p32[k].grad.double()
        dmax, dl2 = max(1e-30, r.abs().max().item()), max(1e-30, r.norm().item())
        rows.append((k, (g - r).abs().max().item() / dmax, (c - r).abs().max().item() / dmax,
                     (g - r).norm().item() / dl2, (c - r).norm().item() / dl2, dmax))
    a = np.array([r[1:5] for r in rows])
    print(f"{mode} {cot}: n={len(rows)}")
    for name, col in (('gpu max', 0), ('cpu max', 1), ('gpu l2', 2), ('cpu l2', 3)):
        v = a[:, col]
        print(f"  {name}: median {np.median(v):.2e}  p90 {np.percentile(v, 90):.2e}  max {v.max():.2e}")
    if os.environ.get('DUMP'):
        for r in rows:
            print(f"   {r[0]:75s} gpu_l2 {r[3]:.1e} cpu_l2 {r[4]:.1e}")
    rows.sort(key=lambda r: -r[3])
    for r in rows[:8]:
        print(f"   {r[0]:70s} gpu_max {r[1]:.1e} cpu_max {r[2]:.1e} gpu_l2 {r[3]:.1e} cpu_l2 {r[4]:.1e} |g|max {r[5]:.1e}")


if __name__ == '__main__':
    main()
