"""Debug tool (GPU box): run the same fwd+bwd several times in one process; report the largest
run-to-run difference per gradient tensor and any NaN (use with EMSA_POISON=1)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from emsanet_amd import full_args, nyuv2_config            # noqa: E402
from emsanet_amd.model import EMSANet                       # noqa: E402
from oracle.emsanet_oracle import EMSANetOracle, deterministic_state_dict, synthetic_batch  # noqa
from util import rnd                                        # noqa: E402


def flatten(outs):
    flat = []
    for o, sides in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            flat += list(s) if isinstance(s, tuple) else [s]
    return flat


def main():
    mode = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    h, w, bs = 96, 128, 4
    args = full_args(input_height=h, input_width=w)
    cfg = nyuv2_config()
    o32 = EMSANetOracle(args, cfg)
    sd = deterministic_state_dict(o32, 0)
    model = EMSANet(args, cfg)
    model.load_state_dict(sd)
    model.to('cuda:0')
    model.train(mode == 'train')
    batch = {k: v.to('cuda:0') for k, v in synthetic_batch(bs, h, w).items()}
    from emsanet_amd import ops
    runs = []
    traces = []
    for r in range(reps):
        ops.TRACE = []
        model.dropout_step = 0
        for p in model.parameters():
            p.grad = None
        flat = flatten(model(batch))
        cots = [rnd(*t.shape, seed=100 + i, scale=1e-1).to('cuda:0') for i, t in enumerate(flat)]
        torch.autograd.backward(flat, cots)
        torch.cuda.synchronize()
        traces.append(ops.TRACE)
        runs.append(({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                     [t.detach().clone() for t in flat]))
        if r == 1:
            # churn the allocator so later runs see recycled (dirty) blocks
            junk = [torch.randn(1 << 22, device='cuda:0') for _ in range(16)]
            del junk
    t0 = traces[0]
    for r, tr in enumerate(traces[1:], 1):
        assert len(tr) == len(t0)
        shown = 0
        for i, ((ka, a), (kb, b)) in enumerate(zip(tr, t0)):
            d = (a - b).abs().max().item() / max(1e-30, b.abs().max().item())
            if d > 1e-5:
                print(f"  trace run{r} #{i} {ka} shape {tuple(a.shape)} rel diff {d:.3e}")
                shown += 1
                if shown >= 6:
                    break
    g0, o0 = runs[0]
    worst = []
    for r, (g, o) in enumerate(runs[1:], 1):
        for i, (a, b) in enumerate(zip(o, o0)):
            d = (a - b).abs().max().item() / max(1e-30, b.abs().max().item())
            if d > 0:
                worst.append((d, f'run{r} output{i}'))
        for k in g0:
            a, b = g[k], g0[k]
            if not torch.isfinite(a).all():
                worst.append((float('inf'), f'run{r} NaN {k}'))
                continue
            d = (a - b).abs().max().item() / max(1e-30, b.abs().max().item())
            worst.append((d, f'run{r} {k}'))
    nan0 = [k for k, v in g0.items() if not torch.isfinite(v).all()]
    nano = [i for i, t in enumerate(o0) if not torch.isfinite(t).all()]
    print(f"{mode}: NaN grads in run0: {nan0[:10]} ; NaN outputs: {nano}")
    worst.sort(key=lambda x: -x[0])
    for d, k in worst[:25]:
        print(f"  {d:.3e}  {k}")


if __name__ == '__main__':
    main()
