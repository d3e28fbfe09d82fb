"""Debug tool (GPU box): replay each backward primitive of an NBt1D block many times on fixed
inputs; report which primitive is not run-to-run reproducible."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from emsanet_amd import functional as Fn       # noqa: E402
from emsanet_amd import ops                     # noqa: E402
from emsanet_amd.nn import NonBottleneck1D      # noqa: E402
from util import rnd, to_act                    # noqa: E402


def maxdiff(a, b):
    return (a - b).abs().max().item() / max(1e-30, b.abs().max().item())


def main():
    c, n, h, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    reps = 30
    torch.manual_seed(0)
    blk = NonBottleneck1D(c, c).to('cuda:0').eval()
    rt = blk._rt
    x = to_act(rnd(n, c, h, w, seed=1))
    dy = to_act(rnd(n, c, h, w, seed=2))
    y3 = to_act(rnd(n, c, h, w, seed=3))
    spec = rt.c31_2.spec
    wt = blk.conv3x1_2.weight.detach()

    def stat(name, fn):
        ref = fn()
        torch.cuda.synchronize()
        worst = 0.0
        for _ in range(reps):
            junk = torch.randn(1 << 20, device='cuda:0')     # churn allocator
            out = fn()
            torch.cuda.synchronize()
            del junk
            ds = [maxdiff(o, r) for o, r in zip(out, ref)]
            worst = max(worst, max(ds))
        print(f"{name:40s} worst run-to-run rel diff {worst:.3e}")

    wpd = Fn.pack_weight(wt, 'dgrad')
    wpf = Fn.pack_weight(wt, 'fwd')
    stat('pack dgrad', lambda: (Fn.pack_weight(wt, 'dgrad'),))
    stat('conv fwd', lambda: (Fn.conv_fwd(x, wpf, spec),))
    stat('conv dgrad (fixed wpd)', lambda: (Fn.conv_dgrad(dy, wpd, spec, (h, w)),))
    stat('conv dgrad (fresh wpd)', lambda: (Fn.conv_dgrad(dy, Fn.pack_weight(wt, 'dgrad'), spec, (h, w)),))
    stat('conv dgrad mask', lambda: (Fn.conv_dgrad(dy, wpd, spec, (h, w), mask_src=y3),))
    stat('conv dgrad residual', lambda: (Fn.conv_dgrad(dy, wpd, spec, (h, w), residual=y3),))
    stat('conv wgrad', lambda: Fn.conv_wgrad(x, dy, spec, True))
    g = blk.bn1.weight.detach()
    s, t, inv = Fn.bn_fold(g, blk.bn1.bias.detach(), blk.bn1.running_mean, blk.bn1.running_var, 1e-3)
    yy = Fn.bn_act(x, s, t, None, None, 1)
    stat('bn_bwd', lambda: [t_ for t_ in Fn.bn_bwd(dy, yy, x, g, blk.bn1.running_mean, inv, None, 1, False, True)])
    stat('bn_bwd train', lambda: [t_ for t_ in Fn.bn_bwd(dy, yy, x, g, blk.bn1.running_mean, inv, None, 1, True, True)])

    def full():
        xx = x.detach().clone().requires_grad_(True)
        out = blk(xx)
        for p in blk.parameters():
            p.grad = None
        out.backward(dy)
        return [xx.grad] + [p.grad for p in blk.parameters()]
    stat('full block fwd+bwd', full)


if __name__ == '__main__':
    main()
