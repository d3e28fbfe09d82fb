"""Per-stage cost / saving of the 16-bit bn1 fold (round 6): bf16, bs 32, the four
NBt1D stage shapes of the 640x480 workload, cold operands (a ring of tensor sets larger than the L2s).
  saved:  bn_act_fwd of bn1 (normalise + ReLU + bit mask, the pass that leaves the step)
  costs:  conv3x1 forward with the loader fold vs plain; single weight gradient with the loader fold vs
          plain; bn1 backward (reduce + sum + apply) with recomputed decisions vs the bit mask
usage: python tools/bn1_fold16_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emsanet_amd import functional as Fn      # noqa: E402


def timeit(fn, iters=24):
    """kernel time per call, us: the calls are captured in a hipGraph (the eager host path costs 20-40 us
    per call, more than the /16 and /32 kernels take) and one replay is timed"""
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            keep = [fn(i) for i in range(iters)]
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    del keep
    return a.elapsed_time(b) / iters * 1e3


DEV = 'cuda:0'
DT = torch.bfloat16


def main():
    n = 32
    print(f"{'stage':12s} {'bn_act pass':>11s} | {'fwd plain':>9s} {'folded':>7s} | {'wgrad plain':>11s} {'folded':>7s} | "
          f"{'bn bwd bits':>11s} {'recomputed':>10s} | net us per block")
    for name, c, h, w in (('c64 /4', 64, 120, 160), ('c128 /8', 128, 60, 80), ('c256 /16', 256, 30, 40),
                          ('c512 /32', 512, 15, 20)):
        spec = Fn.ConvSpec(c, c, (3, 1), (1, 1), (1, 0))
        set_bytes = 2 * n * 3 * c * h * w
        nb = max(2, -(-(768 << 20) // set_bytes))
        Y2 = [Fn.act_empty(n, c, h, w, DEV, dtype=DT).normal_() for _ in range(nb)]
        DY = [Fn.act_empty(n, c, h, w, DEV, dtype=DT).normal_() for _ in range(nb)]
        OUT = [Fn.act_empty(n, c, h, w, DEV, dtype=DT) for _ in range(nb)]
        wt = torch.randn(c, c, 3, 1, device=DEV) * 0.05
        wp = Fn.pack_weight_t(wt, DT, fwd=True, dgrad=False)[0]
        wf = Fn.pack_weight_frag_t(wt, DT, fwd=True, dgrad=False)[0]
        bias = torch.zeros(c, device=DEV)
        scale = torch.rand(c, device=DEV) + 0.5
        shift = torch.randn(c, device=DEV) * 0.2
        mean = torch.randn(c, device=DEV) * 0.1
        invstd = torch.rand(c, device=DEV) + 0.5
        gamma = torch.rand(c, device=DEV) + 0.5
        aff = (scale, shift)
        A2, K1 = zip(*[Fn.bn_act(y, scale, shift, None, None, Fn.ACT_RELU, want_mask=True) for y in Y2])
        t_act = timeit(lambda i: Fn.bn_act(Y2[i % nb], scale, shift, None, None, Fn.ACT_RELU, want_mask=True))
        t_f0 = timeit(lambda i: Fn.conv_fwd(A2[i % nb], wp, spec, bias=bias, act=Fn.ACT_RELU, out=OUT[i % nb], wfrag=wf))
        t_f1 = timeit(lambda i: Fn.conv_fwd(Y2[i % nb], wp, spec, bias=bias, act=Fn.ACT_RELU, out=OUT[i % nb], wfrag=wf,
                                            in_affine=aff))
        t_w0 = timeit(lambda i: Fn.conv_wgrad(A2[i % nb], DY[i % nb], spec, True, like=wt))
        t_w1 = timeit(lambda i: Fn.conv_wgrad(Y2[i % nb], DY[i % nb], spec, True, like=wt, in_affine=aff))
        t_b0 = timeit(lambda i: Fn.bn_bwd(DY[i % nb], K1[i % nb], Y2[i % nb], gamma, mean, invstd, None, Fn.ACT_RELU,
                                          True, want_dres=False))
        t_b1 = timeit(lambda i: Fn.bn_bwd_aff(DY[i % nb], Y2[i % nb], gamma, mean, invstd, aff))
        net = t_act - (t_f1 - t_f0) - (t_w1 - t_w0) - (t_b1 - t_b0)
        print(f"{name:12s} {t_act:11.1f} | {t_f0:9.1f} {t_f1:7.1f} | {t_w0:11.1f} {t_w1:7.1f} | {t_b0:11.1f} {t_b1:10.1f} | "
              f"{net:+.1f}")


if __name__ == '__main__':
    main()
