"""Micro-benchmark of the MFMA convolution kernels on the layer shapes of the bs=32 640x480
workload (GPU box).  usage: python tools/conv_bench.py [fwd|dgrad|wgrad|wino|all] [tile ...]
EMSA_LIB selects an alternative build (ablations)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emsanet_amd import functional as Fn      # noqa: E402

DEV = 'cuda:0'
# (name, cin, cout, kernel, stride, pad, h, w) at batch 32
SHAPES = [
    ('1x3 c64 /4', 64, 64, (1, 3), 1, (0, 1), 120, 160),
    ('3x1 c64 /4', 64, 64, (3, 1), 1, (1, 0), 120, 160),
    ('1x3 c128 /8', 128, 128, (1, 3), 1, (0, 1), 60, 80),
    ('3x1 c256 /16', 256, 256, (3, 1), 1, (1, 0), 30, 40),
    ('1x3 c512 /32', 512, 512, (1, 3), 1, (0, 1), 15, 20),
    ('3x3 512->512 /32', 512, 512, (3, 3), 1, (1, 1), 15, 20),
    ('3x3 256->128 /8', 256, 128, (3, 3), 1, (1, 1), 60, 80),
    ('3x3 128->40 /4', 128, 40, (3, 3), 1, (1, 1), 120, 160),
    # memory-bound 1x1 convs (decoder skip connections) and a stride-2 block conv
    ('1x1 c64 /4', 64, 64, (1, 1), 1, (0, 0), 120, 160),
    ('1x1 c128 /8', 128, 128, (1, 1), 1, (0, 0), 60, 80),
    ('1x1 c256 /16', 256, 256, (1, 1), 1, (0, 0), 30, 40),
    ('3x1 s2 64->128 /4', 64, 128, (3, 1), (2, 1), (1, 0), 120, 160),
]


def timeit(fn, iters=24):
    """fn(i): i selects the buffer set, so consecutive launches do not find their operands in
    the 256 MB Infinity Cache (as in the model, where every layer streams from HBM)"""
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3     # us


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    tiles = [int(t) for t in sys.argv[2:]] or [-1]
    n = int(os.environ.get('EMSA_BENCH_N', '32'))     # batch size
    print(f"lib: {os.environ.get('EMSA_LIB', 'default')}")
    only = os.environ.get('EMSA_BENCH_SHAPE')      # substring filter on the shape name
    for name, cin, cout, k, s, p, h, w in SHAPES:
        if only and only not in name:
            continue
        spec = Fn.ConvSpec(cin, cout, k, s, p)
        oh, ow = spec.out_hw(h, w)
        set_bytes = 4 * n * (2 * cin * h * w + 2 * cout * oh * ow)
        nb = 1 if os.environ.get('EMSA_BENCH_HOT') else max(2, -(-(768 << 20) // set_bytes))
        X = [Fn.act_empty(n, cin, h, w, DEV).normal_() for _ in range(nb)]
        DY = [Fn.act_empty(n, cout, oh, ow, DEV).normal_() for _ in range(nb)]
        Y = [Fn.act_empty(n, cout, oh, ow, DEV) for _ in range(nb)]
        DX = [Fn.act_empty(n, cin, h, w, DEV) for _ in range(nb)]
        wt = torch.randn(cout, cin, *k, device=DEV) * 0.05
        wp, wpd = Fn.pack_weight(wt, 'fwd'), Fn.pack_weight(wt, 'dgrad')
        bias = torch.randn(cout, device=DEV)
        flops = 2.0 * n * oh * ow * cin * cout * k[0] * k[1]
        row = f"{name:20s} {flops / 1e9:7.2f} GF |"
        for t in tiles:
            if t >= 0:
                os.environ['EMSA_CONV_TILE'] = str(t)
            else:
                os.environ.pop('EMSA_CONV_TILE', None)
            if what in ('fwd', 'all'):
                us = timeit(lambda i: Fn.conv_fwd(X[i % nb], wp, spec, bias=bias, act=1,
                                                  out=Y[i % nb]))
                row += f" fwd[t{t}] {us:7.1f}us {flops / us / 1e6:6.1f}TF |"
                us = timeit(lambda i: Fn.conv_fwd(X[i % nb], wp, spec, bias=bias, want_stats=True,
                                                  out=Y[i % nb]))
                row += f" +stats {flops / us / 1e6:6.1f}TF |"
            if what in ('dgrad', 'all'):
                us = timeit(lambda i: Fn.conv_dgrad(DY[i % nb], wpd, spec, (h, w),
                                                    mask_src=X[i % nb], out=DX[i % nb]))
                row += f" dgrad[t{t}] {us:7.1f}us {flops / us / 1e6:6.1f}TF |"
        if what in ('wino', 'all') and Fn.wino_eligible(spec):
            u, ud = Fn.pack_wino(wt, fwd=True, dgrad=True)
            us = timeit(lambda i: Fn.conv_fwd(X[i % nb], None, spec, bias=bias, act=1,
                                              out=Y[i % nb], wino_u=u))
            row += f" WINO fwd {us:7.1f}us {flops / us / 1e6:6.1f}TF(eff) |"
            us = timeit(lambda i: Fn.conv_fwd(X[i % nb], None, spec, bias=bias, want_stats=True,
                                              out=Y[i % nb], wino_u=u))
            row += f" +stats {flops / us / 1e6:6.1f} |"
            us = timeit(lambda i: Fn.conv_dgrad(DY[i % nb], None, spec, (h, w),
                                                mask_src=X[i % nb], out=DX[i % nb], wino_u=ud))
            row += f" dgrad {us:7.1f}us {flops / us / 1e6:6.1f} |"
        if what in ('inbn', 'all') and Fn.wino_eligible(spec) and Fn.wino_rows(spec) == 1:
            # the NBt1D block's bn1 folded into the loaders (forward, weight gradient) and into the
            # data gradient's epilogue, next to the unfolded launches they replace
            u, ud = Fn.pack_wino(wt, fwd=True, dgrad=True)
            aff = (torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV))
            mean, istd = torch.randn(cin, device=DEV), torch.rand(cin, device=DEV) + 0.5
            us0 = timeit(lambda i: Fn.conv_fwd(X[i % nb], None, spec, bias=bias, act=1,
                                               out=Y[i % nb], wino_u=u, want_relu_bits=True))
            us = timeit(lambda i: Fn.conv_fwd(X[i % nb], None, spec, bias=bias, act=1,
                                              out=Y[i % nb], wino_u=u, want_relu_bits=True,
                                              in_affine=aff))
            row += f" INBN fwd {us0:6.1f} -> {us:6.1f}us |"
            us0 = timeit(lambda i: Fn.conv_wgrad(X[i % nb], DY[i % nb], spec, True, like=wt,
                                                 two_pass=True))
            us = timeit(lambda i: Fn.conv_wgrad(X[i % nb], DY[i % nb], spec, True, like=wt,
                                                two_pass=True, in_affine=aff))
            row += f" wgrad {us0:6.1f} -> {us:6.1f}us |"
            us0 = timeit(lambda i: Fn.conv_dgrad(DY[i % nb], None, spec, (h, w), out=DX[i % nb],
                                                 wino_u=ud))
            us = timeit(lambda i: Fn.conv_dgrad_bnb(DY[i % nb], None, spec, (h, w), X[i % nb],
                                                    aff[0], aff[1], mean, istd, wino_u=ud))
            row += f" dgrad {us0:6.1f} -> bnb {us:6.1f}us |"
            pw = timeit(lambda i: Fn.bn_act(X[i % nb], aff[0], aff[1], None, None, 1,
                                            want_mask=True))
            row += f" (bn_act pass {pw:6.1f}us)"
        if what in ('wgrad', 'all'):
            us = timeit(lambda i: Fn.conv_wgrad(X[i % nb], DY[i % nb], spec, True))
            row += f" wgrad {us:7.1f}us {flops / us / 1e6:6.1f}TF |"
            us = timeit(lambda i: Fn.conv_wgrad(X[i % nb], DY[i % nb], spec, True, like=wt,
                                                two_pass=True))
            row += f" two-pass {us:7.1f}us |"
        print(row, flush=True)


if __name__ == '__main__':
    main()
