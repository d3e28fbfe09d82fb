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
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3     # us


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    tiles = [int(t) for t in sys.argv[2:]] or [-1]
    n = 32
    print(f"lib: {os.environ.get('EMSA_LIB', 'default')}")
    for name, cin, cout, k, s, p, h, w in SHAPES:
        spec = Fn.ConvSpec(cin, cout, k, s, p)
        x = Fn.act_empty(n, cin, h, w, DEV).normal_()
        oh, ow = spec.out_hw(h, w)
        dy = Fn.act_empty(n, cout, oh, ow, DEV).normal_()
        wt = torch.randn(cout, cin, *k, device=DEV) * 0.05
        wp, wpd = Fn.pack_weight(wt, 'fwd'), Fn.pack_weight(wt, 'dgrad')
        bias = torch.randn(cout, device=DEV)
        y = Fn.act_empty(n, cout, oh, ow, DEV)
        dx = Fn.act_empty(n, cin, h, w, DEV)
        flops = 2.0 * n * oh * ow * cin * cout * k[0] * k[1]
        row = f"{name:20s} {flops / 1e9:7.2f} GF |"
        for t in tiles:
            if t >= 0:
                os.environ['EMSA_CONV_TILE'] = str(t)
            else:
                os.environ.pop('EMSA_CONV_TILE', None)
            if what in ('fwd', 'all'):
                us = timeit(lambda: Fn.conv_fwd(x, wp, spec, bias=bias, act=1, out=y))
                row += f" fwd[t{t}] {us:7.1f}us {flops / us / 1e6:6.1f}TF |"
                us = timeit(lambda: Fn.conv_fwd(x, wp, spec, bias=bias, want_stats=True, out=y))
                row += f" +stats {flops / us / 1e6:6.1f}TF |"
            if what in ('dgrad', 'all'):
                us = timeit(lambda: Fn.conv_dgrad(dy, wpd, spec, (h, w), mask_src=x, out=dx))
                row += f" dgrad[t{t}] {us:7.1f}us {flops / us / 1e6:6.1f}TF |"
        if what in ('wino', 'all') and Fn.wino_eligible(spec):
            u, ud = Fn.pack_wino(wt, False), Fn.pack_wino(wt, True)
            us = timeit(lambda: Fn.conv_fwd(x, None, spec, bias=bias, act=1, out=y, wino_u=u))
            row += f" WINO fwd {us:7.1f}us {flops / us / 1e6:6.1f}TF(eff) |"
            us = timeit(lambda: Fn.conv_fwd(x, None, spec, bias=bias, want_stats=True, out=y,
                                            wino_u=u))
            row += f" +stats {flops / us / 1e6:6.1f} |"
            us = timeit(lambda: Fn.conv_dgrad(dy, None, spec, (h, w), mask_src=x, out=dx,
                                              wino_u=ud))
            row += f" dgrad {us:7.1f}us {flops / us / 1e6:6.1f} |"
        if what in ('wgrad', 'all'):
            us = timeit(lambda: Fn.conv_wgrad(x, dy, spec, True))
            row += f" wgrad {us:7.1f}us {flops / us / 1e6:6.1f}TF |"
        print(row, flush=True)


if __name__ == '__main__':
    main()
