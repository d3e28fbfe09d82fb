"""Micro-benchmark of the 16-bit convolution kernels (conv_h.hip, bf16 weight gradients) on the
layer shapes of the bs=32 640x480 workload, with the HBM-bound time of each launch beside it
(algorithmic bytes = input + output (+ weights) once, at 6.3 TB/s achievable).
usage: python tools/conv_bench16.py [fwd|dgrad|wgrad|all]     EMSA_LIB selects another build"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emsanet_amd import functional as Fn      # noqa: E402
from tools.conv_bench import SHAPES, timeit    # noqa: E402

DEV = 'cuda:0'
DT = torch.bfloat16


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    n = int(os.environ.get('EMSA_BENCH_N', '32'))
    only = os.environ.get('EMSA_BENCH_SHAPE')
    print(f"lib: {os.environ.get('EMSA_LIB', 'default')}  batch {n}  tile {os.environ.get('EMSA_CONVH_TILE', 'auto')}")
    print(f"{'shape':22s} {'kind':6s} {'us':>8s} {'TFLOP/s':>8s} {'TB/s(algo)':>10s} {'hbm-bound us':>12s}")
    for name, cin, cout, k, s, p, h, w in SHAPES:
        if only and only not in name:
            continue
        if cout % 8:
            continue
        spec = Fn.ConvSpec(cin, cout, k, s, p)
        oh, ow = spec.out_hw(h, w)
        set_bytes = 2 * n * (2 * cin * h * w + 2 * cout * oh * ow)
        nb = max(2, -(-(768 << 20) // set_bytes))
        X = [Fn.act_empty(n, cin, h, w, DEV, dtype=DT).normal_() for _ in range(nb)]
        DY = [Fn.act_empty(n, cout, oh, ow, DEV, dtype=DT).normal_() for _ in range(nb)]
        wt = torch.randn(cout, cin, *k, device=DEV) * 0.05
        wp, wpd = Fn.pack_weight_t(wt, DT, fwd=True, dgrad=True)
        # the stride-1 3-tap 1-D convs run on conv_rs.hip with fragment-ordered weights (as in the model)
        wf, wfd = Fn.pack_weight_frag_t(wt, DT, fwd=True, dgrad=True) if Fn.rs_eligible(spec) else (None, None)
        bias = torch.zeros(cout, device=DEV)
        flops = 2.0 * n * oh * ow * cin * cout * k[0] * k[1]
        wbytes = wt.numel() * 2
        act_bytes = 2.0 * n * (cin * h * w + cout * oh * ow)
        kinds = ('fwd', 'dgrad', 'wgrad') if what == 'all' else (what,)
        for kind in kinds:
            if kind == 'fwd':
                out = [Fn.act_empty(n, cout, oh, ow, DEV, dtype=DT) for _ in range(nb)]
                t = timeit(lambda i: Fn.conv_fwd(X[i % nb], wp, spec, bias=bias, act=Fn.ACT_RELU,
                                                 out=out[i % nb], wfrag=wf))
                b = act_bytes + wbytes
            elif kind == 'dgrad':
                out = [Fn.act_empty(n, cin, h, w, DEV, dtype=DT) for _ in range(nb)]
                t = timeit(lambda i: Fn.conv_dgrad(DY[i % nb], wpd, spec, (h, w), out=out[i % nb], wfrag=wfd))
                b = act_bytes + wbytes
            else:
                t = timeit(lambda i: Fn.conv_wgrad(X[i % nb], DY[i % nb], spec, True, like=wt))
                b = act_bytes + 2 * wbytes
            kern = 'conv_rs' if (kind != 'wgrad' and wf is not None and Fn.CONV_RS) else ''
            print(f"{name:22s} {kind:6s} {t:8.1f} {flops / t / 1e6:8.1f} {b / t / 1e6:10.2f} "
                  f"{b / 6.3e6:12.1f}  {kern}")


if __name__ == '__main__':
    main()
