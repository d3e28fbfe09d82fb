"""Does a 16-bit conv launch's read phase overlap another launch's write phase?  The same forward
conv on ONE stream (back to back) and alternating on TWO streams (two launches in flight), us per
launch.  usage: python tools/conv_overlap_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emsanet_amd import functional as Fn      # noqa: E402
from tools.conv_bench import SHAPES            # noqa: E402

DEV, DT = 'cuda:0', torch.bfloat16


def main():
    n = 32
    side = torch.cuda.Stream()
    for name, cin, cout, k, s, p, h, w in SHAPES:
        if cout % 8 or s != 1:
            continue
        spec = Fn.ConvSpec(cin, cout, k, s, p)
        oh, ow = spec.out_hw(h, w)
        set_bytes = 2 * n * (cin * h * w + cout * oh * ow)
        nb = max(4, -(-(768 << 20) // set_bytes))
        X = [Fn.act_empty(n, cin, h, w, DEV, dtype=DT).normal_() for _ in range(nb)]
        out = [Fn.act_empty(n, cout, oh, ow, DEV, dtype=DT) for _ in range(nb)]
        wt = torch.randn(cout, cin, *k, device=DEV) * 0.05
        wp, _ = Fn.pack_weight_t(wt, DT, fwd=True, dgrad=False)
        bias = torch.zeros(cout, device=DEV)

        def launch(i):
            Fn.conv_fwd(X[i % nb], wp, spec, bias=bias, act=Fn.ACT_RELU, out=out[i % nb])

        def timed(two):
            cur = torch.cuda.current_stream()
            for i in range(4):
                launch(i)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            iters = 48
            a.record()
            side.wait_stream(cur)
            for i in range(iters):
                if two and i % 2:
                    with torch.cuda.stream(side):
                        launch(i)
                else:
                    launch(i)
            cur.wait_stream(side)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / iters * 1e3
        t1, t2 = timed(False), timed(True)
        # the SAME work as one launch, issued as two half-batch launches on two streams at once
        hn = n // 2
        Xh = [(x[:hn], x[hn:]) for x in X]
        Oh = [(o[:hn], o[hn:]) for o in out]

        def split(i):
            cur = torch.cuda.current_stream()
            side.wait_stream(cur)
            Fn.conv_fwd(Xh[i % nb][0], wp, spec, bias=bias, act=Fn.ACT_RELU, out=Oh[i % nb][0])
            with torch.cuda.stream(side):
                Fn.conv_fwd(Xh[i % nb][1], wp, spec, bias=bias, act=Fn.ACT_RELU, out=Oh[i % nb][1])
            cur.wait_stream(side)
        for i in range(4):
            split(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(24):
            split(i)
        b.record()
        torch.cuda.synchronize()
        t3 = a.elapsed_time(b) / 24 * 1e3
        print(f"{name:22s} one stream {t1:7.1f} us   two streams {t2:7.1f} us per launch ({t1 / t2:.2f}x)   "
              f"one conv as two concurrent half-batch launches {t3:7.1f} us ({t1 / t3:.2f}x)")


if __name__ == '__main__':
    main()
