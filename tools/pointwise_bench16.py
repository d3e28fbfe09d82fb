"""Micro-benchmark of the BatchNorm passes of a training step at the bs=32 640x480 shapes, per storage
type (GPU box): us per launch and algorithmic TB/s = (tensors read + written once each) / time.

  python tools/pointwise_bench16.py [bf16|f32|f16 ...]

bn_act      y = relu(x * scale + shift) + the ReLU bit mask        (bn1 of an NBt1D block)
bn_act_res  y = relu((x * scale + shift) * drop + residual) + mask  (bn2)
reduce      emsa_bn_bwd_reduce_t (reads dy, x, the bit mask)
bwd         reduce + slice sums + apply (dy, x -> dx)               (bn1)
bwd_res     same with the residual gradient written as well          (bn2)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emsanet_amd import functional as Fn      # noqa: E402

DEV = 'cuda:0'
DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}


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
    return a.elapsed_time(b) / iters * 1e3


def main():
    kinds = [a for a in sys.argv[1:] if a in DT] or ['bf16', 'f32']
    n = 32
    for kind in kinds:
        dtype = DT[kind]
        print(f"# {kind}: us per call (algorithmic TB/s)")
        for c, h, w in ((64, 240, 320), (64, 120, 160), (128, 60, 80), (256, 30, 40), (512, 15, 20)):
            def act():
                return Fn.act_empty(n, c, h, w, DEV, dtype=dtype).normal_()
            x, dy, res = act(), act(), act()
            sc = torch.rand(c, device=DEV) + 0.5
            sh = torch.randn(c, device=DEV)
            mean, inv = torch.randn(c, device=DEV), torch.rand(c, device=DEV) + 0.5
            drop = (torch.rand(n, c, device=DEV) > 0.1).float()
            mb = x.numel() * x.element_size() / 1e6
            row = f"c{c:<3} {h}x{w} {mb:5.0f} MB"
            t = timeit(lambda: Fn.bn_act(x, sc, sh, None, None, 1, want_mask=True))
            row += f" | bn_act {t:6.1f} ({2 * mb / t:4.2f})"
            t = timeit(lambda: Fn.bn_act(x, sc, sh, drop, res, 1, want_mask=True))
            row += f" | bn_act_res {t:6.1f} ({3 * mb / t:4.2f})"
            _, bits = Fn.bn_act(x, sc, sh, None, None, 1, want_mask=True)
            L, p = Fn._lib.lib(), Fn._p
            rows = L.emsa_bn_bwd_rows(n * h * w, c)
            part = torch.empty((2, rows, c), device=DEV)
            t = timeit(lambda: L.emsa_bn_bwd_reduce_t(Fn.dt(x), p(dy), None, p(bits), p(x), p(mean), p(inv),
                                                       None, n, h * w, c, 1, p(part), Fn._stream()))
            row += f" | reduce {t:6.1f} ({2 * mb / t:4.2f})"
            t = timeit(lambda: Fn.bn_bwd(dy, bits, x, sc, mean, inv, None, 1, True, False))
            row += f" | bwd {t:6.1f} ({5 * mb / t:4.2f})"
            t = timeit(lambda: Fn.bn_bwd(dy, bits, x, sc, mean, inv, drop, 1, True, True))
            row += f" | bwd_res {t:6.1f} ({6 * mb / t:4.2f})"
            print(row, flush=True)


if __name__ == '__main__':
    main()
