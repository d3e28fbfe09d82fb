"""Micro-benchmark of the fused backward pass of the learned x2 up-sampling (emsa_up2x_dw3x3_bwd_t) at the
bs=32 640x480 shapes of the decoders (GPU box): us per launch, algorithmic TB/s = (dy + x read, dx written)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emsanet_amd import functional as Fn      # noqa: E402

DEV = 'cuda:0'


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
    n = 32
    #        c, h, w (input resolution), storage of x, storage of dy
    cases = [(512, 15, 20, 'bf16', 'bf16'), (256, 30, 40, 'bf16', 'bf16'), (128, 60, 80, 'bf16', 'bf16'),
             (40, 120, 160, 'bf16', 'bf16'), (40, 240, 320, 'bf16', 'f32'), (8, 120, 160, 'bf16', 'bf16'),
             (8, 240, 320, 'bf16', 'f32'),
             (512, 15, 20, 'f32', 'f32'), (256, 30, 40, 'f32', 'f32'), (128, 60, 80, 'f32', 'f32'),
             (40, 120, 160, 'f32', 'f32'), (40, 240, 320, 'f32', 'f32'), (8, 240, 320, 'f32', 'f32')]
    dts = {'bf16': torch.bfloat16, 'f32': torch.float32}
    tot = 0.0
    for c, h, w, tx, ty in cases:
        x = Fn.act_empty(n, c, h, w, DEV, dtype=dts[tx]).normal_()
        dy = Fn.act_empty(n, c, 2 * h, 2 * w, DEV, dtype=dts[ty]).normal_()
        wdw = torch.randn(c, 1, 3, 3, device=DEV)
        t = timeit(lambda: Fn.up2x_dw_bwd(dy, x, wdw))
        mb = (dy.numel() * dy.element_size() + 2 * x.numel() * x.element_size()) / 1e6
        tot += t
        print(f"c{c:<3} {h}x{w} x {tx} dy {ty}: {t:7.1f} us  {mb:7.0f} MB  {mb / t:5.2f} TB/s", flush=True)
    print(f"sum {tot:.0f} us")


if __name__ == '__main__':
    main()
