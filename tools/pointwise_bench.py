"""Micro-benchmark of the HBM-bound kernels at the bs=32 640x480 shapes (GPU box): algorithmic
GB/s = (tensors read + written once each) / time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from emsanet_amd import functional as Fn      # noqa: E402

DEV = 'cuda:0'


def timeit(fn, iters=10):
    for _ in range(2):
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
    for c, h, w in ((64, 240, 320), (64, 120, 160), (128, 60, 80), (256, 30, 40), (512, 15, 20)):
        x = Fn.act_empty(n, c, h, w, DEV).normal_()
        y = Fn.act_empty(n, c, h, w, DEV).normal_()
        dy = Fn.act_empty(n, c, h, w, DEV).normal_()
        res = Fn.act_empty(n, c, h, w, DEV).normal_()
        sc = torch.rand(c, device=DEV) + 0.5
        sh = torch.randn(c, device=DEV)
        mean, inv = torch.randn(c, device=DEV), torch.rand(c, device=DEV) + 0.5
        drop = (torch.rand(n, c, device=DEV) > 0.1).float()
        mb = x.numel() * 4 / 1e6
        t = timeit(lambda: Fn.bn_act(x, sc, sh, None, None, 1))
        row = f"c{c} {h}x{w} ({mb:6.0f} MB/tensor) | bn_act {2 * mb / t:5.2f} TB/s"
        t = timeit(lambda: Fn.bn_act(x, sc, sh, drop, res, 1))
        row += f" | bn_act+drop+res {3 * mb / t:5.2f}"
        L = Fn._lib.lib()
        rows = L.emsa_bn_bwd_rows(n * h * w, c)
        part = torch.empty((2, rows, c), device=DEV)
        p = Fn._p
        t = timeit(lambda: L.emsa_bn_bwd_reduce(p(dy), p(y), None, p(x), p(mean), p(inv), None, n, h * w, c, 1, p(part), Fn._stream()))
        row += f" | bwd_reduce {3 * mb / t:5.2f}"
        t = timeit(lambda: Fn.bn_bwd(dy, y, x, sc, mean, inv, None, 1, True, True))
        row += f" | bwd(reduce+apply+dres) {8 * mb / t:5.2f}"
        _, bits = Fn.bn_act(x, sc, sh, None, None, 1, want_mask=True)
        t = timeit(lambda: Fn.bn_act(x, sc, sh, None, None, 1, want_mask=True))
        row += f" | bn_act+bits {2 * mb / t:5.2f} ({t:.0f}us)"
        t = timeit(lambda: Fn.bn_bwd(dy, bits, x, sc, mean, inv, None, 1, True, True))
        row += f" | bwd bits (6 tensors) {6 * mb / t:5.2f} ({t:.0f}us)"
        if h <= 120:
            wdw = torch.randn(c, 1, 3, 3, device=DEV)
            b = torch.randn(c, device=DEV)
            skip = Fn.act_empty(n, c, 2 * h, 2 * w, DEV).normal_()
            t = timeit(lambda: Fn.up2x_dw_fwd(x, wdw, b, skip))
            row += f" | up2x fwd {9 * mb / t:5.2f} ({t:.0f}us)"
            t = timeit(lambda: Fn.up2x_dw_bwd(skip, x, wdw))
            row += f" | up2x bwd(data+weight) {10 * mb / t:5.2f}"
        print(row, flush=True)
    # 40-channel full-resolution prediction upsampling
    x = Fn.act_empty(n, 40, 240, 320, DEV).normal_()
    wdw = torch.randn(40, 1, 3, 3, device=DEV)
    b = torch.randn(40, device=DEV)
    mb = x.numel() * 4 / 1e6
    t = timeit(lambda: Fn.up2x_dw_fwd(x, wdw, b, None))
    print(f"c40 240x320->480x640 up2x fwd {5 * mb / t:5.2f} TB/s ({t:.0f} us)")
    for cc, hh, ww in ((40, 120, 160), (8, 240, 320), (8, 120, 160)):
        xq = Fn.act_empty(n, cc, hh, ww, DEV).normal_()
        wq, bq_ = torch.randn(cc, 1, 3, 3, device=DEV), torch.randn(cc, device=DEV)
        mq = xq.numel() * 4 / 1e6
        t = timeit(lambda: Fn.up2x_dw_fwd(xq, wq, bq_, None))
        print(f"c{cc} {hh}x{ww} -> x2 up2x fwd {5 * mq / t:5.2f} TB/s ({t:.0f} us)")
    xs = Fn.act_empty(n, 64, 240, 320, DEV).normal_()
    t = timeit(lambda: Fn.maxpool_fwd(xs))
    print(f"maxpool fwd {1.25 * xs.numel() * 4 / 1e6 / t:5.2f} TB/s")
    t = timeit(lambda: Fn.channel_mean(xs))
    print(f"channel_mean {xs.numel() * 4 / 1e6 / t:5.2f} TB/s")
    a = torch.empty(xs.numel(), device=DEV)
    bq = torch.empty(xs.numel(), device=DEV)
    t = timeit(lambda: bq.copy_(a))
    print(f"torch copy (reference) {2 * xs.numel() * 4 / 1e6 / t:5.2f} TB/s")


if __name__ == '__main__' and len(sys.argv) == 1:
    main()


def loss_bench():
    from emsanet_amd.loss import CrossEntropyLossSemantic
    n = 32
    x = Fn.act_empty(n, 40, 480, 640, DEV).normal_().requires_grad_(True)
    t = torch.randint(0, 41, (n, 480, 640), device=DEV)
    crit = CrossEntropyLossSemantic(torch.rand(40) + 0.5).to(DEV)
    mb = x.numel() * 4 / 1e6
    tf = timeit(lambda: crit([x.detach()], [t]))
    loss = crit([x], [t])[0][0]
    tb = timeit(lambda: torch.autograd.grad(loss, x, retain_graph=True))
    print(f"semantic CE 40ch 480x640 bs32: fwd {tf:.0f} us ({(mb + t.numel() * 8 / 1e6) / tf:.2f} TB/s), "
          f"bwd {tb:.0f} us ({(2 * mb + t.numel() * 8 / 1e6) / tb:.2f} TB/s)")
    ref = torch.nn.CrossEntropyLoss(weight=crit.weights, ignore_index=-1)
    xr = x.detach().clone().requires_grad_(True)
    tt = timeit(lambda: ref(xr, t - 1))
    lr = ref(xr, t - 1)
    ttb = timeit(lambda: torch.autograd.grad(lr, xr, retain_graph=True))
    print(f"torch.nn.CrossEntropyLoss (reference of the same op): fwd {tt:.0f} us, bwd {ttb:.0f} us")


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'loss':
    loss_bench()
