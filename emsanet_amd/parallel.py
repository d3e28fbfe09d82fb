# -*- coding: utf-8 -*-
"""
Data-parallel training across the GPUs of one node: one process per GPU, RCCL over xGMI through
`torch.distributed` (backend "nccl" is RCCL on ROCm).

The reference has no multi-GPU path at all (/root/reference/main.py:82,443-448 -- single device;
SURVEY.md §0.2), so this is a new capability.  Per SURVEY.md §8(e) the path shards by batch: each
rank runs an independent bs/GPU shard (BatchNorm uses local statistics, as the reference's plain
`batchnorm` would), and the ONLY exchange is the gradient all-reduce.

`GradientBuckets` groups the parameters (reverse layer order) into a few large flat buffers
(32 MiB by default: xGMI is point-to-point, 7 links x ~153 GB/s per GPU, so few large
collectives beat many small ones).  From the autograd post-accumulate hook of the LAST gradient
of a bucket it gathers the bucket's gradients with one multi-tensor copy and launches an
asynchronous all-reduce, so the exchange overlaps the remaining backward pass; afterwards
`param.grad` are views of the reduced buffer.  `finish()` waits and averages.  With a single
process nothing is copied or reduced at all.
"""
import torch
import torch.distributed as dist


class GradientBuckets:
    """usage per step:  buckets.reset(); loss.backward(); buckets.finish(); optimizer.step()"""

    def __init__(self, params, bucket_bytes=32 << 20, process_group=None, average=True,
                 force_collectives=False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collectives: run the gather + all-reduce path even with one rank (validation)
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())
        self.average = average
        params = [p for p in params if p.requires_grad]
        self.params = params
        # reverse registration order ~ order in which backward produces gradients
        order = list(reversed(params))
        self.buckets = []          # (flat, [params], [views])
        cur, cur_bytes = [], 0
        for p in order:
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._bucket_of = {}
        for bi, (_, ps, _) in enumerate(self.buckets):
            for p in ps:
                self._bucket_of[p] = bi
                if self.active:
                    p.register_post_accumulate_grad_hook(self._hook)
        self.reset()

    ALIGN = 4        # elements: every tensor starts on a 16-byte boundary (float4 kernels read
    #                  parameters that FusedSGD re-homes into buffers of this layout)

    def _close(self, ps):
        pad = lambda k: (k + self.ALIGN - 1) // self.ALIGN * self.ALIGN      # noqa: E731
        n = sum(pad(p.numel()) for p in ps)
        flat = torch.zeros(n, device=ps[0].device, dtype=ps[0].dtype)
        views, off = [], 0
        for p in ps:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += pad(p.numel())
        self.buckets.append((flat, ps, views))

    def reset(self):
        """call before every backward.  Gradients are dropped (set to None) so that autograd
        ASSIGNS the freshly produced tensors instead of launching one `grad += new` kernel per
        parameter (742 tiny kernels per step for EMSANet)."""
        for p in self.params:
            p.grad = None
        for bi, (_, ps, _) in enumerate(self.buckets):
            self._pending[bi] = len(ps)
        self._handles = []

    def _launch(self, bi):
        flat, ps, views = self.buckets[bi]
        have = [(v, p.grad) for v, p in zip(views, ps) if p.grad is not None]
        if len(have) != len(ps):
            flat.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group,
                                             async_op=True))
        for v, p in zip(views, ps):
            p.grad = v                    # optimizer reads the (soon averaged) bucket view
        self._pending[bi] = -1

    def _hook(self, p):
        # one multi-tensor gather + one asynchronous all-reduce per bucket, fired as soon as the
        # bucket's last gradient exists -> the exchange overlaps the remaining backward pass
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def finish(self):
        """call after backward: wait for the collectives; gradients become the world average"""
        if self.active:
            for bi in range(len(self.buckets)):
                if self._pending[bi] >= 0:      # some parameter received no gradient this step
                    self._launch(bi)
            for h in self._handles:
                h.wait()
            if self.average and self.world > 1:
                torch._foreach_mul_([f for f, _, _ in self.buckets], 1.0 / self.world)
        self._handles = []

    def n_bytes(self):
        return sum(f.numel() * f.element_size() for f, _, _ in self.buckets)


def broadcast_parameters(module, src=0, process_group=None):
    """make every replica start from rank `src`'s parameters and buffers"""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
