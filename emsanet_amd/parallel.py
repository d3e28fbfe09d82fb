# -*- coding: utf-8 -*-
"""
Data-parallel training across the GPUs of one node: one process per GPU, RCCL over xGMI through
`torch.distributed` (backend "nccl" is RCCL on ROCm).

The reference has no multi-GPU path at all (/root/reference/main.py:82,443-448 -- single device;
SURVEY.md §0.2), so this is a new capability.  Per SURVEY.md §8(e) the path shards by batch: each
rank runs an independent bs/GPU shard (BatchNorm uses local statistics, as the reference's plain
`batchnorm` would), and the ONLY exchange is the gradient all-reduce.

`GradientBuckets` keeps every `param.grad` as a view into a few large flat buffers (32 MiB by
default: xGMI is point-to-point, 7 links x ~153 GB/s per GPU, so few large collectives beat many
small ones) and launches an asynchronous all-reduce for a bucket from the autograd
post-accumulate hook of the LAST gradient that lands in it -- in reverse layer order, so the
exchange overlaps the remaining backward pass.  `finish()` waits and averages.
"""
import torch
import torch.distributed as dist


class GradientBuckets:
    def __init__(self, params, bucket_bytes=32 << 20, process_group=None, average=True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.average = average
        params = [p for p in params if p.requires_grad]
        # reverse registration order ~ order in which backward produces gradients
        params = list(reversed(params))
        self.buckets = []          # (flat, [params])
        cur, cur_bytes = [], 0
        for p in params:
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
        for bi, (_, ps) in enumerate(self.buckets):
            for p in ps:
                self._bucket_of[p] = bi
                p.register_post_accumulate_grad_hook(self._hook)
        self.reset()

    def _close(self, ps):
        n = sum(p.numel() for p in ps)
        flat = torch.zeros(n, device=ps[0].device, dtype=ps[0].dtype)
        off = 0
        for p in ps:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.buckets.append((flat, ps))

    def reset(self):
        """call before every backward: zero the flat buffers (grads are views into them)"""
        for bi, (flat, ps) in enumerate(self.buckets):
            flat.zero_()
            self._pending[bi] = len(ps)
        self._handles = []

    def _hook(self, p):
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and self.world > 1:
            flat = self.buckets[bi][0]
            self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True))

    def finish(self):
        """call after backward: wait for the collectives; gradients become the world average"""
        if self.world > 1:
            # parameters that received no gradient this step leave their bucket un-reduced
            for bi, (flat, _) in enumerate(self.buckets):
                if self._pending[bi] != 0:
                    self._handles.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM,
                                                         group=self.group, async_op=True))
            for h in self._handles:
                h.wait()
            if self.average:
                for flat, _ in self.buckets:
                    flat.mul_(1.0 / self.world)
        self._handles = []

    def n_bytes(self):
        return sum(f.numel() * f.element_size() for f, _ in self.buckets)


def broadcast_parameters(module, src=0, process_group=None):
    """make every replica start from rank `src`'s parameters and buffers"""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=process_group)
