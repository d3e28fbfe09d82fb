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
of a bucket it launches an asynchronous all-reduce, so the exchange overlaps the remaining
backward pass; afterwards `param.grad` are views of the reduced buffer.  The backward kernels
write the large gradients (conv weights) STRAIGHT into their bucket views (`grad_target`), only
gradients that were produced elsewhere are gathered with one multi-tensor copy.  `finish()`
waits and averages (or leaves the sum for `FusedSGD`, which folds 1/world into its update).
`comm_dtype=torch.bfloat16` exchanges the buckets as bf16 (half the xGMI bytes).
With a single process nothing is copied or reduced at all.
"""
import torch
import torch.distributed as dist


# HIP streams the engine runs backward kernels on (the model's second stream for the depth encoder /
# instance decoder, and the stream the step was issued on).  A bucket can hold gradients written on
# either; before it goes on the wire the issuing stream waits for all of them (the collective only
# orders itself behind the stream that is current when it is issued).
_ENGINE_STREAMS = []


def register_stream(stream):
    if all(s.cuda_stream != stream.cuda_stream for s in _ENGINE_STREAMS):
        _ENGINE_STREAMS.append(stream)
        del _ENGINE_STREAMS[:-8]              # (bounded: a few long-lived streams)


def _join_engine_streams():
    if not _ENGINE_STREAMS or not torch.cuda.is_available():
        return
    cur = torch.cuda.current_stream()
    for s in _ENGINE_STREAMS:
        if s.cuda_stream != cur.cuda_stream:
            cur.wait_stream(s)


# CUs left free of the persistent conv_rs grid while several ranks train: RCCL's all-reduce kernel
# lives through a whole collective on up to this many workgroups (one per channel), and a
# 2-workgroups-per-CU persistent grid with a static tile partition would make either side queue
# behind the other.  EMSA_RS_CUS (total CUs for conv_rs) overrides;
# EMSA_RCCL_CUS sets the reserve (default 32: RCCL's channel count on a 7-link xGMI ring is <= 32).
def reserve_cus_for_collectives():
    """-> the CU count conv_rs now plans with (None: left alone)"""
    import os
    from . import functional as Fn
    if os.environ.get('EMSA_RS_CUS'):
        return int(os.environ['EMSA_RS_CUS'])
    reserve = int(os.environ.get('EMSA_RCCL_CUS', '32'))
    if reserve <= 0:
        return None
    total = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    return Fn.set_rs_cu_budget(max(8, total - reserve))


def rccl_env():
    """the RCCL / HSA settings that shape a multi-rank run, for the benchmark's JSON line"""
    import os
    keys = ('NCCL_MIN_NCHANNELS', 'NCCL_MAX_NCHANNELS', 'NCCL_PROTO', 'NCCL_ALGO', 'NCCL_BUFFSIZE',
            'NCCL_NTHREADS', 'NCCL_P2P_LEVEL', 'NCCL_IB_DISABLE', 'NCCL_DEBUG', 'RCCL_MSCCL_ENABLE',
            'RCCL_MSCCLPP_ENABLE', 'HSA_ENABLE_IPC_MODE_LEGACY', 'HSA_ENABLE_SDMA',
            'HSA_FORCE_FINE_GRAIN_PCIE', 'GPU_MAX_HW_QUEUES', 'EMSA_RS_CUS', 'EMSA_RCCL_CUS',
            'EMSA_DIST_BACKEND', 'OMP_NUM_THREADS')
    env = {k: os.environ[k] for k in keys if k in os.environ}
    ver = None
    try:
        v = torch.cuda.nccl.version()
        ver = '.'.join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:       # noqa: BLE001
        pass
    return {'set': env, 'unset_means_library_default': True, 'rccl_version': ver}


def grad_target(p):
    """The flat-bucket view this parameter's gradient should be written into by the backward
    kernel that produces it (so that neither the all-reduce nor the fused optimizer needs a
    gather copy) -- or None: no buckets, a gradient already exists (accumulation: autograd must
    add, so the kernel may not overwrite), or the view was already handed out since `reset()`."""
    slot = getattr(p, '_emsa_grad_slot', None)
    if slot is None or p.grad is not None:
        return None
    view, owner = slot
    if owner._taken.get(p) == owner._epoch:
        return None
    owner._taken[p] = owner._epoch
    # a FRESH alias of the bucket view: autograd's AccumulateGrad only adopts ("steals") a gradient
    # tensor nobody else references; handed the long-lived view object itself it clones it -- one
    # memcpy per parameter per step (measured: 742 x 4 us) and a second copy back into the bucket
    return view.detach()


class GradientBuckets:
    """usage per step:  buckets.reset(); loss.backward(); buckets.finish(); optimizer.step()

    Two backward passes between `reset()` and `finish()` (gradient accumulation) are an error when
    collectives are active: a bucket is all-reduced as soon as its last gradient of the FIRST pass
    exists, so a later gradient would silently stay local.  The hook raises in that case."""

    def __init__(self, params, bucket_bytes=32 << 20, process_group=None, average=True,
                 force_collectives=False, comm_dtype=None, order=None, tail_bytes=None,
                 groups=None, manual=False):
        """order: the parameters in the order their gradients ARRIVE in the backward pass (see
        `record_arrival_order`); default: reverse registration order, which is only approximately
        that.  tail_bytes: the LAST bucket (the first layers' gradients, which only exist when the
        backward pass ends -- its all-reduce cannot hide behind anything) is kept at most this
        large.  groups: lists of parameters that close a bucket at every group boundary (the
        backward SEGMENTS of `SegmentedGraphedTrainStep`: a bucket never straddles two captured
        graphs).  manual: no autograd hooks -- the caller gathers / reduces bucket by bucket
        (`gather`, `reduce`) at its own segment boundaries."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # force_collectives: run the gather + all-reduce path even with one rank (validation)
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())
        self.average = average
        self.comm_dtype = comm_dtype
        self.manual = manual
        params = [p for p in params if p.requires_grad]
        self.params = params
        if groups is not None and order is not None:
            raise ValueError("GradientBuckets: pass `order` OR `groups` (the groups fix the order)")
        if groups is None:
            # reverse registration order ~ order in which backward produces gradients
            groups = [list(order) if order is not None else list(reversed(params))]
        else:
            groups = [[p for p in g if p.requires_grad] for g in groups]
        listed = [p for g in groups for p in g]
        if len(listed) != len(params) or {id(p) for p in listed} != {id(p) for p in params}:
            raise ValueError("GradientBuckets: `order` / `groups` must list every parameter once")
        self.buckets = []          # (flat, [params], [views])
        self.group_buckets = []    # bucket indices per group
        for gi, g in enumerate(groups):
            first = len(self.buckets)
            size = lambda p: p.numel() * p.element_size()          # noqa: E731
            tail = []
            if tail_bytes is not None and gi == len(groups) - 1:
                # peel the small tail off the END of the arrival order
                acc = 0
                while g and acc + size(g[-1]) <= tail_bytes:
                    acc += size(g[-1])
                    tail.insert(0, g[-1])
                    g = g[:-1]
            cur, cur_bytes = [], 0
            for p in g:
                cur.append(p)
                cur_bytes += size(p)
                if cur_bytes >= bucket_bytes:
                    self._close(cur)
                    cur, cur_bytes = [], 0
            if cur:
                self._close(cur)
            if tail:
                self._close(tail)
            self.group_buckets.append(list(range(first, len(self.buckets))))
        nb = len(self.buckets)
        self._arrived = [0] * nb
        self._launched = [False] * nb
        self._handles = []
        self._comm = [None] * nb       # bf16 staging buffers (comm_dtype)
        self._bucket_of = {}
        self._epoch = 0
        self._taken = {}
        # evidence for bench.py: what was exchanged and how much of it was NOT hidden behind backward
        self.stats = {'steps': 0, 'collectives': 0, 'bytes': 0, 'gathered_tensors': 0,
                      'direct_tensors': 0}
        self._exposed = []             # (event before the wait, event after) per step
        for bi, (_, ps, views) in enumerate(self.buckets):
            for p, v in zip(ps, views):
                self._bucket_of[p] = bi
                p._emsa_grad_slot = (v, self)
                if self.active and not manual:
                    p.register_post_accumulate_grad_hook(self._hook)
        self.rs_cu_budget = None
        # (force_collectives: the one-rank rehearsal of the multi-rank step -- the collective kernels
        #  are resident there too, so the budget engages and the evidence file shows it)
        if self.active and params and params[0].is_cuda:
            self.rs_cu_budget = reserve_cus_for_collectives()
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
        for bi in range(len(self.buckets)):
            self._arrived[bi] = 0
            self._launched[bi] = False
        self._handles = []
        self._epoch += 1

    def begin_step_host(self):
        """host-side part of `reset()` for a step whose device work is replayed from a graph"""
        for bi in range(len(self.buckets)):
            self._arrived[bi] = 0
            self._launched[bi] = False
        self._handles = []
        self._epoch += 1

    def gather(self, bi):
        """bucket `bi` becomes the complete gradient of its parameters: gradients that were not
        written in place are copied in, parameters without a gradient contribute zeros; afterwards
        `param.grad` are the bucket views.  Plain device work on the current stream (capturable)."""
        flat, ps, views = self.buckets[bi]
        have = [(v, p.grad) for v, p in zip(views, ps) if p.grad is not None]
        if len(have) != len(ps):
            # (a parameter without a gradient this step contributes zeros; its stale bucket
            #  content must not be exchanged)
            missing = [v for v, p in zip(views, ps) if p.grad is None]
            torch._foreach_zero_(missing)
        stray = [(v, g) for v, g in have if g.data_ptr() != v.data_ptr()]
        if stray:
            torch._foreach_copy_([v for v, _ in stray], [g for _, g in stray])
        self.stats['gathered_tensors'] += len(stray)
        self.stats['direct_tensors'] += len(have) - len(stray)
        for v, p in zip(views, ps):
            p.grad = v                    # optimizer reads the (soon reduced) bucket view

    def reduce(self, bi):
        """asynchronous all-reduce (SUM) of bucket `bi` on the communication stream; `finish()`
        waits.  Not capturable: issued eagerly, between graph replays in the segmented step."""
        flat = self.buckets[bi][0]
        if flat.is_cuda:
            _join_engine_streams()
        buf = flat
        if self.comm_dtype is not None and self.comm_dtype != flat.dtype:
            if self._comm[bi] is None:
                self._comm[bi] = torch.empty_like(flat, dtype=self.comm_dtype)
            buf = self._comm[bi]
            buf.copy_(flat)
        self._handles.append((bi, dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group,
                                                  async_op=True)))
        self.stats['collectives'] += 1
        self.stats['bytes'] += buf.numel() * buf.element_size()
        self._launched[bi] = True

    def _launch(self, bi):
        if self.buckets[bi][0].is_cuda:
            _join_engine_streams()        # (the gather copies read gradients of either stream)
        self.gather(bi)
        self.reduce(bi)

    def _hook(self, p):
        # one asynchronous all-reduce per bucket, fired as soon as the bucket's last gradient
        # exists -> the exchange overlaps the remaining backward pass
        bi = self._bucket_of[p]
        if self._launched[bi]:
            raise RuntimeError(
                "GradientBuckets: a gradient arrived for a bucket that was already all-reduced in "
                "this step (second backward pass without reset()? gradient accumulation is not "
                "supported with overlapped collectives: call reset() before every backward)")
        self._arrived[bi] += 1
        if self._arrived[bi] == len(self.buckets[bi][1]):
            self._launch(bi)

    def finish(self):
        """call after backward: wait for the collectives; gradients become the world average
        (`average=True`) or stay the world SUM (`average=False`: FusedSGD folds 1/world in)"""
        if self.active:
            for bi in range(len(self.buckets)):
                if not self._launched[bi]:      # some parameter received no gradient this step
                    self._launch(bi)
            timed = torch.cuda.is_available() and self.buckets[0][0].is_cuda
            if timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
            for bi, h in self._handles:
                h.wait()
                if self._comm[bi] is not None:
                    self.buckets[bi][0].copy_(self._comm[bi])
            if self.average and self.world > 1:
                torch._foreach_mul_([f for f, _, _ in self.buckets], 1.0 / self.world)
            if timed:
                e1.record()
                self._exposed.append((e0, e1))
                if len(self._exposed) > 64:
                    self._exposed.pop(0)
            self.stats['steps'] += 1
        self._handles = []

    def exposed_comm_ms(self):
        """mean time per step the compute stream spent in finish() waiting for collectives that
        the backward pass did not hide (plus the bf16 copy-back / averaging kernels); call after
        a device synchronisation"""
        if not self._exposed:
            return None
        return sum(a.elapsed_time(b) for a, b in self._exposed) / len(self._exposed)

    def reset_stats(self):
        for k in self.stats:
            self.stats[k] = 0
        self._exposed = []

    def n_bytes(self):
        return sum(f.numel() * f.element_size() for f, _, _ in self.buckets)


def record_arrival_order(params, run_backward, process_group=None):
    """the order in which the gradients of `params` arrive during `run_backward()` (one ordinary
    forward + backward of the training step): temporary post-accumulate hooks log it.  Feed the
    result to `GradientBuckets(params, order=...)` so that bucket boundaries follow the measured
    arrival order instead of the registration order."""
    seen, handles = [], []
    for p in params:
        if p.requires_grad:
            handles.append(p.register_post_accumulate_grad_hook(lambda q: seen.append(q)))
    try:
        run_backward()
    finally:
        for h in handles:
            h.remove()
    got = {id(p) for p in seen}
    # parameters that received no gradient go last (they contribute zeros)
    order = seen + [p for p in params if p.requires_grad and id(p) not in got]
    return agree_on_order(params, order, process_group)


def agree_on_order(params, order, process_group=None, src=0):
    """every rank adopts rank `src`'s ordering of `params` (bucket layouts must be identical on all
    ranks; the hook order is deterministic today, this makes it so by construction)"""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return order
    params = [p for p in params if p.requires_grad]
    index = {id(p): i for i, p in enumerate(params)}
    dev = params[0].device if dist.get_backend(process_group) == 'nccl' else torch.device('cpu')
    idx = torch.tensor([index[id(p)] for p in order], dtype=torch.int64, device=dev)
    dist.broadcast(idx, src=src, group=process_group)
    got = idx.tolist()
    if sorted(got) != list(range(len(params))):
        raise RuntimeError("agree_on_order: rank %d did not list every parameter once" % src)
    return [params[i] for i in got]


def broadcast_parameters(module, src=0, process_group=None):
    """make every replica start from rank `src`'s parameters and buffers"""
    if not dist.is_initialized() or dist.get_world_size(process_group) == 1:
        return
    with torch.no_grad():
        params = list(module.parameters())
        # ONE flat broadcast per dtype instead of one per tensor (> 1,500 of them): pack, send,
        # unpack -- RCCL launch + rendezvous latency per call dominates tensors this small
        by_dtype = {}
        for t in params + list(module.buffers()):
            by_dtype.setdefault((t.dtype, t.device), []).append(t.detach())
        for (dtype, device), ts in by_dtype.items():
            flat = torch.empty(sum(t.numel() for t in ts), dtype=dtype, device=device)
            off = 0
            for t in ts:
                flat[off:off + t.numel()].copy_(t.reshape(-1))
                off += t.numel()
            dist.broadcast(flat, src=src, group=process_group)
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
        # the broadcast wrote in place through detached aliases: tell autograd -- and with it the
        # engine's packed-weight caches, which key on `_version` -- that the parameters changed
        torch.autograd.graph.increment_version(params)
