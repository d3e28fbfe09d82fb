# -*- coding: utf-8 -*-
"""
Whole-model hipGraph capture for inference (BASELINE.json config 5: 640x480, bs=1).

The reference's low-latency path is ONNX -> TensorRT (`inference_time_whole_model.py:173-262,
350-453`); the MI355X-native analogue is to capture the eval-mode forward -- ~260 launches of
libemsanet_hip.so kernels, BatchNorm folded into the conv epilogues -- once into a hipGraph and
replay it: bs=1 is launch-bound when run eagerly (host ~10 ms per forward), the replay is
GPU-bound.  Kernels are captured through torch's stream capture (they are plain launches on
`torch.cuda.current_stream()`), activations live in the graph's private memory pool.
"""
import ctypes

import torch

from . import _lib
from .ops import mark_bn_stats_written


def _flatten(out):
    if torch.is_tensor(out):
        return [out]
    if isinstance(out, dict):
        return [t for v in out.values() for t in _flatten(v)]
    if isinstance(out, (list, tuple)):
        return [t for v in out for t in _flatten(v)]
    return []


# Stream captures run in THREAD-LOCAL error mode: with a process group alive, c10d's watchdog thread
# polls the events of earlier collectives (hipEventQuery) whenever it likes -- in the default global
# mode that call is "not permitted when stream is capturing" and takes the process down (seen in
# round 4: bench.py --dtype bf16 --force-dist --graph under RCCL).  The capturing thread itself only
# launches kernels and uses the caching allocator's capture-aware paths.
_CAPTURE_MODE = 'thread_local'


def _new_graph():
    """a CUDAGraph whose raw hipGraph_t survives the capture (torch >= 2.8: keep_graph) so that it
    can be repaired before it is instantiated"""
    try:
        return torch.cuda.CUDAGraph(keep_graph=True), True
    except TypeError:                                  # older torch: no access to the raw graph
        return torch.cuda.CUDAGraph(), False


def _repair_and_instantiate(graph, kept):
    """replace the capture's memset nodes by fill-kernel nodes (emsa_graph_replace_memsets: captured
    memsets -- torch's zero-fills of reduction semaphores / scratch -- corrupt replays when eager
    memsets run between them, ROCm 7.2) and instantiate.  -> {'nodes', 'memset_nodes', 'replaced'}"""
    info = {'nodes': None, 'memset_nodes': None, 'replaced': 0}
    if not kept:
        # no access to the raw graph (torch < 2.8): the capture cannot be repaired, and an
        # un-repaired capture with memset nodes is the configuration known to replay wrongly
        #  -- say so instead of running it silently
        import warnings
        warnings.warn("emsanet_amd.graph: this torch has no CUDAGraph(keep_graph=True); captured "
                      "memset nodes cannot be replaced by kernel nodes -- replays of a step that "
                      "zero-fills memory (device-side losses, gradient buffers) may be wrong on "
                      "ROCm 7.2", RuntimeWarning, stacklevel=2)
        return info
    raw = graph.raw_cuda_graph()
    raw = int(raw) if not hasattr(raw, 'value') else int(raw.value)
    L = _lib.lib()
    n, ms, kn, rep = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    _lib.check(L.emsa_graph_count_nodes(raw, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(kn)),
               'emsa_graph_count_nodes')
    _lib.check(L.emsa_graph_replace_memsets(raw, ctypes.byref(rep)), 'emsa_graph_replace_memsets')
    graph.instantiate()
    info.update(nodes=n.value, memset_nodes=ms.value, replaced=rep.value)
    return info


class GraphedInference:
    """model must be in eval mode; inputs must keep shape/dtype/device of `example_batch`."""

    def __init__(self, model, example_batch, do_postprocessing=False, warmup=3):
        if model.training:
            raise ValueError("GraphedInference captures the eval-mode forward")
        self.model = model
        self.do_postprocessing = do_postprocessing
        self.warmup = warmup
        self.static_in = {k: v.clone() for k, v in example_batch.items() if torch.is_tensor(v)}
        self.extra = {k: v for k, v in example_batch.items() if not torch.is_tensor(v)}
        self.captures = 0
        self._capture()

    def _weights_key(self):
        # the packed / Winograd-transformed weights are built by the warm-up runs and the graph
        # holds pointers to them: a parameter that changed afterwards (optimizer step,
        # load_state_dict) makes the capture stale
        # (buffers too: the frozen BatchNorm folds are cached per state of the running statistics and
        #  are no longer re-derived inside the captured forward)
        return tuple((p._version, p.data_ptr()) for p in self.model.parameters()) + \
            tuple((b._version, b.data_ptr()) for b in self.model.buffers())

    def _capture(self):
        model = self.model
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):  # packs weights, sets kernel attributes, warms the pool
                model({**self.static_in, **self.extra}, do_postprocessing=self.do_postprocessing)
        torch.cuda.current_stream().wait_stream(side)
        self.graph, kept = _new_graph()
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
            self.static_out = model({**self.static_in, **self.extra},
                                    do_postprocessing=self.do_postprocessing)
        self.graph_info = _repair_and_instantiate(self.graph, kept)
        self._key = self._weights_key()
        self.captures += 1

    def __call__(self, batch):
        if self._weights_key() != self._key:
            self._capture()               # weights changed since the capture: re-pack, re-capture
        for k, v in self.static_in.items():
            v.copy_(batch[k], non_blocking=True)
        self.graph.replay()
        return self.static_out

    def outputs(self):
        return _flatten(self.static_out)


class _TrainStateSnapshot:
    """everything a training step changes besides gradients: parameters + momentum (the flat
    buffers of FusedSGD), BatchNorm running statistics and step counters, Dropout2d step,
    first-step flag"""

    def __init__(self, model, optimizer, bns):
        self.model, self.opt, self.bns = model, optimizer, bns
        self.params = [p.clone() for p in optimizer.flat_params]
        self.momentum = [m.clone() for m in optimizer.flat_momentum]
        self.buffers = [(b, b.clone()) for b in model.buffers()]
        self.pending = [m._emsa_pending for m in bns]
        self.first = optimizer._first
        self.dropout_step = model.dropout_step

    @torch.no_grad()
    def restore(self):
        for dst, src in zip(self.opt.flat_params, self.params):
            dst.copy_(src)
        for dst, src in zip(self.opt.flat_momentum, self.momentum):
            dst.copy_(src)
        for (_, ps, _) in self.opt.buckets.buckets:     # raw copies into the flat storage
            torch.autograd.graph.increment_version(ps)
        for b, src in self.buffers:
            b.copy_(src)
        for m, n in zip(self.bns, self.pending):
            m._emsa_pending = n
        self.opt._first = self.first
        self.opt._upload_hyper()
        self.model.dropout_step = self.dropout_step
        self.model._sync_dropout_state()


class GraphedTrainStep:
    """One whole training step -- forward, backward, fused SGD update -- captured in a hipGraph.

    The eager step issues ~2,500 launches (kernels, zero-fills, gradient bookkeeping) and needs
    35-40 ms of host time however small the batch is; at fp32 / bs=32 the GPU hides that, with
    16-bit storage (58 ms of GPU work) or smaller batches it does not.  The replay needs one launch.

    What makes the step replayable:
      * static shapes: `example_batch` fixes them; inputs are copied into static tensors;
      * Dropout2d masks: {seed, step} live in device memory (`EMSANet.use_device_dropout_state`),
        the mask kernels form the step's seed on the device and a captured one-thread kernel bumps
        the step counter -- every replay draws the masks the eager step would have drawn;
      * learning rate / momentum: read by the update kernel from device memory
        (`FusedSGD.use_device_hyperparameters`), `set_schedule()` between replays is honoured;
      * gradients are (re)written in place: bucket views for the convolution / BatchNorm parameters,
        graph-pool tensors for the rest; BatchNorm running statistics are updated by the captured
        kernels, their step counters by `replay()` on the host.
    `loss_fn(outputs) -> scalar` (or fixed cotangents, `cotangents=[...]` per flattened output).
    Single process: the gradient all-reduce of the data-parallel path is not captured (RCCL
    collectives issued from autograd hooks inside a capture are not supported here)."""

    def __init__(self, model, example_batch, buckets, optimizer, loss_fn=None, cotangents=None,
                 warmup=3, keep_warmup_updates=False):
        """The constructor has to RUN `warmup` real steps on `example_batch` before it can record one
        (they pack the weights, set kernel attributes and size the graph's memory pool).  By default
        their effects are taken back: parameters, momentum buffers, BatchNorm running statistics and
        step counters, the Dropout2d step and the optimizer's first-step flag are snapshotted before
        and restored after the warm-up, so that building the graph does not train the model on one
        batch behind the user's back.  `keep_warmup_updates=True` keeps them (the
        warm-up then counts as `warmup` ordinary training steps at the optimizer's current lr)."""
        if not model.training:
            raise ValueError("GraphedTrainStep captures the train-mode step")
        if buckets.active:
            raise NotImplementedError("graph capture of the multi-rank step (collectives in hooks); "
                                      "see SegmentedGraphedTrainStep")
        if (loss_fn is None) == (cotangents is None):
            raise ValueError("give either loss_fn or cotangents")
        self.model, self.buckets, self.opt = model, buckets, optimizer
        self.loss_fn, self.cots = loss_fn, cotangents
        self.static_in = {k: v.clone() for k, v in example_batch.items() if torch.is_tensor(v)}
        self.extra = {k: v for k, v in example_batch.items() if not torch.is_tensor(v)}
        model.use_device_dropout_state(True)
        optimizer.use_device_hyperparameters(True)
        self._bns = [m for m in model.modules()
                     if hasattr(m, '_emsa_pending') and m.training]
        snap = None if keep_warmup_updates else _TrainStateSnapshot(model, optimizer, self._bns)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):     # packs weights, sets kernel attributes, warms the pool
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if snap is not None:
            snap.restore()
            torch.cuda.synchronize()
        pend = {id(m): m._emsa_pending for m in self._bns}
        self.graph, kept = _new_graph()
        with torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
            self.static_loss, self.static_out = self._step()
        # memset nodes (a user loss written with torch reductions brings them in) -> kernel nodes
        self.graph_info = _repair_and_instantiate(self.graph, kept)
        # the capture RECORDED a step, it did not run one: take its host-side effects back (what
        # ONE step adds to the host-side BatchNorm step counters is replayed by `replay()`)
        self._bn_inc = [(m, m._emsa_pending - pend[id(m)]) for m in self._bns]
        for m, inc in self._bn_inc:
            m._emsa_pending -= inc
        model.dropout_step -= 1
        model._seed_dev_host = (model.dropout_seed & 0xFFFFFFFF, model.dropout_step & 0xFFFFFFFF)
        if snap is not None:
            optimizer._first = snap.first          # (step() inside a capture leaves it alone)
            optimizer._upload_hyper()
        self.replays = 0

    def _step(self):
        self.buckets.reset()
        out = self.model({**self.static_in, **self.extra})
        flat = _flatten(out)
        if self.loss_fn is not None:
            loss = self.loss_fn(out)
            loss.backward()
        else:
            loss = None
            torch.autograd.backward(flat, self.cots)
        self.buckets.finish()
        self.opt.step()
        return loss, out

    def replay(self, batch=None):
        """one training step on `batch` (same shapes as the example; None: the static inputs)"""
        if batch is not None:
            for k, v in self.static_in.items():
                v.copy_(batch[k], non_blocking=True)
        self.graph.replay()
        self.replays += 1
        m = self.model
        m.dropout_step += 1                      # (the device counter was bumped by the graph)
        m._seed_dev_host = (m.dropout_seed & 0xFFFFFFFF, m.dropout_step & 0xFFFFFFFF)
        for bn, inc in self._bn_inc:
            bn._emsa_pending += inc
            if inc:
                mark_bn_stats_written(bn)   # (the graph rewrote the running statistics)
        self.opt.after_replay()     # version counters of the parameters, first-step flag
        return self.static_loss, self.static_out


def segment_parameter_groups(model, cut_stages=(2, 1), decoder_cut=False):
    """Parameters of `model` grouped by the backward SEGMENT that produces their gradients, in
    backward order: [context module + decoders], [encoder stages behind the last cut], ...,
    [encoder stages up to the first cut].  Pass as `GradientBuckets(params, groups=...,
    manual=True)`: no bucket then straddles two segments (= two captured graphs).
    decoder_cut (nn.CutPlan): the first group is two -- [heads + decoder modules 1.. of the dense
    decoders], [their first modules + first side heads, context module, the other decoders]."""
    from .decoder import DecoderBody
    from .nn import CutPlan
    plan = CutPlan(cut_stages, decoder_cut)
    enc = model.encoder
    dec_all = list(model.context_module.parameters()) + list(model.decoders.parameters())
    if decoder_cut:
        bodies = [m for m in model.decoders.modules() if isinstance(m, DecoderBody)
                  and len(m.decoder_modules) > 1]
        dense = {id(p) for b_ in bodies for p in b_.parameters()}
        # (a PanopticHelper wraps two bodies and has no parameters of its own; what is not inside a
        #  dense body -- the scene head -- hangs on the context module: second segment)
        first = {id(p) for b_ in bodies
                 for p in list(b_.decoder_modules[0].parameters()) + list(b_.side_output_heads[0].parameters())}
        seg_a = [p for p in dec_all if id(p) in dense and id(p) not in first]
        seg_b = [p for p in dec_all if id(p) not in dense or id(p) in first]
        groups = [list(reversed(seg_a)), list(reversed(seg_b))]
    else:
        groups = [list(reversed(dec_all))]
    hi = 4
    for c in plan.stages:                         # descending
        ps = []
        for i in range(hi, c, -1):
            ps += list(reversed(enc.stage_parameters(i)))
        groups.append(ps)
        hi = c
    ps = []
    for i in range(hi, -1, -1):
        ps += list(reversed(enc.stage_parameters(i)))
    groups.append(ps)
    groups = [[p for p in g if p.requires_grad] for g in groups]
    listed = {id(p) for g in groups for p in g}
    rest = [p for p in model.parameters() if p.requires_grad and id(p) not in listed]
    if rest:
        raise RuntimeError(f"segment_parameter_groups: {len(rest)} parameters outside encoder / "
                           "context module / decoders")
    return groups


class SegmentedGraphedTrainStep:
    """The MULTI-RANK training step under hipGraphs: RCCL collectives cannot
    be captured from autograd hooks, so the backward pass is cut into segments (nn.CutPlan:
    decoders | encoder stages behind / in front of each cut), each captured as its own graph
    (segment 0 also holds the forward pass and the loss); between two replays the buckets of the
    finished segment are all-reduced EAGERLY on RCCL's stream while the next segment's graph
    already runs.  The optimizer update is the last graph.  Host work per step: one replay per
    segment + one `all_reduce` call per bucket instead of ~4,900 kernel launches (35 ms).

        groups  = segment_parameter_groups(model, cut_stages)
        buckets = GradientBuckets(params, groups=groups, manual=True, average=False,
                                  tail_bytes=4 << 20)
        opt     = FusedSGD(buckets, ...)
        step    = SegmentedGraphedTrainStep(model, batch, buckets, opt, cotangents=cots)
        step.replay(next_batch)

    Works with inactive buckets too (single process: the same graphs, no collectives).  The
    eager twin of one step (`eager_step()`, the same segmented backward without capture) is what
    the tests compare against and what the warm-up runs."""

    def __init__(self, model, example_batch, buckets, optimizer, loss_fn=None, cotangents=None,
                 cut_stages=(2, 1), warmup=2, keep_warmup_updates=False, eager_fallback=False,
                 decoder_cut=False):
        """eager_fallback: if the capture raises (a runtime / RCCL that refuses it), keep the object
        and let replay() run the eager twin -- the same step, same collectives -- instead of
        failing; `capture_error` then holds the reason (bench.py reports it).  Default: raise."""
        from .nn import CutPlan
        if not model.training:
            raise ValueError("SegmentedGraphedTrainStep captures the train-mode step")
        if (loss_fn is None) == (cotangents is None):
            raise ValueError("give either loss_fn or cotangents")
        if buckets.active and not buckets.manual:
            raise ValueError("GradientBuckets(manual=True, groups=segment_parameter_groups(...))")
        self.model, self.buckets, self.opt = model, buckets, optimizer
        self.loss_fn, self.cots = loss_fn, cotangents
        self.plan = CutPlan(cut_stages, decoder_cut)
        self.n_seg = len(self.plan.stages) + 2 + (1 if decoder_cut else 0)
        if len(buckets.group_buckets) != self.n_seg:
            raise ValueError(f"buckets have {len(buckets.group_buckets)} parameter groups, the cut "
                             f"plan {self.n_seg} backward segments")
        self.seg_params = [[p for bi in g for p in buckets.buckets[bi][1]]
                           for g in buckets.group_buckets]
        self.static_in = {k: v.clone() for k, v in example_batch.items() if torch.is_tensor(v)}
        self.extra = {k: v for k, v in example_batch.items() if not torch.is_tensor(v)}
        model.use_device_dropout_state(True)
        optimizer.use_device_hyperparameters(True)
        self._bns = [m for m in model.modules() if hasattr(m, '_emsa_pending') and m.training]
        self.graphs = None
        snap = None if keep_warmup_updates else _TrainStateSnapshot(model, optimizer, self._bns)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if snap is not None:
            snap.restore()
            torch.cuda.synchronize()
        pend = {id(m): m._emsa_pending for m in self._bns}
        self.graphs, self.graph_info = [], []
        keeps = []
        for _ in range(self.n_seg + 1):
            g, kept = _new_graph()
            self.graphs.append(g)
            keeps.append(kept)
        self._capturing = True
        self.capture_error = None
        st0 = dict(buckets.stats)
        ds0, first0 = model.dropout_step, optimizer._first
        try:
            self.static_loss, self.static_out = self._run()
        except Exception as e:                    # noqa: BLE001
            if not eager_fallback:
                raise
            import warnings
            self.capture_error = f'{type(e).__name__}: {e}'[:300]
            warnings.warn(f"SegmentedGraphedTrainStep: capture failed ({self.capture_error}); "
                          "replay() runs the eager segmented step", RuntimeWarning, stacklevel=2)
            self._capturing = False
            torch.cuda.synchronize()
            for k, v in st0.items():
                buckets.stats[k] = v
            # host-side state the aborted capture pass advanced (nothing of it ran on the device):
            # the fallback must start where a successful capture would have
            for m in self._bns:
                m._emsa_pending = pend[id(m)]
            model.dropout_step = ds0
            model._seed_dev_host = None
            model._sync_dropout_state()
            optimizer._first = first0
            optimizer._upload_hyper()
            self._bn_inc = []
            self.graphs, self.graph_info = None, []
            self.replays = 0
            self._ev = None
            return
        finally:
            self._capturing = False
        # what one step gathers (host-side counters of the captured `gather` calls)
        self._per_step = {k: buckets.stats[k] - st0[k] for k in ('gathered_tensors', 'direct_tensors')}
        for k, v in st0.items():
            buckets.stats[k] = v
        for g, kept in zip(self.graphs, keeps):
            self.graph_info.append(_repair_and_instantiate(g, kept))
        self._bn_inc = [(m, m._emsa_pending - pend[id(m)]) for m in self._bns]
        for m, inc in self._bn_inc:
            m._emsa_pending -= inc
        model.dropout_step -= 1
        model._seed_dev_host = (model.dropout_seed & 0xFFFFFFFF, model.dropout_step & 0xFFFFFFFF)
        if snap is not None:
            optimizer._first = snap.first
            optimizer._upload_hyper()
        self.replays = 0
        self._ev = None

    # -- one step, eager or under capture -------------------------------------------------------
    def _segment(self, k):
        """context of backward segment k (k = n_seg: the optimizer update)"""
        import contextlib
        if not getattr(self, '_capturing', False):
            return contextlib.nullcontext()
        pool = self.graphs[0].pool() if k > 0 else None
        return torch.cuda.graph(self.graphs[k], pool=pool, capture_error_mode=_CAPTURE_MODE)

    def _reduce(self, k):
        if self.buckets.active and not getattr(self, '_capturing', False):
            for bi in self.buckets.group_buckets[k]:
                self.buckets.reduce(bi)

    def _run(self):
        model, plan, b = self.model, self.plan, self.buckets
        model._cut_plan = plan
        try:
            D, DM = plan.DECODERS, plan.DECODER_MID
            with self._segment(0):
                b.reset()
                out = model({**self.static_in, **self.extra})
                flat = _flatten(out)
                recs = plan.records
                if plan.decoder_cut:
                    # first decoder segment: heads + later modules; its leaves are the cuts behind
                    # the first modules and the skips the LATER modules add
                    leaves = [c for _, c, st, g in recs
                              if g == DM or (g == D and st not in plan.late_stages)]
                else:
                    leaves = [c for _, c, _, g in recs if g == D]
                if self.loss_fn is not None:
                    loss = self.loss_fn(out)
                    roots, grads = [loss], [None]
                else:
                    loss = None
                    roots, grads = list(flat), list(self.cots)
                # (decoder_cut: the second segment starts from the same roots again -- the first side
                #  outputs and the scene logits reach the first modules / the context module without
                #  passing a cut -- so this pass keeps the graph)
                torch.autograd.backward(roots, grads, inputs=leaves + self.seg_params[0],
                                        retain_graph=plan.decoder_cut)
                for bi in b.group_buckets[0]:
                    b.gather(bi)
            self._reduce(0)
            k0 = 1
            if plan.decoder_cut:
                with self._segment(1):
                    mid = [(o, c.grad) for o, c, _, g in recs if g == DM and c.grad is not None]
                    leaves = [c for _, c, st, g in recs if g == D and st in plan.late_stages]
                    torch.autograd.backward(roots + [o for o, _ in mid], grads + [g for _, g in mid],
                                            inputs=leaves + self.seg_params[1])
                    for bi in b.group_buckets[1]:
                        b.gather(bi)
                self._reduce(1)
                k0 = 2
            pending = [(o, c.grad, st) for o, c, st, g in recs
                       if g == D and c.grad is not None]
            bounds = list(plan.stages) + [-1]
            for k, cstage in enumerate(bounds, start=k0):
                with self._segment(k):
                    roots_k = [(o, g) for o, g, st in pending if st > cstage]
                    pending = [(o, g, st) for o, g, st in pending if st <= cstage]
                    merged = {}
                    for o, g in roots_k:      # one tensor may have been cut twice (skip + encoder cut)
                        merged[id(o)] = (o, g if id(o) not in merged else merged[id(o)][1] + g)
                    leaves = [c for _, c, _, g in plan.records if g == cstage]
                    torch.autograd.backward([o for o, _ in merged.values()],
                                            [g for _, g in merged.values()],
                                            inputs=leaves + self.seg_params[k])
                    for bi in b.group_buckets[k]:
                        b.gather(bi)
                self._reduce(k)
                pending += [(o, c.grad, st) for o, c, st, g in plan.records
                            if g == cstage and c.grad is not None]
            if not getattr(self, '_capturing', False):
                b.finish()
            with self._segment(self.n_seg):
                self.opt.step()
        finally:
            model._cut_plan = None
            plan.records = []
        return loss, out

    def eager_step(self, batch=None):
        """the same segmented step without graphs (warm-up; the tests' eager twin)"""
        if batch is not None:
            for k, v in self.static_in.items():
                v.copy_(batch[k], non_blocking=True)
        return self._run()

    def replay(self, batch=None):
        """one training step on `batch` (same shapes as the example; None: the static inputs)"""
        if self.graphs is None:                   # eager_fallback after a failed capture
            self.replays += 1
            return self.eager_step(batch)
        if batch is not None:
            for k, v in self.static_in.items():
                v.copy_(batch[k], non_blocking=True)
        b = self.buckets
        b.begin_step_host()
        for k, v in self._per_step.items():
            b.stats[k] += v
        timed = b.active
        ev = []
        for k in range(self.n_seg):
            self.graphs[k].replay()
            if timed:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((e, list(b.group_buckets[k])))
                for bi in b.group_buckets[k]:
                    b.reduce(bi)
        if timed:
            e_end = torch.cuda.Event(enable_timing=True)
            e_end.record()
            self._ev = (ev, e_end)
            b.finish()
        self.graphs[self.n_seg].replay()
        self.replays += 1
        m = self.model
        m.dropout_step += 1
        m._seed_dev_host = (m.dropout_seed & 0xFFFFFFFF, m.dropout_step & 0xFFFFFFFF)
        for bn, inc in self._bn_inc:
            bn._emsa_pending += inc
            if inc:
                mark_bn_stats_written(bn)   # (the graph rewrote the running statistics)
        self.opt.after_replay()
        return self.static_loss, self.static_out

    def bucket_launch_ms_before_backward_end(self):
        """per bucket: how long before the END of the backward pass (device time line) its
        all-reduce was issued, for the last replay; call after a device synchronisation.  0 for the
        buckets of the last segment -- by construction the only exposed ones."""
        if self._ev is None:
            return None
        ev, e_end = self._ev
        lead = {}
        for e, bis in ev:
            ms = e.elapsed_time(e_end)
            for bi in bis:
                lead[bi] = round(ms, 3)
        return [lead[bi] for bi in sorted(lead)]
