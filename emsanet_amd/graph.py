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


def _flatten(out):
    if torch.is_tensor(out):
        return [out]
    if isinstance(out, dict):
        return [t for v in out.values() for t in _flatten(v)]
    if isinstance(out, (list, tuple)):
        return [t for v in out for t in _flatten(v)]
    return []


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
        return tuple((p._version, p.data_ptr()) for p in self.model.parameters())

    def _capture(self):
        model = self.model
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):  # packs weights, sets kernel attributes, warms the pool
                model({**self.static_in, **self.extra}, do_postprocessing=self.do_postprocessing)
        torch.cuda.current_stream().wait_stream(side)
        self.graph, kept = _new_graph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
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
        with torch.cuda.graph(self.graph):
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
        self.opt.after_replay()     # version counters of the parameters, first-step flag
        return self.static_loss, self.static_out
