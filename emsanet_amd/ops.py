# -*- coding: utf-8 -*-
"""
Fused forward/backward operators of the EMSANet engine as torch.autograd.Functions.

Granularity follows the fusion plan of DESIGN.md, not the reference's module list: one Function
per NonBottleneck1D block (4 MFMA convs + 2 BatchNorms + Dropout2d + residual, hand-written
backward with the ReLU masks and the residual add fused into the data-gradient epilogues), one
per conv+BN+act, per SE fusion, per learned upsampling ...  Every Function only sequences calls
into libemsanet_hip.so (emsanet_amd/functional.py); autograd is used for graph bookkeeping only.
"""
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from . import functional as Fn
from ._lib import check
from .functional import ACT_NONE, ACT_RELU
from .parallel import grad_target

# debug: set to a list to record (tag, tensor clone) for every backward's inputs and outputs
TRACE = None
# tests: set to a list to record every sign decision of the forward pass (ReLU outputs > 0) in
# execution order -- tests/test_model_gpu.py replays them inside the fp64 oracle so that the
# engine's gradients can be compared at a TIGHT tolerance (same piecewise-linear branch on both
# sides; without it a single ReLU flipped by fp32 roundoff moves every upstream gradient by ~1e-2)
MASK_TRACE = None


# tools/actgrad_compare.py: set to a dict to record, per convolution module (key: id(module)), the
# gradient w.r.t. that convolution's OUTPUT as its backward kernels receive it -- the quantity a
# tensor hook on the conv output gives in plain PyTorch -- to localise where activation gradients
# of engine and oracle part
GRAD_TRACE = None


def _trace_grad(module, dy):
    if GRAD_TRACE is not None:
        GRAD_TRACE[id(module)] = dy.detach().float().cpu()


def _trace_mask(tag, t):
    if MASK_TRACE is not None:
        MASK_TRACE.append((tag, t > 0))


def _traced(fn):
    """wrap a Function.backward: with ops.TRACE = [] every incoming / outgoing gradient is cloned
    into the list (tools/determinism.py uses it to find the first diverging tensor)."""
    def wrapper(ctx, *grads):
        if TRACE is None:
            return fn(ctx, *grads)
        tag = fn.__qualname__.split('.')[0]
        for i, g in enumerate(grads):
            if torch.is_tensor(g):
                TRACE.append((f'{tag}.in{i}', g.detach().clone()))
        out = fn(ctx, *grads)
        for i, g in enumerate(out if isinstance(out, tuple) else (out,)):
            if torch.is_tensor(g):
                TRACE.append((f'{tag}.out{i}', g.detach().clone()))
        return out
    wrapper.__qualname__ = fn.__qualname__
    return wrapper


# ---------------------------------------------------------------------------------------------
# runtime views of parameter containers
# ---------------------------------------------------------------------------------------------
class ConvRT:
    """nn.Conv2d used as a parameter container + cached packed weights.  fp32 activations run on
    the fp32 kernels (Winograd where eligible); 16-bit activations on conv_h.hip with the weights
    packed in the activations' dtype (the fp32 parameter stays the master copy)."""

    def __init__(self, conv):
        self.conv = conv
        self.spec = Fn.ConvSpec(conv.in_channels, conv.out_channels, tuple(conv.kernel_size),
                                tuple(conv.stride), tuple(conv.padding))
        self._key = None
        self._wp = None
        self._keyd = None
        self._wpd = None
        # stride-1 3x1 / 1x3 convs run on the Winograd F(2,3) kernel (EMSA_WINO=0 disables it)
        self.wino = Fn.wino_eligible(self.spec) and os.environ.get('EMSA_WINO', '1') != '0'
        self._keyu = None
        self._u = None
        self._ud = None
        self._h = {}              # 16-bit packs: dtype -> [key, fwd, key_d, dgrad]
        # stride-1 3x1 / 1x3 convs with 64..512 channels in 16-bit storage: the register-stationary
        # streaming kernel (conv_rs.hip) with fragment-ordered weights, where it takes the geometry
        self.rs = Fn.rs_eligible(self.spec)
        self._hf = {}             # fragment-ordered 16-bit packs: dtype -> [key, fwd, key_d, dgrad]
        self.plain_asked = 0      # times the [tap][n][k] 16-bit operand of an rs conv was read (PackPlan)

    def spec_key(self):
        s = self.spec
        return (s.cin, s.cout, s.kh, s.kw, s.sh, s.sw, s.ph, s.pw)

    def _wino_weights(self):
        w = self.conv.weight
        key = (w._version, w.data_ptr())
        if key != self._keyu:
            self._u, self._ud = Fn.pack_wino(w.detach(), fwd=True, dgrad=w.requires_grad)
            self._keyu = key
        return self._u, self._ud

    def forward(self, x, **kw):
        """conv forward through the best kernel for this layer (fused-epilogue kwargs of conv_fwd)"""
        if x.dtype != torch.float32:
            wf = self.frag(x.dtype)
            return Fn.conv_fwd(x, self._plain(x.dtype, wf, self.packed), self.spec, wfrag=wf, **kw)
        if self.wino:
            return Fn.conv_fwd(x, None, self.spec, wino_u=self._wino_weights()[0], **kw)
        return Fn.conv_fwd(x, self.packed(), self.spec, **kw)

    def folds_input_bn(self, t):
        """this conv can take the BatchNorm + ReLU in front of it (input tensor `t`) into its
        loader: fp32 1-D Winograd form, forward and weight gradient; Fn.bn1_fold decides whether
        it pays at this size"""
        if t.dtype != torch.float32:
            return self.rs and Fn.bn1_fold16(t, self.spec)
        return self.wino and Fn.wino_rows(self.spec) == 1 and Fn.bn1_fold(t)

    def dgrad(self, dy, in_hw, mask_bits=None, **kw):
        """mask_bits: the ReLU mask of the producing layer as bits (conv_fwd(want_relu_bits));
        used by the Winograd kernel, otherwise the float `mask_src` applies"""
        if dy.dtype != torch.float32:
            wf = self.frag_dgrad(dy.dtype)
            return Fn.conv_dgrad(dy, self._plain(dy.dtype, wf, self.packed_dgrad), self.spec, in_hw,
                                 wfrag=wf, **kw)
        if self.wino:
            u, ud = self._wino_weights()
            if ud is None:
                ud = Fn.pack_wino(self.conv.weight.detach(), fwd=False, dgrad=True)[1]
            return Fn.conv_dgrad(dy, None, self.spec, in_hw, wino_u=ud, mask_bits=mask_bits, **kw)
        return Fn.conv_dgrad(dy, self.packed_dgrad(), self.spec, in_hw, **kw)

    def dgrad_bnb(self, dy, in_hw, t, affine, mean, invstd, residual=None, force=False):
        """data gradient + the backward reduction of the BatchNorm in front of this conv
        (Fn.conv_dgrad_bnb), or None when this conv's kernel has no such epilogue (fp32 without
        the Winograd form).  force: the caller has no other way (the forward folded the BatchNorm
        into this conv's loader: no ReLU bit mask exists)"""
        if affine is None or not (force or Fn.bn_fused_reduce(dy.dtype, self.spec.cin)):
            return None
        scale, shift = affine
        if dy.dtype != torch.float32:
            wf = self.frag_dgrad(dy.dtype)
            return Fn.conv_dgrad_bnb(dy, self._plain(dy.dtype, wf, self.packed_dgrad), self.spec, in_hw,
                                     t, scale, shift, mean, invstd, residual=residual, wfrag=wf)
        if not self.wino:
            return None
        u, ud = self._wino_weights()
        if ud is None:
            ud = Fn.pack_wino(self.conv.weight.detach(), fwd=False, dgrad=True)[1]
        return Fn.conv_dgrad_bnb(dy, None, self.spec, in_hw, t, scale, shift, mean, invstd,
                                 residual=residual, wino_u=ud)

    def _plain(self, dtype, wfrag, getter):
        """the [tap][n][k] operand of the implicit GEMM: at once for a conv without fragment-ordered
        weights, ON DEMAND for a conv_rs-capable one -- where conv_rs takes the geometry (every NBt1D
        conv of a training step) nobody reads it, and PackPlan stops packing it (`plain_asked`
        tells it when somebody did)"""
        if wfrag is None:
            return getter(dtype)

        return lambda: getter(dtype)

    def frag(self, dtype):
        """fragment-ordered forward weights for conv_rs.hip (None: not that kind of conv)"""
        if not (self.rs and Fn.CONV_RS):
            return None
        w = self.conv.weight
        key = (w._version, w.data_ptr())
        ent = self._hf.setdefault(dtype, [None, None, None, None])
        if ent[0] != key:
            ent[1], d = Fn.pack_weight_frag_t(w.detach(), dtype, fwd=True, dgrad=w.requires_grad)
            ent[0] = key
            if d is not None:
                ent[2], ent[3] = key, d
        return ent[1]

    def frag_dgrad(self, dtype):
        if not (self.rs and Fn.CONV_RS):
            return None
        w = self.conv.weight
        key = (w._version, w.data_ptr())
        ent = self._hf.setdefault(dtype, [None, None, None, None])
        if ent[2] != key:
            ent[3] = Fn.pack_weight_frag_t(w.detach(), dtype, fwd=False, dgrad=True)[1]
            ent[2] = key
        return ent[3]

    def packed(self, dtype=torch.float32):
        w = self.conv.weight
        key = (w._version, w.data_ptr())
        if dtype != torch.float32:
            self.plain_asked += 1
            ent = self._h.setdefault(dtype, [None, None, None, None])
            if ent[0] != key:
                ent[1], d = Fn.pack_weight_t(w.detach(), dtype, fwd=True, dgrad=w.requires_grad)
                ent[0] = key
                if d is not None:
                    ent[2], ent[3] = key, d
            return ent[1]
        if key != self._key:
            if w.requires_grad:
                # trainable: the data-gradient layout will be needed too -> one launch for both
                self._wp, self._wpd = Fn.pack_weight_pair(w.detach())
                self._keyd = key
            else:
                self._wp = Fn.pack_weight(w.detach(), 'fwd')
            self._key = key
        return self._wp

    def packed_dgrad(self, dtype=torch.float32):
        w = self.conv.weight
        key = (w._version, w.data_ptr())
        if dtype != torch.float32:
            self.plain_asked += 1
            ent = self._h.setdefault(dtype, [None, None, None, None])
            if ent[2] != key:
                ent[3] = Fn.pack_weight_t(w.detach(), dtype, fwd=False, dgrad=True)[1]
                ent[2] = key
            return ent[3]
        if key != self._keyd:
            self._wpd = Fn.pack_weight(w.detach(), 'dgrad')
            self._keyd = key
        return self._wpd


# EMSA_PACK_LEAN=0: always pack both 16-bit operand forms of the conv_rs-capable convs (A/B runs)
PACK_LEAN = os.environ.get('EMSA_PACK_LEAN', '1') != '0'


class PackPlan:
    """All weight transforms of a model (packed implicit-GEMM layouts, Winograd U, or the 16-bit
    operands of the mixed-precision path) in ONE launch per optimizer step instead of one ~5 us
    kernel per layer (380 launches = 2.3 ms per step).  `refresh(dtype)` (called at the start of
    the model's forward) re-packs every registered ConvRT when any weight changed and installs the
    results as the ConvRTs' own caches, so their per-layer fallbacks (`packed()`,
    `_wino_weights()`) find fresh entries and launch nothing."""

    def __init__(self, rts):
        self.rts = [rt for rt in rts if isinstance(rt, ConvRT)]
        # merged / channel-padded head convs (MultiConvRT: task heads, side heads, scene head): one
        # job per placed parameter (+ one per bias) into persistent zero-initialised operands --
        # they used to re-pack themselves every step with a zero-fill and 2-4 kernels each
        self.multi = [rt for rt in rts if isinstance(rt, MultiConvRT)]
        self._mviews = None
        self._key = None
        self._ptrs = None
        self._jobs = None
        self._n_blocks = 0
        self._arena = None
        self._views = None
        self._dtype = None
        # lean: the conv_rs-capable convs get their fragment-ordered 16-bit operands only -- their
        # [tap][n][k] form (half of the 16-bit pack launch: 188 convs of the full model) is read only
        # where conv_rs refuses a geometry.  Decided from what the previous step asked for
        # (ConvRT.plain_asked): full table first, lean once a whole step went by without a reader,
        # full again as soon as one turns up (who meanwhile packs per layer, ConvRT.packed)
        self._lean = False
        self._asked_seen = None
        self._refreshes = 0
        # a hipGraph that captured `emsa_pack_batch` holds RAW pointers to the job table, the arena
        # and the merged-head operands.  Once a refresh ran under capture the lean / full state is
        # frozen (the table a replay reads must stay the one it recorded) and whatever a later
        # rebuild supersedes (new dtype, moved parameters) is kept alive in `_retired` instead of
        # going back to the allocator, where a replay would scribble over someone else's memory
        # (ADVICE r5).  `thaw()` drops both when the graphs are gone.
        self._captured = False
        self._retired = []

    def thaw(self):
        """no captured graph replays this plan's pack launch any more: allow lean / full switches
        again and release the superseded tables"""
        self._captured = False
        self._retired.clear()

    def _asked(self):
        return sum(rt.plain_asked for rt in self.rts if rt.rs)

    def _build(self, dev, dtype, lean=False):
        if self._captured and self._jobs is not None:
            self._retired.append((self._jobs, self._arena, self._views, self._frag, self._mviews))
        self._lean = lean
        n = len(self.rts)
        n_multi = sum(len(m.placements) + sum(1 for q, _, _ in m.placements if q.bias is not None)
                      for m in self.multi)
        half = dtype != torch.float32
        # the stride-1 3-tap 1-D convs additionally get the fragment-ordered operands of conv_rs.hip
        # (pack kinds 5 / 6) -- the [tap][n][k] form stays for geometries that kernel refuses
        frag_rts = [rt for rt in self.rts if half and rt.rs and Fn.CONV_RS]
        lean = self._lean = bool(lean and frag_rts)
        skip = set(id(rt) for rt in frag_rts) if lean else set()
        jobs = (_lib.EmsaPackJob * (n + n_multi + len(frag_rts)))()
        esz = 2 if half else 4
        sizes = []
        for rt in self.rts:
            w = rt.conv.weight
            numel = w.numel()
            per = numel if half else (numel * 4 // 3 if rt.wino else numel)   # U: 4 comps per 3 taps
            per_b = (per * esz + 15) // 16 * 16            # every pack starts 16-byte aligned
            if id(rt) in skip:
                sizes.append((0, 0, 0))
            else:
                sizes.append((per, per_b, per_b if w.requires_grad else 0))
        fsizes = [((rt.conv.weight.numel() * esz + 15) // 16 * 16) * (2 if rt.conv.weight.requires_grad else 1)
                  for rt in frag_rts]
        arena = torch.empty(sum(a + b for _, a, b in sizes) + sum(fsizes), device=dev,
                            dtype=torch.uint8)
        views, off, blk = [], 0, 0
        for j, (rt, (per, a, b)) in enumerate(zip(self.rts, sizes)):
            w = rt.conv.weight
            if id(rt) in skip:
                # lean: no [tap][n][k] operand for this conv -- an empty job keeps the table's indexing
                views.append((None, None))
                jobs[j] = _lib.EmsaPackJob(w.data_ptr(), w.data_ptr(), None, 0, 1, 1, 1, 4, blk)
                blk += 1
                continue
            v0 = arena[off:off + per * esz].view(dtype)
            v1 = arena[off + a:off + a + per * esz].view(dtype) if b else None
            off += a + b
            views.append((v0, v1))
            cout, cin, kh, kw = w.shape
            kind = Fn.DT[dtype] + 1 if half else (1 if rt.wino else 0)   # 2 = bf16, 3 = fp16
            jobs[j] = _lib.EmsaPackJob(w.data_ptr(), v0.data_ptr(), v1.data_ptr() if b else None,
                                       cout, cin, kh, kw, kind, blk)
            blk += max(1, min(64, (w.numel() + 2047) // 2048))
        j = n
        fviews = []
        for rt, fs in zip(frag_rts, fsizes):
            w = rt.conv.weight
            nb = w.numel() * esz
            one = fs // 2 if w.requires_grad else fs
            f0 = arena[off:off + nb].view(dtype)
            f1 = arena[off + one:off + one + nb].view(dtype) if w.requires_grad else None
            off += fs
            fviews.append((f0, f1))
            cout, cin, kh, kw = w.shape
            jobs[j] = _lib.EmsaPackJob(w.data_ptr(), f0.data_ptr(),
                                       f1.data_ptr() if f1 is not None else None, cout, cin, kh, kw,
                                       Fn.DT[dtype] + 4, blk)            # 5 = bf16, 6 = fp16
            blk += max(1, min(64, (w.numel() + 2047) // 2048))
            j += 1
        self._frag = (frag_rts, fviews)
        mviews = []
        for m in self.multi:
            sp = m.spec
            taps = sp.kh * sp.kw
            wino = m.wino and not half
            numel = taps * sp.cout * sp.cin
            per = numel * 4 // 3 if wino else numel
            d0 = torch.zeros(per, device=dev, dtype=dtype)          # forward operand / Winograd U
            d1 = torch.zeros(per, device=dev, dtype=dtype)          # data-gradient operand / U
            bias = torch.zeros(sp.cout, device=dev, dtype=torch.float32) if m.has_bias else None
            kind = Fn.DT[dtype] + 1 if half else (1 if wino else 0)
            for q, co, ci in m.placements:
                w4 = m._w4(q)
                cout, cin, kh, kw = w4.shape
                jobs[j] = _lib.EmsaPackJob(q.weight.data_ptr(), d0.data_ptr(), d1.data_ptr(), cout,
                                           cin, kh, kw, kind, blk, sp.cout, co, sp.cin, ci)
                blk += max(1, min(64, (w4.numel() + 2047) // 2048))
                j += 1
                if q.bias is not None:
                    jobs[j] = _lib.EmsaPackJob(q.bias.data_ptr(), bias.data_ptr(), None,
                                               q.bias.shape[0], 1, 1, 1, 4, blk, sp.cout, co, 1, 0)
                    blk += 1
                    j += 1
            mviews.append((d0, d1, bias, wino))
        self._mviews = mviews
        self._n_jobs = j
        raw = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        self._jobs, self._n_blocks, self._arena, self._views = raw, blk, arena, views
        self._mptrs = tuple(q.weight.data_ptr() for m in self.multi for q, _, _ in m.placements)
        self._ptrs = tuple(rt.conv.weight.data_ptr() for rt in self.rts)
        self._dtype = dtype

    def refresh(self, dtype=torch.float32):
        if not self.rts:
            return
        ws = [rt.conv.weight for rt in self.rts]
        mkeys = [m._key_now() for m in self.multi]
        key = (dtype,) + tuple((w._version, w.data_ptr(), w.requires_grad) for w in ws) + \
            tuple(mkeys)
        if key == self._key:
            return
        if not ws[0].is_cuda:
            return                                   # host-side dry runs: per-layer path
        mptrs = tuple(q.weight.data_ptr() for m in self.multi for q, _, _ in m.placements)
        if self._dtype != dtype or self._ptrs != tuple(w.data_ptr() for w in ws) or \
                getattr(self, '_mptrs', None) != mptrs or \
                any(v0 is not None and (v1 is None) == w.requires_grad
                    for (v0, v1), w in zip(self._views or [], ws)):
            self._build(ws[0].device, dtype, self._lean)
        # lean / full job table, from what the step behind us read (never switched inside a capture:
        # _build uploads the table)
        asked = self._asked()
        if torch.cuda.is_current_stream_capturing():
            self._captured = True
        if PACK_LEAN and dtype != torch.float32 and not self._captured:
            if self._lean and asked != self._asked_seen:
                self._build(ws[0].device, dtype, False)
            elif not self._lean and self._asked_seen is not None and asked == self._asked_seen:
                self._build(ws[0].device, dtype, True)
        self._asked_seen = asked
        check(_lib.lib().emsa_pack_batch(self._jobs.data_ptr(), self._n_jobs, self._n_blocks,
                                         Fn._stream()), 'emsa_pack_batch')
        for m, mk, (d0, d1, bias, wino) in zip(self.multi, mkeys, self._mviews):
            if dtype != torch.float32:
                m._h[dtype] = (mk, d0, bias)
                m._hd[dtype] = (mk, d1, None)
            elif wino:
                m._wp, m._bias, m._u, m._key = None, bias, d0, mk
                m._hd[dtype] = (mk, None, d1)
            else:
                m._wp, m._bias, m._u, m._key = d0, bias, None, mk
                m._hd[dtype] = (mk, d1, None)
        for rt, (f0, f1) in zip(*self._frag):
            w = rt.conv.weight
            k = (w._version, w.data_ptr())
            rt._hf[dtype] = [k, f0, k if f1 is not None else None, f1]
        for rt, (v0, v1), w in zip(self.rts, self._views, ws):
            k = (w._version, w.data_ptr())
            if v0 is None:
                continue                       # lean: ConvRT.packed packs this one itself if asked
            if dtype != torch.float32:
                rt._h[dtype] = [k, v0, k if v1 is not None else None, v1]
            elif rt.wino:
                rt._u, rt._ud, rt._keyu = v0, v1, k
            else:
                rt._wp, rt._key = v0, k
                if v1 is not None:
                    rt._wpd, rt._keyd = v1, k
        self._key = key


# BatchNorm step counters (`num_batches_tracked`): nn.BatchNorm2d.forward is never called, so the
# engine counts training steps on the host -- ON THE MODULE (`bn._emsa_pending`), not in a global
# registry: a deep copy of the model (EMA twin) carries its own counts and hooks, and a
# sub-module's `state_dict()` flushes exactly like the whole model's.  The hooks are
# module-level functions (no closure over a runtime object), so `copy.deepcopy` keeps them valid.
def _bn_flush_hook(module, prefix=None, keep_vars=None):
    n = getattr(module, '_emsa_pending', 0)
    if n and module.num_batches_tracked is not None:
        module.num_batches_tracked += n
    module._emsa_pending = 0


def _bn_load_hook(module, state_dict, prefix, *unused):
    # counts gathered before a checkpoint is loaded belong to the state that is being replaced
    module._emsa_pending = 0


def mark_bn_stats_written(bn):
    """the running statistics of `bn` were (or are about to be) rewritten through raw pointers --
    `emsa_bn_finalize`, a replayed hipGraph -- which does not move their tensor version counters.
    Everything that caches a function of them keys on `_version` (BNRT._frozen_fold, the
    inference graph's `_weights_key`): bump it here, or the next eval silently runs on the folded
    statistics of an older state (eval -> train-mode forward under no_grad, frozen
    affine parameters -> eval)."""
    if bn.running_mean is not None:
        torch.autograd.graph.increment_version(bn.running_mean)
        torch.autograd.graph.increment_version(bn.running_var)


def flush_bn_counters(root):
    """add the host-side training-step counts of every BatchNorm below `root` to the
    `num_batches_tracked` buffers.  (`state_dict()` of the model or of any sub-module does this
    through the per-module pre-hook; a direct read of `bn.num_batches_tracked` during training
    needs this call first.)"""
    for m in root.modules():
        if hasattr(m, '_emsa_pending'):
            _bn_flush_hook(m)


class BNRT:
    """nn.BatchNorm2d used as a parameter/buffer container."""

    def __init__(self, bn):
        self.bn = bn
        if not hasattr(bn, '_emsa_pending'):
            bn._emsa_pending = 0
            bn.register_state_dict_pre_hook(_bn_flush_hook)
            bn.register_load_state_dict_pre_hook(_bn_load_hook)

    @property
    def pending_batches(self):
        return self.bn._emsa_pending

    @pending_batches.setter
    def pending_batches(self, value):
        self.bn._emsa_pending = value

    def batch_stats(self):
        # torch semantics: batch statistics in training mode or when no running stats exist
        return self.bn.training or self.bn.running_mean is None

    def running(self):
        bn = self.bn
        if bn.training and bn.track_running_stats and bn.running_mean is not None:
            return bn.running_mean, bn.running_var
        return None, None

    def forward_stats(self, stats, count):
        """-> scale, shift, mean, invstd for this call (updates running stats in train mode)."""
        bn = self.bn
        g, b = bn.weight.detach(), bn.bias.detach()
        if self.batch_stats():
            rm, rv = self.running()
            mom = bn.momentum if bn.momentum is not None else 0.1
            if rm is not None and bn.num_batches_tracked is not None:
                # nn.BatchNorm2d.forward is never called: keep its step counter like torch does
                # (checkpoints carry it); counted on the host, flushed by `flush_bn_counters`
                self.pending_batches += 1
            if rm is not None:
                mark_bn_stats_written(bn)
            return Fn.bn_finalize(stats, count, g, b, bn.eps, mom, rm, rv)
        scale, shift, invstd = self._frozen_fold()
        return scale, shift, bn.running_mean, invstd

    def _frozen_fold(self):
        """scale / shift / invstd of the frozen BatchNorm (running statistics), computed once per
        state of (weight, bias, running_mean, running_var) instead of once per forward pass: the
        bs=1 inference graph spent 127 of its ~440 launches re-deriving these constants (17 % of its
        GPU time, round 4).  Re-derived IN PLACE when a tensor changed, so that a captured graph's
        pointers stay valid."""
        bn = self.bn
        ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = tuple((t._version, t.data_ptr()) for t in ts) + (bn.eps,)
        ent = getattr(self, '_fold_cache', None)
        if ent is None or ent[0] != key or ent[1].device != bn.weight.device:
            buf = ent[1] if ent is not None and ent[1].device == bn.weight.device else None
            s, t, inv = Fn.bn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                   bn.running_var, bn.eps, out=buf)
            self._fold_cache = ent = (key, s._base if s._base is not None else s, s, t, inv)
        return ent[2], ent[3], ent[4]

    def folded(self):
        s, t, _ = self._frozen_fold()
        return s, t


def _conv_bn_forward(x, crt, brt, act, drop=None, residual=None):
    """conv (+bias) -> BN (batch or frozen statistics) -> *drop -> +residual -> act.
    returns out, y_raw, mean, invstd, relu_mask (bit mask of out > 0 for the backward pass, or
    None without ReLU)"""
    bias = crt.conv.bias.detach() if crt.conv.bias is not None else None
    if brt.batch_stats():
        y, stats = crt.forward(x, bias=bias, want_stats=True)
        count = y.shape[0] * y.shape[2] * y.shape[3]
    else:
        y, stats, count = crt.forward(x, bias=bias), None, 0
    scale, shift, mean, invstd = brt.forward_stats(stats, count)
    out, mask = Fn.bn_act(y, scale, shift, drop, residual, act, want_mask=True)
    # (the affine form travels with the statistics: a data gradient with the fused BatchNorm
    #  reduction recomputes the ReLU mask from it)
    if brt.batch_stats():                # (`mean` is this call's own tensor, not the running buffer)
        mean._emsa_affine = (scale, shift)
    return out, y, mean, invstd, mask


def _bn_targets(brt):
    """destinations of dgamma / dbeta inside the flat gradient buckets (or nothing)"""
    tg, tb = grad_target(brt.bn.weight), grad_target(brt.bn.bias)
    return {'dg_out': tg, 'db_out': tb} if tg is not None and tb is not None else {}


# EMSA_WGRAD_STREAM=1 / 2: the (deferred) weight gradients of the NBt1D blocks run on a side stream of
# LOW (1) / default (2) priority that joins the issuing stream when the backward pass ends.  The idea
# (round 5): the backward critical path alternates matrix-bound data gradients with HBM-bound
# BatchNorm passes; weight gradients are matrix-bound and nothing but the optimizer waits for them,
# so at low queue priority they can take the CU slots the HBM-bound passes leave idle instead of
# extending the critical path.  (Rounds 2 / 3 measured a plain second stream at default priority,
# joined per block, SLOWER: the weight gradients then displace their own block's data gradients.)
WGRAD_STREAM = int(os.environ.get('EMSA_WGRAD_STREAM', '0'))
_wgrad_sides = {}          # id of the issuing stream -> side stream
_wgrad_pending = []        # side streams with work of the current backward pass


def _wgrad_side(cur):
    key = (cur.device.index, cur.cuda_stream)
    side = _wgrad_sides.get(key)
    if side is None:
        lo, hi = 0, 0
        try:
            # (torch: smaller number = higher priority; ROCm exposes low = +1 where the runtime has it)
            side = torch.cuda.Stream(device=cur.device, priority=1 if WGRAD_STREAM == 1 else 0)
        except Exception:                       # noqa: BLE001
            side = torch.cuda.Stream(device=cur.device)
        _wgrad_sides[key] = side
        from .parallel import register_stream
        register_stream(side)
    return side


def _join_wgrad_streams():
    cur = torch.cuda.current_stream()
    while _wgrad_pending:
        cur.wait_stream(_wgrad_pending.pop())


class _Deferred:
    """placeholder of a weight / bias gradient whose launch was deferred (`_flush_wgrads`)"""

    def __init__(self, index, which):
        self.index, self.which = index, which


_in_side = False


class _SideGuard:
    def __enter__(self):
        global _in_side
        _in_side = True

    def __exit__(self, *exc):
        global _in_side
        _in_side = False


def _wgrad_now(x, dy, crt, tw, tb, in_affine=None):
    conv = crt.conv
    dw, db, packed = Fn.conv_wgrad(x, dy, crt.spec, conv.bias is not None, like=conv.weight,
                                   dw_out=tw, db_out=tb, in_affine=in_affine)
    if packed:
        dw = Fn.unpack_wgrad(dw, conv.weight, out=tw)
    elif tw is not None and dw.data_ptr() == tw.data_ptr():
        dw = tw
    return dw, db


def _flush_wgrads(defer, grads):
    """launch the deferred weight gradients -- convs of equal channel counts together in one
    multi-job launch (Fn.conv_wgrad_multi; the four convs of an NBt1D block), the rest one by one --
    and put the results in the places their `_Deferred` placeholders hold in `grads`"""
    if not defer:
        return grads
    if WGRAD_STREAM and defer[0][0].is_cuda and not _in_side:
        cur = torch.cuda.current_stream()
        side = _wgrad_side(cur)
        side.wait_stream(cur)
        with torch.cuda.stream(side), _SideGuard():
            out = _flush_wgrads(defer, grads)
        for x, dy, *_ in defer:                 # (the caching allocator must not recycle them early)
            x.record_stream(side)
            dy.record_stream(side)
        # The join may wait for the end of the backward pass only where nobody reads these gradients
        # before: every one of them written into a bucket view of MANUAL buckets (no autograd hooks: the
        # caller reduces after backward).  Otherwise AccumulateGrad (adding into an existing .grad) or a
        # hook-driven bucket all-reduce would consume them while the side stream still writes (ADVICE
        # r5): join right here, and tell the allocator which stream the fresh tensors are used on.
        def _manual(job):
            slot = getattr(job[2].conv.weight, '_emsa_grad_slot', None)
            bias = job[2].conv.bias
            bslot = getattr(bias, '_emsa_grad_slot', None) if bias is not None else slot
            return (job[3] is not None and slot is not None and getattr(slot[1], 'manual', False) and
                    (bias is None or (job[4] is not None and bslot is not None)))
        if all(_manual(job) for job in defer):
            if not _wgrad_pending:
                torch.autograd.Variable._execution_engine.queue_callback(_join_wgrad_streams)
            if side not in _wgrad_pending:
                _wgrad_pending.append(side)
        else:
            cur.wait_stream(side)
            for g in out:
                if torch.is_tensor(g):
                    g.record_stream(cur)
        return out
    res = [None] * len(defer)
    groups = {}
    for i, job in enumerate(defer):
        ok = (job[5] is None or job[0].dtype != torch.float32) and \
            Fn.wgrad_multi_eligible(job[0], job[2].spec)
        groups.setdefault((job[2].spec.cin, job[2].spec.cout) if ok else ('single', i), []).append(i)
    for idx in groups.values():
        for k in range(0, len(idx), Fn.WGRAD_MULTI_MAX):
            part = idx[k:k + Fn.WGRAD_MULTI_MAX]
            out = None
            if len(part) >= 2:
                out = Fn.conv_wgrad_multi([(defer[i][0], defer[i][1], defer[i][2].spec,
                                            defer[i][2].conv.weight, defer[i][3], defer[i][4],
                                            defer[i][2].conv.bias is not None, defer[i][5])
                                           for i in part])
            if out is None:
                for i in part:
                    x, dy, crt, tw, tb, aff = defer[i]
                    res[i] = _wgrad_now(x, dy, crt, tw, tb, aff)
            else:
                for i, (dw, db) in zip(part, out):
                    tw = defer[i][3]
                    res[i] = (tw if tw is not None and dw.data_ptr() == tw.data_ptr() else dw, db)
    return [res[g.index][g.which] if isinstance(g, _Deferred) else g for g in grads]


def _conv_backward(x, dy, crt, need_dx, mask_src=None, residual=None, mask_bits=None, bnb=None,
                   in_affine=None, defer=None):
    """-> dx (or None), dw (OIHW), dbias (or None); with bnb = (t, (scale, shift), mean, invstd) of
    the BatchNorm+ReLU in front of the conv: -> dx, dw, dbias, (partial, rows) or None -- dx then
    already carries that ReLU's mask (ConvRT.dgrad_bnb).  in_affine: the forward folded that
    BatchNorm + ReLU into the conv's loader (x = the BatchNorm's INPUT).  defer: a list -- the
    weight gradient is not launched here but queued (`_flush_wgrads`), dw / dbias are placeholders"""
    conv = crt.conv
    _trace_grad(conv, dy)
    # the gradients go straight into their flat all-reduce / optimizer bucket views when
    # GradientBuckets manages the parameters (no gather copy later)
    tw = grad_target(conv.weight)
    tb = grad_target(conv.bias) if conv.bias is not None else None
    half = dy.dtype != torch.float32
    if defer is not None and (WGRAD_STREAM or ((in_affine is None or half) and
                                               Fn.wgrad_multi_eligible(x, crt.spec))):
        defer.append((x, dy, crt, tw, tb, in_affine))
        dw, db = _Deferred(len(defer) - 1, 0), _Deferred(len(defer) - 1, 1)
    else:
        dw, db = _wgrad_now(x, dy, crt, tw, tb, in_affine)
    dx = None
    if bnb is not None and in_affine is not None and half:
        # 16-bit fold: plain data gradient; bn1's backward passes recompute the ReLU decisions from
        # the BatchNorm's input themselves (Fn.bn_bwd_aff)
        return crt.dgrad(dy, x.shape[2:], residual=residual), dw, db, None
    if bnb is not None:
        t, affine, mean, invstd = bnb
        r = crt.dgrad_bnb(dy, x.shape[2:], t, affine, mean, invstd, residual=residual,
                          force=in_affine is not None)
        if r is not None:
            return r[0], dw, db, (r[1], r[2])
        if in_affine is not None:
            raise _lib.EmsaError("folded BatchNorm: the data gradient has no fused reduction here")
        return crt.dgrad(dy, x.shape[2:], residual=residual), dw, db, None
    if need_dx:
        dx = crt.dgrad(dy, x.shape[2:], mask_src=mask_src, residual=residual,
                       mask_bits=mask_bits)
    return dx, dw, db


# ---------------------------------------------------------------------------------------------
# NonBottleneck1D block
# ---------------------------------------------------------------------------------------------
class NBt1DRT:
    def __init__(self, block):
        self.c31_1, self.c13_1 = ConvRT(block.conv3x1_1), ConvRT(block.conv1x3_1)
        self.c31_2, self.c13_2 = ConvRT(block.conv3x1_2), ConvRT(block.conv1x3_2)
        self.bn1, self.bn2 = BNRT(block.bn1), BNRT(block.bn2)
        if block.downsample is not None:
            self.cds, self.bnds = ConvRT(block.downsample[0]), BNRT(block.downsample[1])
        else:
            self.cds = self.bnds = None

    def params(self):
        ps = []
        for c in (self.c31_1, self.c13_1):
            ps += [c.conv.weight, c.conv.bias]
        ps += [self.bn1.bn.weight, self.bn1.bn.bias]
        for c in (self.c31_2, self.c13_2):
            ps += [c.conv.weight, c.conv.bias]
        ps += [self.bn2.bn.weight, self.bn2.bn.bias]
        if self.cds is not None:
            ps += [self.cds.conv.weight, self.bnds.bn.weight, self.bnds.bn.bias]
        return ps


class NBt1DFunction(Function):
    """conv3x1+b,ReLU -> conv1x3+b,BN,ReLU -> conv3x1+b,ReLU -> conv1x3+b,BN -> Dropout2d
    -> + identity -> ReLU   (reference block: SURVEY.md §8 a3)."""

    @staticmethod
    def forward(ctx, x, rt, drop, *params):
        x = Fn.as_act(x, dense=True)
        b = lambda c: c.conv.bias.detach()   # noqa: E731
        # y1, y3 = relu(conv): their ReLU masks go to the backward pass as bits (q1, q3; None when
        # the layer does not run on the Winograd kernel)
        y1, q1 = rt.c31_1.forward(x, bias=b(rt.c31_1), act=ACT_RELU, want_relu_bits=True)
        fold = rt.bn1.batch_stats() and rt.c31_2.folds_input_bn(y1)
        if fold:
            # bn1's normalise + ReLU happens in the loaders of conv3x1_2 (here) and of its weight
            # gradient; a2 = relu(bn1(y2)) is never written, no ReLU bit mask either (the data
            # gradient recomputes the decisions from y2)
            y2, stats = rt.c13_1.forward(y1, bias=b(rt.c13_1), want_stats=True)
            scale, shift, m1, is1 = rt.bn1.forward_stats(stats, y2.shape[0] * y2.shape[2] * y2.shape[3])
            m1._emsa_affine = (scale, shift)
            a2 = k1 = None
            y3, q3 = rt.c31_2.forward(y2, bias=b(rt.c31_2), act=ACT_RELU, want_relu_bits=True,
                                      in_affine=(scale, shift))
        else:
            a2, y2, m1, is1, k1 = _conv_bn_forward(y1, rt.c13_1, rt.bn1, ACT_RELU)
            y3, q3 = rt.c31_2.forward(a2, bias=b(rt.c31_2), act=ACT_RELU, want_relu_bits=True)
        if rt.cds is not None:
            idn, yd, md, isd, _ = _conv_bn_forward(x, rt.cds, rt.bnds, ACT_NONE)
        else:
            idn, yd, md, isd = x, None, None, None
        out, y4, m2, is2, k2 = _conv_bn_forward(y3, rt.c13_2, rt.bn2, ACT_RELU, drop=drop,
                                                residual=idn)
        if MASK_TRACE is not None:
            a2t = a2 if a2 is not None else Fn.bn_act(y2, scale, shift, None, None, ACT_RELU)
            for tag, t in (('nbt.y1', y1), ('nbt.a2', a2t), ('nbt.y3', y3), ('nbt.out', out)):
                _trace_mask(tag, t)
        ctx.rt, ctx.drop = rt, drop
        ctx.save_for_backward(x)
        # the ReLU masks of the two BatchNorm outputs travel as bit masks (k1, k2): the block
        # output itself is not kept for the backward pass
        ctx.saved = (y1, y2, a2, y3, y4, yd, m1, is1, m2, is2, md, isd, k1, k2, q1, q3)
        ctx.bn_train = (rt.bn1.batch_stats(), rt.bn2.batch_stats(),
                        rt.bnds.batch_stats() if rt.bnds is not None else False)
        return out

    @staticmethod
    @once_differentiable
    @_traced
    def backward(ctx, dout):
        rt, drop = ctx.rt, ctx.drop
        (x,) = ctx.saved_tensors
        y1, y2, a2, y3, y4, yd, m1, is1, m2, is2, md, isd, k1, k2, q1, q3 = ctx.saved
        ctx.saved = None
        dout = Fn.as_act(dout, dense=True)
        t1, t2, tds = ctx.bn_train
        need_dx = ctx.needs_input_grad[0]
        defer = []           # the block's weight gradients: launched together at the end

        # out = relu(bn2(y4)*drop + idn)
        dy4, dres, dg2, db2 = Fn.bn_bwd(dout, k2, y4, rt.bn2.bn.weight.detach(), m2, is2, drop,
                                        ACT_RELU, t2, want_dres=True, **_bn_targets(rt.bn2))
        # conv1x3_2 (input y3 = relu(.)): ReLU mask fused into the dgrad epilogue
        dz3, dw4, dbias4 = _conv_backward(y3, dy4, rt.c13_2, True, mask_src=y3, mask_bits=q3,
                                          defer=defer)
        # conv3x1_2 (input a2 = relu(bn1(y2)))
        # the data gradient's epilogue applies bn1's ReLU mask and emits bn1's backward sums (one
        # pass over da2 and y2 less); falls back to the separate reduction pass
        aff1 = getattr(m1, '_emsa_affine', None)
        if a2 is None:                    # folded forward: the conv's input is relu(bn1(y2))
            da2, dw3, dbias3, fused = _conv_backward(y2, dz3, rt.c31_2, True,
                                                     bnb=(y2, aff1, m1, is1), in_affine=aff1,
                                                     defer=defer if y2.dtype != torch.float32 else None)
        else:
            da2, dw3, dbias3, fused = _conv_backward(a2, dz3, rt.c31_2, True,
                                                     bnb=(y2, aff1, m1, is1), defer=defer)
        if fused is not None:
            dy2, dg1, db1 = Fn.bn_bwd_from_rows(da2, y2, rt.bn1.bn.weight.detach(), m1, is1,
                                                fused[0], fused[1], t1, **_bn_targets(rt.bn1))
        elif a2 is None:
            # 16-bit fold: no a2, no bit mask -- the passes recompute the decisions from y2
            dy2, dg1, db1 = Fn.bn_bwd_aff(da2, y2, rt.bn1.bn.weight.detach(), m1, is1, aff1,
                                          **_bn_targets(rt.bn1))
        else:
            dy2, _, dg1, db1 = Fn.bn_bwd(da2, k1, y2, rt.bn1.bn.weight.detach(), m1, is1, None,
                                         ACT_RELU, t1, want_dres=False, **_bn_targets(rt.bn1))
        dz1, dw2, dbias2 = _conv_backward(y1, dy2, rt.c13_1, True, mask_src=y1, mask_bits=q1,
                                          defer=defer)
        grads = []
        if rt.cds is None:
            # identity skip: dx = dgrad(conv3x1_1) + dres, add fused into the epilogue
            dx, dw1, dbias1 = _conv_backward(x, dz1, rt.c31_1, need_dx, residual=dres, defer=defer)
        else:
            dyd, _, dgd, dbd = Fn.bn_bwd(dres, None, yd, rt.bnds.bn.weight.detach(), md, isd,
                                         None, ACT_NONE, tds, want_dres=False,
                                         **_bn_targets(rt.bnds))
            dxd, dwd, _ = _conv_backward(x, dyd, rt.cds, need_dx)
            dx, dw1, dbias1 = _conv_backward(x, dz1, rt.c31_1, need_dx, residual=dxd, defer=defer)
        grads = [dw1, dbias1, dw2, dbias2, dg1, db1, dw3, dbias3, dw4, dbias4, dg2, db2]
        if rt.cds is not None:
            grads += [dwd, dgd, dbd]
        grads = _flush_wgrads(defer, grads)
        return (dx, None, None) + tuple(grads)


def _half_block_specs(rt):
    """the block's two half-blocks are conv3x1 -> conv1x3 pairs the fused kernel takes (stride 1, equal
    channel counts); the first half of a down-sampling block is not"""
    def ok(ca, cb):
        sa, sb = ca.spec, cb.spec
        return (Fn.rs_eligible(sa) and Fn.rs_eligible(sb) and (sa.kh, sa.kw) == (3, 1) and
                (sb.kh, sb.kw) == (1, 3) and sa.cout == sb.cin)
    return ok(rt.c31_1, rt.c13_1), ok(rt.c31_2, rt.c13_2)


def _half(xs, cas, cbs, bns, residuals):
    """one fused half-block launch over 1 or 2 tensor sets (Fn.nbt_half_block)"""
    b = lambda c: c.conv.bias.detach() if c.conv.bias is not None else None   # noqa: E731
    folds = [bn.folded() for bn in bns]
    dtype = xs[0].dtype
    return Fn.nbt_half_block(xs, [c.frag(dtype) for c in cas], [b(c) for c in cas],
                             [c.frag(dtype) for c in cbs], [b(c) for c in cbs],
                             [f[0] for f in folds], [f[1] for f in folds], residuals, ACT_RELU)


def nbt1d_eval(x, rt):
    """no-grad / eval fast path: both BatchNorms folded into the conv epilogues (4 launches; 2 where
    the fused half-block kernel takes the map: small-batch 16-bit inference at C = 64 / 128)."""
    b = lambda c: c.conv.bias.detach()   # noqa: E731
    h1, h2 = _half_block_specs(rt)
    if h1 and Fn.half_block_ok(x, rt.c31_1.spec.cin):
        (a2,) = _half([x], [rt.c31_1], [rt.c13_1], [rt.bn1], [None])
    else:
        y1 = rt.c31_1.forward(x, bias=b(rt.c31_1), act=ACT_RELU)
        s1, t1 = rt.bn1.folded()
        a2 = rt.c13_1.forward(y1, bias=b(rt.c13_1), scale=s1, shift=t1, act=ACT_RELU)
    if rt.cds is not None:
        sd, td = rt.bnds.folded()
        idn = rt.cds.forward(x, scale=sd, shift=td)
    else:
        idn = x
    if h2 and Fn.half_block_ok(a2, rt.c31_2.spec.cin) and Fn.ld_of(idn) == idn.shape[1]:
        return _half([a2], [rt.c31_2], [rt.c13_2], [rt.bn2], [idn])[0]
    y3 = rt.c31_2.forward(a2, bias=b(rt.c31_2), act=ACT_RELU)
    s2, t2 = rt.bn2.folded()
    return rt.c13_2.forward(y3, bias=b(rt.c13_2), scale=s2, shift=t2, residual=idn, act=ACT_RELU)


def _conv_pair(xa, xb, ca, cb, **kw):
    """the same conv layer of two twin modules on their two inputs: one twin launch -- of the
    register-stationary kernel where it takes the geometry (Fn.conv_fwd_pair), else of the implicit
    GEMM (Fn.conv_igemm_pair) -- or two launches (fp32 storage).
    kw: per-operand pairs (bias=, scale=, shift=, residual=) and act="""
    act = kw.pop('act', ACT_NONE)
    pr = {k: kw.get(k, (None, None)) for k in ('bias', 'scale', 'shift', 'residual')}
    if ca.spec_key() == cb.spec_key():
        r = Fn.conv_fwd_pair((xa, xb), (ca.frag(xa.dtype), cb.frag(xb.dtype)), ca.spec,
                             biases=pr['bias'], scales=pr['scale'], shifts=pr['shift'],
                             residuals=pr['residual'], act=act)
        if r is None and xa.dtype != torch.float32:
            r = Fn.conv_igemm_pair((xa, xb), (ca.packed(xa.dtype), cb.packed(xb.dtype)), ca.spec,
                                   biases=pr['bias'], scales=pr['scale'], shifts=pr['shift'],
                                   residuals=pr['residual'], act=act)
        if r is not None:
            return r
    one = lambda i, x, c: c.forward(x, act=act, **{k: v[i] for k, v in pr.items() if v[i] is not None})  # noqa: E731
    return one(0, xa, ca), one(1, xb, cb)


def conv_bn_act_eval_pair(xa, xb, ma, mb):
    """conv_bn_act_eval of the same ConvNormAct layer of two twin modules (nn.ConvNormAct: `_crt`,
    `_brt`, `act`) as one twin launch; xa may be xb (the skip both decoders read)"""
    if ma.act != mb.act:
        raise _lib.EmsaError("twin ConvNormAct layers differ in their activation")
    (sa, ta), (sb, tb) = ma._brt.folded(), mb._brt.folded()
    return _conv_pair(xa, xb, ma._crt, mb._crt, scale=(sa, sb), shift=(ta, tb), act=ma.act)


def nbt1d_eval_pair(xa, xb, ra, rb):
    """nbt1d_eval of the SAME block of two twin networks (rgb | depth encoder, semantic | instance
    decoder) in lockstep: every conv both blocks share a geometry for is one twin launch (4 instead
    of 8 launches per block pair; at batch 1 a launch is its fixed cost).  Results == nbt1d_eval."""
    b = lambda c: c.conv.bias.detach()   # noqa: E731
    if (ra.cds is None) != (rb.cds is None):
        raise _lib.EmsaError("twin NBt1D blocks differ in their skip path")
    h1a, h2a = _half_block_specs(ra)
    h1b, h2b = _half_block_specs(rb)
    same1 = ra.c31_1.spec_key() == rb.c31_1.spec_key() and ra.c13_1.spec_key() == rb.c13_1.spec_key()
    same2 = ra.c31_2.spec_key() == rb.c31_2.spec_key() and ra.c13_2.spec_key() == rb.c13_2.spec_key()
    twin_ok = xa.shape == xb.shape and xa.dtype == xb.dtype
    if twin_ok and h1a and h1b and same1 and Fn.half_block_ok(xa, ra.c31_1.spec.cin) and \
            Fn.half_block_ok(xb, rb.c31_1.spec.cin):
        # one launch for both twins' first half-block (fused conv3x1 -> conv1x3, csrc/conv_hb.hip)
        a2a, a2b = _half([xa, xb], [ra.c31_1, rb.c31_1], [ra.c13_1, rb.c13_1], [ra.bn1, rb.bn1],
                         [None, None])
    else:
        y1a, y1b = _conv_pair(xa, xb, ra.c31_1, rb.c31_1, bias=(b(ra.c31_1), b(rb.c31_1)), act=ACT_RELU)
        (s1a, t1a), (s1b, t1b) = ra.bn1.folded(), rb.bn1.folded()
        a2a, a2b = _conv_pair(y1a, y1b, ra.c13_1, rb.c13_1, bias=(b(ra.c13_1), b(rb.c13_1)),
                              scale=(s1a, s1b), shift=(t1a, t1b), act=ACT_RELU)
    if ra.cds is not None:
        (sda, tda), (sdb, tdb) = ra.bnds.folded(), rb.bnds.folded()
        ida, idb = _conv_pair(xa, xb, ra.cds, rb.cds, scale=(sda, sdb), shift=(tda, tdb))
    else:
        ida, idb = xa, xb
    if h2a and h2b and same2 and a2a.shape == a2b.shape and Fn.half_block_ok(a2a, ra.c31_2.spec.cin) and \
            Fn.half_block_ok(a2b, rb.c31_2.spec.cin) and Fn.ld_of(ida) == ida.shape[1] and \
            Fn.ld_of(idb) == idb.shape[1]:
        return _half([a2a, a2b], [ra.c31_2, rb.c31_2], [ra.c13_2, rb.c13_2], [ra.bn2, rb.bn2],
                     [ida, idb])
    y3a, y3b = _conv_pair(a2a, a2b, ra.c31_2, rb.c31_2, bias=(b(ra.c31_2), b(rb.c31_2)), act=ACT_RELU)
    (s2a, t2a), (s2b, t2b) = ra.bn2.folded(), rb.bn2.folded()
    return _conv_pair(y3a, y3b, ra.c13_2, rb.c13_2, bias=(b(ra.c13_2), b(rb.c13_2)),
                      scale=(s2a, s2b), shift=(t2a, t2b), residual=(ida, idb), act=ACT_RELU)


# ---------------------------------------------------------------------------------------------
# conv + BN (+ReLU)
# ---------------------------------------------------------------------------------------------
class ConvBNActFunction(Function):
    @staticmethod
    def forward(ctx, x, crt, brt, act, weight, gamma, beta):
        x = Fn.as_act(x)
        out, y, mean, invstd, mask = _conv_bn_forward(x, crt, brt, act)
        if act == ACT_RELU:
            _trace_mask('conv_bn_act', out)
        ctx.crt, ctx.brt, ctx.act = crt, brt, act
        ctx.save_for_backward(x)
        ctx.saved = (y, mean, invstd, mask)
        ctx.bn_train = brt.batch_stats()
        return out

    @staticmethod
    @once_differentiable
    @_traced
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        y, mean, invstd, mask = ctx.saved
        ctx.saved = None
        dout = Fn.as_act(dout, dense=True)
        dy, _, dg, db = Fn.bn_bwd(dout, mask, y, ctx.brt.bn.weight.detach(), mean, invstd, None,
                                  ctx.act, ctx.bn_train, want_dres=False, **_bn_targets(ctx.brt))
        dx, dw, _ = _conv_backward(x, dy, ctx.crt, ctx.needs_input_grad[0])
        return dx, None, None, None, dw, dg, db


class ConvBNAddActFunction(Function):
    """act(bn(conv(x)) + residual): the closing conv of a basic / bottleneck ResNet block"""

    @staticmethod
    def forward(ctx, x, residual, crt, brt, act, weight, gamma, beta):
        x, residual = Fn.as_act(x), Fn.as_act(residual, dense=True)
        out, y, mean, invstd, mask = _conv_bn_forward(x, crt, brt, act, residual=residual)
        if act == ACT_RELU:
            _trace_mask('conv_bn_add_act', out)
        ctx.crt, ctx.brt, ctx.act = crt, brt, act
        ctx.save_for_backward(x)
        ctx.saved = (y, mean, invstd, mask)
        ctx.bn_train = brt.batch_stats()
        return out

    @staticmethod
    @once_differentiable
    @_traced
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        y, mean, invstd, mask = ctx.saved
        ctx.saved = None
        dout = Fn.as_act(dout, dense=True)
        dy, dres, dg, db = Fn.bn_bwd(dout, mask, y, ctx.brt.bn.weight.detach(), mean, invstd, None,
                                     ctx.act, ctx.bn_train, want_dres=ctx.needs_input_grad[1],
                                     **_bn_targets(ctx.brt))
        dx, dw, _ = _conv_backward(x, dy, ctx.crt, ctx.needs_input_grad[0])
        return dx, dres, None, None, None, dw, dg, db


def conv_bn_act_eval(x, crt, brt, act, residual=None):
    s, t = brt.folded()
    return crt.forward(x, scale=s, shift=t, act=act, residual=residual)


# ---------------------------------------------------------------------------------------------
# plain conv (+bias) whose packed weight may host several parameters (padding / block-diagonal)
# ---------------------------------------------------------------------------------------------
class MultiConvRT:
    """One GEMM-level convolution assembled from several nn.Conv2d / nn.Linear parameters.
    `placements` = [(module, cout_off, cin_off)], total (cout_total, cin_total) multiples of 4."""

    def __init__(self, placements, cout_total, cin_total, kernel, padding):
        self.placements = placements
        self.spec = Fn.ConvSpec(cin_total, cout_total, kernel, 1, padding)
        self._key = None
        self._wp = None
        self._bias = None
        self._u = None
        self._h = {}                 # 16-bit packs: dtype -> (key, wp, bias)
        self._hd = {}                # data-gradient packs: dtype -> (key, wpd, Winograd U or None)
        self.has_bias = any(m.bias is not None for m, _, _ in placements)
        # 3x3 stride-1 merged convs (the task heads) run on the Winograd kernel as well
        self.wino = Fn.wino_eligible(self.spec) and os.environ.get('EMSA_WINO', '1') != '0'

    def _w4(self, m):
        w = m.weight
        return w if w.dim() == 4 else w[:, :, None, None]

    def _key_now(self):
        return tuple((m.weight._version, m.weight.data_ptr(),
                      m.bias._version if m.bias is not None else 0) for m, _, _ in self.placements)

    def _bias_vec(self, dev):
        bias = torch.zeros(self.spec.cout, device=dev, dtype=torch.float32)
        for m, co, _ in self.placements:
            if m.bias is not None:
                bias[co:co + m.bias.shape[0]].copy_(m.bias.detach())
        return bias

    def packed(self):
        key = self._key_now()
        if key != self._key:
            s = self.spec
            wp = torch.zeros(s.kh * s.kw * s.cout * s.cin, device=self.placements[0][0].weight.device,
                             dtype=torch.float32)
            for m, co, ci in self.placements:
                Fn.pack_weight(self._w4(m).detach(), 'fwd', s.cout, co, s.cin, ci, out=wp)
            bias = self._bias_vec(wp.device) if self.has_bias else None
            self._wp, self._bias, self._key = wp, bias, key
            self._u = Fn.pack_wino_packed(wp, s.cout, s.cin, Fn.wino_rows(s), flip=False) \
                if self.wino else None
        return self._wp, self._bias

    def packed_t(self, dtype):
        """16-bit forward operand [tap][cout][cin] (+ fp32 bias vector)"""
        key = self._key_now()
        ent = self._h.get(dtype)
        if ent is None or ent[0] != key:
            s = self.spec
            dev = self.placements[0][0].weight.device
            wp = torch.zeros(s.kh * s.kw * s.cout * s.cin, device=dev, dtype=dtype)
            for m, co, ci in self.placements:
                Fn.pack_weight_t(self._w4(m).detach(), dtype, True, False, s.cout, co, s.cin, ci,
                                 out_fwd=wp)
            ent = (key, wp, self._bias_vec(dev) if self.has_bias else None)
            self._h[dtype] = ent
        return ent[1], ent[2]

    def forward(self, x):
        Fn.prof_flops(self.real_flops(x))
        if x.dtype != torch.float32:
            wp, bias = self.packed_t(x.dtype)
            return Fn.conv_fwd(x, wp, self.spec, bias=bias)
        wp, bias = self.packed()
        if self.wino:
            return Fn.conv_fwd(x, None, self.spec, bias=bias, wino_u=self._u)
        return Fn.conv_fwd(x, wp, self.spec, bias=bias)

    def real_flops(self, x):
        """direct-convolution FLOPs of the REAL (un-padded, un-merged) convolutions on `x`"""
        n, _, h, w = x.shape
        return 2.0 * n * h * w * sum(m.weight.numel() for m, _, _ in self.placements)

    def dgrad(self, dy, in_hw):
        Fn.prof_flops(self.real_flops(dy))
        # the data-gradient operands are rebuilt only when a parameter moved (they used to be
        # re-packed at every call: a zero-fill and 2-4 small kernels per head and step)
        key = self._key_now()
        ent = self._hd.get(dy.dtype)
        if ent is None or ent[0] != key:
            wpd = self.packed_dgrad(dy.dtype)
            ud = None
            if self.wino and dy.dtype == torch.float32:
                s = self.spec
                ud = Fn.pack_wino_packed(wpd, s.cin, s.cout, Fn.wino_rows(s), flip=True)
            ent = (key, wpd, ud)
            self._hd[dy.dtype] = ent
        _, wpd, ud = ent
        if ud is not None:
            return Fn.conv_dgrad(dy, None, self.spec, in_hw, wino_u=ud)
        return Fn.conv_dgrad(dy, wpd, self.spec, in_hw)

    def packed_dgrad(self, dtype=torch.float32):
        s = self.spec
        wp = torch.zeros(s.kh * s.kw * s.cout * s.cin, device=self.placements[0][0].weight.device,
                         dtype=dtype)
        for m, co, ci in self.placements:
            if dtype == torch.float32:
                Fn.pack_weight(self._w4(m).detach(), 'dgrad', s.cout, co, s.cin, ci, out=wp)
            else:
                Fn.pack_weight_t(self._w4(m).detach(), dtype, False, True, s.cout, co, s.cin, ci,
                                 out_dgrad=wp)
        return wp

    def params(self):
        ps = []
        for m, _, _ in self.placements:
            ps.append(m.weight)
            if m.bias is not None:
                ps.append(m.bias)
        return ps


class MultiConvFunction(Function):
    @staticmethod
    def forward(ctx, x, rt, *params):
        x = Fn.as_act(x)
        y = rt.forward(x)
        ctx.rt = rt
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @once_differentiable
    @_traced
    def backward(ctx, dy):
        rt = ctx.rt
        (x,) = ctx.saved_tensors
        dy = Fn.as_act(dy)
        s = rt.spec
        Fn.prof_flops(rt.real_flops(x))
        if GRAD_TRACE is not None:
            for m, co, _ in rt.placements:            # (merged convs: each module's own channels)
                _trace_grad(m, dy[:, co:co + m.weight.shape[0]])
        dwp, db, _ = Fn.conv_wgrad(x, dy, s, rt.has_bias)
        grads = []
        for m, co, ci in rt.placements:
            w4 = rt._w4(m)
            tw = grad_target(m.weight)
            dw = Fn.unpack_wgrad(dwp, w4, s.cout, co, s.cin, ci,
                                 out=tw.view(w4.shape) if tw is not None else None)
            grads.append(tw if tw is not None else dw.reshape(m.weight.shape))
            if m.bias is not None:
                grads.append(db[co:co + m.bias.shape[0]].clone())
        dx = None
        if ctx.needs_input_grad[0]:
            dx = rt.dgrad(dy, x.shape[2:])
        return (dx, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------
# stem: 7x7/2 conv + BN + ReLU on the NCHW network input
# ---------------------------------------------------------------------------------------------
class StemRT:
    def __init__(self, conv, bn):
        self.conv, self.brt = conv, BNRT(bn)
        self.spec = Fn.StemSpec(conv.in_channels, conv.out_channels)
        self._packs = {}             # dtype -> (key, packed weights)

    def packed(self, dtype=torch.float32):
        w = self.conv.weight
        key = (w._version, w.data_ptr())
        ent = self._packs.get(dtype)
        if ent is None or ent[0] != key:
            ent = (key, Fn.stem_pack_weight(w.detach(), dtype))
            self._packs[dtype] = ent
        return ent[1]


class StemFunction(Function):
    @staticmethod
    def forward(ctx, x_nchw, rt, weight, gamma, beta, bias=None, dtype=torch.float32):
        n, c, h, w = x_nchw.shape
        xp = Fn.stem_pack_input(x_nchw.detach().float(), dtype)
        brt = rt.brt
        b = bias.detach() if bias is not None else None      # [U] Spec.STEM_BIAS
        wpk = rt.packed(dtype)
        if brt.batch_stats():
            y, stats = Fn.stem_fwd(xp, wpk, rt.spec, n, h, w, want_stats=True, bias=b)
            count = y.shape[0] * y.shape[2] * y.shape[3]
        else:
            y, _ = Fn.stem_fwd(xp, wpk, rt.spec, n, h, w, want_stats=False, bias=b)
            stats, count = None, 0
        scale, shift, mean, invstd = brt.forward_stats(stats, count)
        out, mask = Fn.bn_act(y, scale, shift, None, None, ACT_RELU, want_mask=True)
        _trace_mask('stem', out)
        ctx.rt = rt
        ctx.hw = (n, h, w)
        ctx.saved = (xp, y, mean, invstd, mask)
        ctx.bn_train = brt.batch_stats()
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    @once_differentiable
    @_traced
    def backward(ctx, dout):
        rt = ctx.rt
        xp, y, mean, invstd, mask = ctx.saved
        ctx.saved = None
        n, h, w = ctx.hw
        dout = Fn.as_act(dout, dense=True)
        dy, _, dg, db = Fn.bn_bwd(dout, mask, y, rt.brt.bn.weight.detach(), mean, invstd, None,
                                  ACT_RELU, ctx.bn_train, want_dres=False, **_bn_targets(rt.brt))
        _trace_grad(rt.conv, dy)
        res = Fn.stem_wgrad(xp, dy, rt.spec, n, h, w, rt.conv.weight,
                            out=grad_target(rt.conv.weight), want_bias=ctx.has_bias)
        dw, dbias = res if ctx.has_bias else (res, None)
        # gradient w.r.t. the network input is not produced (the reference never needs it:
        # /root/reference/main.py:597-599 back-propagates into parameters only)
        return None, None, dw, dg, db, dbias, None


def stem_eval(x_nchw, rt, dtype=torch.float32):
    n, c, h, w = x_nchw.shape
    xp = Fn.stem_pack_input(x_nchw.float(), dtype)
    s, t = rt.brt.folded()
    b = rt.conv.bias.detach() if rt.conv.bias is not None else None
    return Fn.stem_fwd_folded(xp, rt.packed(dtype), rt.spec, n, h, w, s, t, bias=b)


# ---------------------------------------------------------------------------------------------
# max pool
# ---------------------------------------------------------------------------------------------
class MaxPoolFunction(Function):
    @staticmethod
    def forward(ctx, x):
        x = Fn.as_act(x, dense=True)
        y, idx = Fn.maxpool_fwd(x)
        ctx.idx, ctx.hw = idx, tuple(x.shape[2:])
        return y

    @staticmethod
    @once_differentiable
    @_traced
    def backward(ctx, dy):
        return Fn.maxpool_bwd(Fn.as_act(dy, dense=True), ctx.idx, ctx.hw)


# ---------------------------------------------------------------------------------------------
# SE-add fusion
# ---------------------------------------------------------------------------------------------
# EMSA_SE_PAIR=0: the squeeze / excitation of the two inputs of an SE-add fusion as separate launches
SE_PAIR = os.environ.get('EMSA_SE_PAIR', '1') != '0'


def _se_params(se):
    f0, f2 = se.fc[0], se.fc[2]
    return [f0.weight, f0.bias, f2.weight, f2.bias]


class SEAddFunction(Function):
    """out = rgb * SE_rgb(rgb) + depth * SE_depth(depth)   ('se-add-uni-rgb'), and the depth stream
    handed on as a second output.  The depth stream has TWO consumers -- this fusion and the next
    depth stage -- so autograd used to add their two gradients with a torch kernel per stage (the
    largest one on the 629 MB stem output: 0.33 ms).  With the pass-through output this Function is
    the depth tensor's only consumer: the next stage's gradient arrives as `ddepth_next` and is
    added inside the SE backward kernel (`dx_extra` of emsa_se_scale_bwd_apply), one extra read
    instead of a read-read-write pass."""

    @staticmethod
    def forward(ctx, rgb, depth, w1r, b1r, w2r, b2r, w1d, b1d, w2d, b2d):
        rgb, depth = Fn.as_act(rgb, dense=True), Fn.as_act(depth, dense=True)
        flat = lambda w: w.detach().reshape(w.shape[0], -1)   # noqa: E731
        if SE_PAIR:
            # both squeezes in one launch, both excitation MLPs in one (7 launches -> 3 per fusion)
            gr, gd, hr, hd, sr, sd = Fn.se_pair_fwd(
                rgb, depth, (flat(w1r), b1r.detach(), flat(w2r), b2r.detach()),
                (flat(w1d), b1d.detach(), flat(w2d), b2d.detach()))
        else:
            gr, gd = Fn.channel_mean(rgb), Fn.channel_mean(depth)
            hr, sr = Fn.se_mlp_fwd(gr, flat(w1r), b1r.detach(), flat(w2r), b2r.detach())
            hd, sd = Fn.se_mlp_fwd(gd, flat(w1d), b1d.detach(), flat(w2d), b2d.detach())
        _trace_mask('se.rgb', hr)
        _trace_mask('se.depth', hd)
        out = Fn.se_scale_add(rgb, sr, depth, sd)
        # an unused pass-through depth output (last stage, cut boundaries) arrives as None in
        # backward instead of a materialised zeros tensor + one more read in the kernel
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(rgb, depth)
        ctx.saved = (gr, gd, hr, sr, hd, sd, flat(w1r), flat(w2r), flat(w1d), flat(w2d))
        ctx.shapes = (w1r.shape, w2r.shape)
        return out, depth[:]              # (an alias: same memory, this node as its grad_fn)

    @staticmethod
    @once_differentiable
    @_traced
    def backward(ctx, dout, ddepth_next):
        rgb, depth = ctx.saved_tensors
        gr, gd, hr, sr, hd, sd, w1r, w2r, w1d, w2d = ctx.saved
        ctx.saved = None
        if dout is None:                     # (only the pass-through output was used downstream)
            dout = torch.zeros_like(rgb)
        dout = Fn.as_act(dout, dense=True)
        if ddepth_next is not None:
            ddepth_next = Fn.as_act(ddepth_next, dense=True)
            if ddepth_next.dtype != dout.dtype:
                ddepth_next = Fn.cast(ddepth_next, dout.dtype)
        s1, s2 = ctx.shapes
        res = []
        for x, g, h, s, w1, w2, extra in ((rgb, gr, hr, sr, w1r, w2r, None),
                                          (depth, gd, hd, sd, w1d, w2d, ddepth_next)):
            ds = Fn.se_scale_bwd_reduce(dout, x)
            dgap, dw1, db1, dw2, db2 = Fn.se_mlp_bwd(g, w1, w2, h, s, ds)
            dx = Fn.se_scale_bwd_apply(dout, s, dgap, extra)
            res.append((dx, dw1.reshape(s1), db1, dw2.reshape(s2), db2))
        (dr, a1, a2, a3, a4), (dd, e1, e2, e3, e4) = res
        return dr, dd, a1, a2, a3, a4, e1, e2, e3, e4


# ---------------------------------------------------------------------------------------------
# learned upsampling (nearest x2 + depth-wise 3x3) with optional fused skip add
# ---------------------------------------------------------------------------------------------
class UpsampleDWFunction(Function):
    """`wdw`/`bias` are the (possibly zero-padded) [c,1,3,3] / [c] tensors the kernel reads.
    out_f32: the result is written as fp32 from 16-bit features (the last up-sampling of a head
    produces the model's fp32 output directly; its cotangent arrives as fp32 too)."""

    @staticmethod
    def forward(ctx, x, wdw, bias, skip, out_f32=False):
        x = Fn.as_act(x, dense=True)
        if skip is not None:
            skip = Fn.as_act(skip, dense=True)
        w = wdw.detach().contiguous()
        y = Fn.up2x_dw_fwd(x, w, bias.detach() if bias is not None else None, skip,
                           out_f32=out_f32 and x.dtype != torch.float32)
        ctx.save_for_backward(x, w)
        ctx.has = (bias is not None, skip is not None)
        ctx.wshape = wdw.shape
        return y

    @staticmethod
    @once_differentiable
    @_traced
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = Fn.as_act(dy, dense=True)
        dx, dw, db = Fn.up2x_dw_bwd(dy, x, w, need_dx=ctx.needs_input_grad[0])
        has_bias, has_skip = ctx.has
        dskip = None
        if has_skip:
            dskip = dy if dy.dtype == x.dtype else Fn.cast(dy, x.dtype)
        return dx, dw.reshape(ctx.wshape), (db if has_bias else None), dskip, None


class PlainUpsampleFunction(Function):
    """x2 up-sampling without weights -- 'nearest' / 'bilinear' (align_corners=False) of
    `--{semantic,instance,normal}-decoder-upsampling` / `--upsampling-prediction`
    (/root/reference/emsanet/args.py:280-298,363-372) -- plus the optional skip add; out_f32 as in
    UpsampleDWFunction (16-bit features -> the model's fp32 output)."""

    @staticmethod
    def forward(ctx, x, mode, skip, out_f32=False):
        if mode not in ('bilinear', 'nearest'):
            raise NotImplementedError(f"upsampling '{mode}'")
        x = Fn.as_act(x, dense=True)
        ctx.mode, ctx.x_dtype, ctx.hw = mode, x.dtype, tuple(x.shape[2:])
        if out_f32 and x.dtype != torch.float32:
            x = Fn.cast(x, torch.float32)
        n, c, h, w = x.shape
        y = Fn.act_empty(n, c, 2 * h, 2 * w, x.device, dtype=x.dtype)
        (Fn.bilinear_fwd if mode == 'bilinear' else Fn.nearest_fwd)(x, y)
        ctx.has_skip = skip is not None
        if skip is not None:
            skip = Fn.as_act(skip, dense=True)
            y = Fn.add(y, skip if skip.dtype == y.dtype else Fn.cast(skip, y.dtype))
        return y

    @staticmethod
    @once_differentiable
    @_traced
    def backward(ctx, dy):
        dy = Fn.as_act(dy, dense=True)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = (Fn.bilinear_bwd if ctx.mode == 'bilinear' else Fn.nearest_bwd)(dy, ctx.hw)
            if dx.dtype != ctx.x_dtype:
                dx = Fn.cast(dx, ctx.x_dtype)
        dskip = None
        if ctx.has_skip:
            dskip = dy if dy.dtype == ctx.x_dtype else Fn.cast(dy, ctx.x_dtype)
        return dx, None, dskip, None


# ---------------------------------------------------------------------------------------------
# pyramid pooling pieces
# ---------------------------------------------------------------------------------------------
class AdaptiveAvgPoolFunction(Function):
    @staticmethod
    def forward(ctx, x, bins):
        x = Fn.as_act(x, dense=True)
        ctx.bins, ctx.shape = bins, tuple(x.shape)
        return Fn.adaptive_avgpool_fwd(x, bins)

    @staticmethod
    @once_differentiable
    @_traced
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        dy = Fn.as_act(dy, dense=True)
        dx = Fn.act_empty(n, c, h, w, dy.device, dtype=dy.dtype)
        Fn.adaptive_avgpool_bwd(dy, dx, ctx.bins, accumulate=False)
        return dx, None


class PPMConcatFunction(Function):
    """cat([x, up(y_i) ...], dim=1) written straight into one NHWC buffer; up = bilinear
    (align_corners=False) or -- a trailing 'nearest' argument -- nearest, the two choices of
    `--upsampling-context-module` (/root/reference/emsanet/args.py:250-256)."""

    @staticmethod
    def forward(ctx, x, *ys):
        mode = 'bilinear'
        if ys and isinstance(ys[-1], str):
            mode, ys = ys[-1], ys[:-1]
            ctx.has_mode = True
        else:
            ctx.has_mode = False
        if mode not in ('bilinear', 'nearest'):
            raise NotImplementedError(f"upsampling_context_module='{mode}'")
        up = Fn.bilinear_fwd if mode == 'bilinear' else Fn.nearest_fwd
        x = Fn.as_act(x)
        ys = [Fn.as_act(y, dense=True) for y in ys]
        n, c, h, w = x.shape
        total = c + sum(y.shape[1] for y in ys)
        buf = Fn.act_empty(n, total, h, w, x.device, dtype=x.dtype)
        Fn.copy_channels(x, buf[:, :c])
        off = c
        for y in ys:
            up(y, buf[:, off:off + y.shape[1]])
            off += y.shape[1]
        ctx.meta = (c, [tuple(y.shape) for y in ys], mode)
        return buf

    @staticmethod
    @once_differentiable
    @_traced
    def backward(ctx, dbuf):
        dbuf = Fn.as_act(dbuf)
        c, yshapes, mode = ctx.meta
        down = Fn.bilinear_bwd if mode == 'bilinear' else Fn.nearest_bwd
        n, _, h, w = dbuf.shape
        dx = Fn.act_empty(n, c, h, w, dbuf.device, dtype=dbuf.dtype)
        Fn.copy_channels(dbuf[:, :c], dx)
        off, dys = c, []
        for ys in yshapes:
            dys.append(down(dbuf[:, off:off + ys[1]], ys[2:]))
            off += ys[1]
        return (dx,) + tuple(dys) + ((None,) if ctx.has_mode else ())


# ---------------------------------------------------------------------------------------------
# head activations
# ---------------------------------------------------------------------------------------------
class HeadActFunction(Function):
    """sigmoid / tanh (/ L2 normalisation of the orientation pair) on the leading channels of the
    (channel-padded) instance head output and the split into the task tensors.  The outputs are
    channel-slice VIEWS of one activated tensor; backward gathers their gradients into one padded
    tensor with strided channel copies -- slicing outside (autograd's SliceBackward) materialised
    a zero-filled full-size tensor per task and added them up: three fills, three copies and two
    adds of the 315 MB full-resolution tensor."""

    @staticmethod
    def forward(ctx, x, n_sig, n_tanh, sizes, n_norm=0):
        x = Fn.as_act(x, dense=True)
        y = Fn.head_act_fwd(x, n_sig, n_tanh, n_norm)
        ctx.save_for_backward(y, x if n_norm else None)
        ctx.cfg = (n_sig, n_tanh, n_norm)
        ctx.dtype = x.dtype
        ctx.sizes = tuple(sizes)
        outs, o = [], 0
        for sz in sizes:
            outs.append(y[:, o:o + sz])
            o += sz
        return tuple(outs)

    @staticmethod
    @once_differentiable
    @_traced
    def backward(ctx, *dys):
        y, x = ctx.saved_tensors
        n, c, h, w = y.shape
        # the three task gradients are gathered INSIDE the backward kernel (round 5: the padded
        # 8-channel gradient tensor -- 315 MB at bs 32 -- is no longer written and read again)
        dx = Fn.head_act_bwd_gather(dys, ctx.sizes, y, *ctx.cfg, x=x, dtype=ctx.dtype)
        if dx is not None:
            return dx, None, None, None, None
        dy = Fn.act_empty(n, c, h, w, y.device)
        o = 0
        for sz, g in zip(ctx.sizes, dys):
            Fn.copy_channels(Fn.as_act(g.float()), dy[:, o:o + sz])
            o += sz
        if o < c:
            dy[:, o:].zero_()                      # padding channels carry no gradient
        return Fn.head_act_bwd(dy, y, *ctx.cfg, x=x, dtype=ctx.dtype), None, None, None, None


# ---------------------------------------------------------------------------------------------
# storage-type boundary: small 16-bit tensors leave the engine as fp32 (side outputs, scene logits)
# ---------------------------------------------------------------------------------------------
class CastFunction(Function):
    @staticmethod
    def forward(ctx, x, dtype):
        ctx.src = x.dtype
        return Fn.cast(x, dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        return Fn.cast(dy, ctx.src), None


def to_float(x):
    """fp32 view of an engine tensor (identity for the fp32 engine)"""
    return x if x.dtype == torch.float32 else CastFunction.apply(x, torch.float32)
