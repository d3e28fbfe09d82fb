# -*- coding: utf-8 -*-
"""
Semantic-segmentation loss on device.

`CrossEntropyLossSemantic(weights, label_smoothing=0.0, weighted_reduction=True)` mirrors the
constructor and call convention the reference exercises in
/root/reference/emsanet/tests/test_semantic_loss.py:68-97 (`loss_object(pred_scales,
target_scales)[i][0]` is the loss of scale i), i.e. what `task_helper.training_step` applies to
the semantic head and its side outputs (/root/reference/main.py:131-141).  Arithmetic runs in
libemsanet_hip.so (csrc/loss.hip): logits are read once in forward, once in backward.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from . import functional as Fn
from ._lib import check


class SemanticCEFunction(Function):
    @staticmethod
    def forward(ctx, logits, target, weights):
        logits = Fn.as_act(logits)
        n, c, h, w = logits.shape
        if target.dtype != torch.int64:
            target = target.long()
        target = target.contiguous()
        if target.numel() != n * h * w:
            raise _lib.EmsaError("target must be (N,H,W) matching the logits")
        L = _lib.lib()
        pixels = n * h * w
        partial = Fn._empty((2 * L.emsa_ce_semantic_blocks(pixels),), logits.device)
        out = Fn._empty((2,), logits.device)
        weights = weights.detach().float().contiguous()
        check(L.emsa_ce_semantic_fwd(logits.data_ptr(), Fn.ld_of(logits), target.data_ptr(),
                                     weights.data_ptr(), c, pixels, partial.data_ptr(),
                                     out.data_ptr(), Fn._stream()), 'emsa_ce_semantic_fwd')
        ctx.save_for_backward(logits, target, weights, out)
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        logits, target, weights, out = ctx.saved_tensors
        n, c, h, w = logits.shape
        d = Fn.act_empty(n, Fn.pad4(c), h, w, logits.device)
        g = gout.detach().float().reshape(1).contiguous()
        check(_lib.lib().emsa_ce_semantic_bwd(logits.data_ptr(), Fn.ld_of(logits),
                                              target.data_ptr(), weights.data_ptr(), c,
                                              n * h * w, out.data_ptr(), g.data_ptr(),
                                              d.data_ptr(), Fn.ld_of(d), Fn._stream()),
              'emsa_ce_semantic_bwd')
        return d[:, :c], None, None


class CrossEntropyLossSemantic(torch.nn.Module):
    def __init__(self, weights, label_smoothing=0.0, weighted_reduction=True):
        super().__init__()
        if label_smoothing != 0.0:
            raise NotImplementedError("label smoothing: only the reference default 0.0 "
                                      "(emsanet/args.py:724-729) is pinned and implemented")
        if not weighted_reduction:
            raise NotImplementedError("only the weighted reduction the reference trains with")
        self.register_buffer('weights', torch.as_tensor(weights, dtype=torch.float32))

    def forward(self, input_scales, target_scales):
        """-> [(loss_scale_i, ), ...] like the reference's loss object (test_semantic_loss.py:93-97)"""
        return [(SemanticCEFunction.apply(x, t, self.weights),)
                for x, t in zip(input_scales, target_scales)]
