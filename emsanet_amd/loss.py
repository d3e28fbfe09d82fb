# -*- coding: utf-8 -*-
"""
Training losses on device: what `task_helper.training_step` +
`loss_weighting.reduce_losses` compute in the reference's training step
(/root/reference/main.py:129-152).

* `CrossEntropyLossSemantic(weights, label_smoothing=0.0, weighted_reduction=True)` mirrors the
  constructor and call convention the reference exercises in
  /root/reference/emsanet/tests/test_semantic_loss.py:68-97 (`loss_object(pred_scales,
  target_scales)[i][0]` is the loss of scale i): full-resolution logits and the /8, /16, /32 side
  outputs.  PARITY PINNED by the reference's in-tree oracle class (same file, :15-48).
* `CrossEntropyLossScene(n_classes, label_smoothing=0.1)`: scene head
  (/root/reference/emsanet/task_helper.py:40-47, args.py:790-796) -- the same kernel on 1x1
  "images"; semantics of weights / label smoothing / ignored void label are those of
  torch.nn.CrossEntropyLoss, which the tests use as the reference.
* `InstanceLosses(kappa=1.0)`: MSE centre / L1 offset / von-Mises orientation in one fused pass
  (args.py:739-770).  The loss classes are in the un-vendored nicr_mt_scene_analysis library:
  PARITY UNPINNED, restated in oracle/instance_loss_oracle.py.
* `loss_weights(args)` / `reduce_losses`: /root/reference/emsanet/loss_weighting.py:11-49
  (fixed weights, defaults 1 / 0.25 / 3 x (2 : 1) / 0.5, README.md:621-622).

Arithmetic runs in libemsanet_hip.so (csrc/loss.hip): logits are read once in forward and once in
backward; reductions are deterministic (fixed order, fp64 final sums).
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from . import functional as Fn
from ._lib import check


class SemanticCEFunction(Function):
    @staticmethod
    def forward(ctx, logits, target, weights, label_smoothing, weights_sum):
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
                                     weights.data_ptr(), c, pixels, label_smoothing, weights_sum,
                                     partial.data_ptr(), out.data_ptr(), Fn._stream()),
              'emsa_ce_semantic_fwd')
        ctx.save_for_backward(logits, target, weights, out)
        ctx.smoothing = (label_smoothing, weights_sum)
        return out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        logits, target, weights, out = ctx.saved_tensors
        n, c, h, w = logits.shape
        d = Fn.act_empty(n, Fn.pad4(c), h, w, logits.device)
        g = gout.detach().float().reshape(1).contiguous()
        eps, wsum = ctx.smoothing
        check(_lib.lib().emsa_ce_semantic_bwd(logits.data_ptr(), Fn.ld_of(logits),
                                              target.data_ptr(), weights.data_ptr(), c,
                                              n * h * w, eps, wsum, out.data_ptr(), g.data_ptr(),
                                              d.data_ptr(), Fn.ld_of(d), Fn._stream()),
              'emsa_ce_semantic_bwd')
        return d[:, :c], None, None, None, None


class CrossEntropyLossSemantic(torch.nn.Module):
    def __init__(self, weights, label_smoothing=0.0, weighted_reduction=True):
        super().__init__()
        if not 0.0 <= label_smoothing < 1.0:
            raise ValueError("label_smoothing must be in [0, 1)")
        if not weighted_reduction:
            raise NotImplementedError("only the weighted reduction the reference trains with")
        w = torch.as_tensor(weights, dtype=torch.float32)
        self.register_buffer('weights', w)
        self.label_smoothing = float(label_smoothing)
        self.weights_sum = float(w.double().sum())

    def forward(self, input_scales, target_scales):
        """-> [(loss_scale_i, ), ...] like the reference's loss object (test_semantic_loss.py:93-97)"""
        return [(SemanticCEFunction.apply(x, t, self.weights, self.label_smoothing,
                                          self.weights_sum),)
                for x, t in zip(input_scales, target_scales)]


class CrossEntropyLossScene(torch.nn.Module):
    """scene-classification loss: logits (N, n_classes), target (N,) with 0 = void (ignored)"""

    def __init__(self, n_classes, class_weights=None, label_smoothing=0.1):
        super().__init__()
        w = torch.ones(n_classes) if class_weights is None else \
            torch.as_tensor(class_weights, dtype=torch.float32)
        self.register_buffer('weights', w)
        self.label_smoothing = float(label_smoothing)
        self.weights_sum = float(w.double().sum())

    def forward(self, logits, target):
        n, c = logits.shape
        cp = Fn.pad4(c)
        if cp != c:       # rows must be 16-byte aligned for the kernel: (N, 10) -> row stride 12
            buf = logits.new_zeros(n, cp)
            x = torch.cat([logits, buf[:, c:]], 1)       # tiny (N x 12 floats); autograd slices back
        else:
            x = logits
        x4 = x.view(n, 1, 1, cp).permute(0, 3, 1, 2)[:, :c]    # logical (N,C,1,1), NHWC memory
        return SemanticCEFunction.apply(x4, target.view(n, 1, 1), self.weights,
                                        self.label_smoothing, self.weights_sum)


class InstanceLossFunction(Function):
    @staticmethod
    def forward(ctx, center, offset, orient, center_gt, offset_gt, orient_gt, center_mask, fg,
                fg_orient, kappa):
        center, offset = Fn.as_act(center), Fn.as_act(offset)
        n, _, h, w = center.shape
        pixels = n * h * w
        dev = center.device
        has_o = orient is not None
        if has_o:
            orient = Fn.as_act(orient)

        def u8(m):
            return None if m is None else m.reshape(-1).to(torch.uint8).contiguous()
        cgt = center_gt.reshape(-1).float().contiguous()
        ogt = offset_gt.float().permute(0, 2, 3, 1).contiguous()        # [pixels][2]
        rgt = orient_gt.reshape(-1).float().contiguous() if has_o else None
        cm, fgm, fgo = u8(center_mask), u8(fg), u8(fg_orient) if has_o else None
        if cgt.numel() != pixels or ogt.numel() != 2 * pixels or fgm.numel() != pixels:
            raise _lib.EmsaError("instance loss: target shapes do not match the predictions")
        L = _lib.lib()
        partial = Fn._empty((6 * L.emsa_instance_loss_blocks(pixels),), dev)
        out = Fn._empty((6,), dev)
        p = Fn._p
        check(L.emsa_instance_loss_fwd(p(center), Fn.ld_of(center), p(offset), Fn.ld_of(offset),
                                       p(orient), Fn.ld_of(orient) if has_o else 0, p(cgt),
                                       p(ogt), p(rgt), p(cm), p(fgm), p(fgo), pixels, kappa,
                                       p(partial), p(out), Fn._stream()), 'emsa_instance_loss_fwd')
        ctx.save_for_backward(center, offset, orient, cgt, ogt, rgt, cm, fgm, fgo, out)
        ctx.kappa = kappa
        return out[:3].clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        center, offset, orient, cgt, ogt, rgt, cm, fgm, fgo, out = ctx.saved_tensors
        n, _, h, w = center.shape
        dev = center.device
        has_o = orient is not None
        dc = Fn.act_empty(n, 1, h, w, dev)
        do = Fn.act_empty(n, 2, h, w, dev)
        dr = Fn.act_empty(n, 2, h, w, dev) if has_o else None
        g = g.detach().float().contiguous()
        p = Fn._p
        check(_lib.lib().emsa_instance_loss_bwd(
            p(center), Fn.ld_of(center), p(offset), Fn.ld_of(offset), p(orient),
            Fn.ld_of(orient) if has_o else 0, p(cgt), p(ogt), p(rgt), p(cm), p(fgm), p(fgo),
            n * h * w, ctx.kappa, p(out), p(g), p(dc), 1, p(do), 2, p(dr), 2, Fn._stream()),
            'emsa_instance_loss_bwd')
        return dc, do, dr, None, None, None, None, None, None, None


class InstanceLosses(torch.nn.Module):
    """-> dict(instance_center=..., instance_offset=..., instance_orientation=...) of one scale"""

    def __init__(self, kappa=1.0, center_loss='mse'):
        super().__init__()
        if center_loss != 'mse':
            raise NotImplementedError("instance centre loss: only the reference default 'mse' "
                                      "(emsanet/args.py:749-755)")
        self.kappa = float(kappa)

    def forward(self, center, offset, orientation, center_gt, offset_gt, foreground,
                orientation_gt=None, orientation_foreground=None, center_mask=None):
        out = InstanceLossFunction.apply(center, offset, orientation, center_gt, offset_gt,
                                         orientation_gt, center_mask, foreground,
                                         orientation_foreground, self.kappa)
        res = {'instance_center': out[0], 'instance_offset': out[1]}
        if orientation is not None:
            res['instance_orientation'] = out[2]
        return res


def loss_weights(args):
    """flat per-loss weights from the task weights, /root/reference/emsanet/loss_weighting.py:15-49:
    orientation keeps its own task weight, the instance weight is multiplied into the
    (centre, offset) pair of `--instance-weighting`, every other task keeps its weight."""
    tasks, weighting = list(args.tasks), list(args.tasks_weighting)
    if len(tasks) != len(weighting):
        raise ValueError("tasks and tasks_weighting differ in length")
    tw = dict(zip(tasks, weighting))
    out = {}
    if 'orientation' in tw:
        out['instance_orientation'] = tw.pop('orientation')
    if 'instance' in tw:
        wi = tw.pop('instance')
        wc, wo = args.instance_weighting
        out['instance_center'] = wi * wc
        out['instance_offset'] = wi * wo
    out.update(tw)
    return out


def reduce_losses(losses, weights):
    """total = sum_k weights[k] * losses[k] (FixedLossWeighting, loss_weighting.py:49); `losses`
    may hold lists (multi-scale supervision): their entries are summed first"""
    total = None
    for k, wgt in weights.items():
        if k not in losses:
            continue
        v = losses[k]
        if isinstance(v, (list, tuple)):
            v = torch.stack([x[0] if isinstance(x, (list, tuple)) else x for x in v]).sum()
        total = wgt * v if total is None else total + wgt * v
    return total


class TrainingLosses(torch.nn.Module):
    """All task losses of one training step on the engine's raw outputs (the list returned by
    `EMSANet.forward(batch)` in training mode), weighted like the reference
    (`RunHelper.training_step`, /root/reference/main.py:129-152): every task helper's losses at
    full resolution and -- multi-scale supervision, on by default -- on the side outputs, reduced
    with the fixed weights of `loss_weights(args)`.

    targets: dict with, per scale (index 0 = full resolution, then the side outputs in the
    order the decoder returns them: /32, /16, /8),
        'semantic'  [ (N,H_s,W_s) int, 0 = void ]
        'instance'  [ dict(center (N,1,H_s,W_s), offset (N,2,H_s,W_s), foreground (N,H_s,W_s),
                           orientation (N,H_s,W_s) rad, orientation_foreground (N,H_s,W_s)) ]
        'scene'     (N,) int, 0 = void
    (how the reference's batch dictionary names the down-scaled targets is defined by the
    un-vendored library's preprocessing: [U], so the caller maps them.)"""

    def __init__(self, args, semantic_class_weights, n_scene_classes):
        super().__init__()
        self.tasks = tuple(args.tasks)
        self.panoptic = bool(getattr(args, 'enable_panoptic', False))
        self.weights = loss_weights(args)
        self.sem_multiscale = not args.semantic_no_multiscale_supervision
        self.inst_multiscale = not args.instance_no_multiscale_supervision
        self.semantic = CrossEntropyLossSemantic(semantic_class_weights,
                                                 args.semantic_loss_label_smoothing)
        self.scene = CrossEntropyLossScene(n_scene_classes,
                                           label_smoothing=args.scene_loss_label_smoothing)
        self.instance = InstanceLosses(args.orientation_kappa, args.instance_center_loss)

    def forward(self, outputs, targets):
        """-> (total, dict of unweighted per-task losses summed over the scales)"""
        losses = {}
        panoptic = self.panoptic
        if isinstance(outputs, dict):
            # the merged dictionary of `model(batch, do_postprocessing=True)` in TRAIN mode -- what the
            # reference's training step hands to its task helpers (`predictions_post`,
            # /root/reference/main.py:126-141): the same tensors under '<task>_output' /
            # '<task>_side_outputs'
            merged, outputs, panoptic = outputs, [], False
            if 'semantic' in self.tasks:
                outputs.append((merged['semantic_output'], merged['semantic_side_outputs']))
            if 'instance' in self.tasks or 'orientation' in self.tasks:
                outputs.append((merged['instance_output'], merged['instance_side_outputs']))
            if 'scene' in self.tasks:
                outputs.append((merged['scene_output'],))
        outputs = list(outputs)
        if panoptic:
            # PanopticHelper: ((semantic, instance), (semantic sides, instance sides)) first
            (sem, inst), (sem_side, inst_side) = outputs[0]
            outputs = [(sem, sem_side), (inst, inst_side)] + outputs[1:]
        it = iter(outputs)
        if 'semantic' in self.tasks:
            full, side = next(it)
            xs = [full] + (list(side) if self.sem_multiscale else [])
            ls = self.semantic(xs, targets['semantic'][:len(xs)])
            losses['semantic'] = torch.stack([l[0] for l in ls]).sum()
        if 'instance' in self.tasks or 'orientation' in self.tasks:
            full, side = next(it)
            scales = [full] + (list(side) if self.inst_multiscale else [])
            acc = {}
            for pred, tgt in zip(scales, targets['instance']):
                center, offset = pred[0], pred[1]
                orient = pred[2] if len(pred) > 2 else None
                r = self.instance(center, offset, orient, tgt['center'], tgt['offset'],
                                  tgt['foreground'], tgt.get('orientation'),
                                  tgt.get('orientation_foreground'), tgt.get('center_mask'))
                for k, v in r.items():
                    acc[k] = v if k not in acc else acc[k] + v
            losses.update(acc)
        if 'scene' in self.tasks:
            logits = next(it)[0]
            losses['scene'] = self.scene(logits, targets['scene'])
        return reduce_losses(losses, self.weights), losses
