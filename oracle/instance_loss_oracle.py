# -*- coding: utf-8 -*-
"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

CPU restatement (plain PyTorch) of the instance-decoder losses of the reference's training step
(`InstanceTaskHelper.training_step`, wired in /root/reference/emsanet/task_helper.py:56-63 with
`loss_name_instance_center=args.instance_center_loss` = 'mse', /root/reference/emsanet/args.py:
749-755; `--orientation-kappa 1.0`, args.py:763-770) and of the scene / label-smoothed
cross-entropy (task_helper.py:40-47, args.py:790-796).

The loss classes themselves live in the third-party dependency `nicr_mt_scene_analysis` (git
submodule pinned at v0.3.1 by the reference, NOT vendored under /root/reference), so this file
restates the published definitions (EMSANet, Seichter et al., IJCNN 2022, Sec. III; the
Panoptic-DeepLab instance encoding it adopts; biternion / von-Mises orientation loss of Beyer et
al.) and keeps every choice the in-tree sources do not determine a named constant of `Spec`:

* centre: mean squared error between the (sigmoid) heatmap and the Gaussian target heatmap over
  the pixels of `center_mask` (all pixels when no mask is given);
* offset: L1 between predicted and target (dy, dx) over the instance foreground, averaged over
  the foreground ELEMENTS (2 per pixel)                                  [Spec.OFFSET_DIVISOR]
* orientation: 1 - exp(kappa * (cos(theta - theta*) - 1)) averaged over the foreground pixels
  that carry an orientation label; theta from the L2-normalised (sin, cos) prediction pair
                                                                          [Spec.ORIENT_ORDER]
* scene / smoothed CE: torch.nn.functional.cross_entropy(weight, label_smoothing,
  ignore_index) semantics with targets shifted by one (0 = void), divisor = sum of the target
  class weights of the non-void samples (identical to the pinned semantic oracle for eps = 0).
"""
import torch
import torch.nn.functional as F


class Spec:
    OFFSET_DIVISOR = 2      # [U] L1 offsets averaged over 2*n_fg elements (1: over n_fg pixels)
    ORIENT_ORDER = ('sin', 'cos')    # [U] channel order of the orientation prediction
    NORM_EPS = 1e-12        # clamp of |v|^2 before normalising the orientation vector


def smoothed_ce(logits, target, weights, label_smoothing):
    """logits (N,C,...) float, target (N,...) integer with 0 = void -> scalar"""
    w = torch.as_tensor(weights, dtype=logits.dtype)
    t = target.long() - 1
    per = F.cross_entropy(logits, t, weight=w, reduction='none', ignore_index=-1,
                          label_smoothing=label_smoothing)
    valid = t >= 0
    return per.sum() / w[t.clamp(min=0)][valid].sum()


def instance_losses(center, offset, orientation, center_gt, offset_gt, foreground,
                    orientation_gt=None, orientation_foreground=None, center_mask=None,
                    kappa=1.0):
    """center (N,1,H,W), offset (N,2,H,W), orientation (N,2,H,W) or None; targets alike
    (orientation_gt (N,H,W) angle in rad); masks (N,H,W) bool -> dict of scalar losses"""
    out = {}
    cm = torch.ones_like(center_gt.reshape(center.shape), dtype=torch.bool) if center_mask is None \
        else center_mask.reshape(center.shape).bool()
    d = (center - center_gt.reshape(center.shape)) ** 2
    out['instance_center'] = (d * cm).sum() / cm.sum().clamp(min=1)
    fg = foreground.bool().unsqueeze(1)
    l1 = (offset - offset_gt).abs() * fg
    n_el = fg.sum() * 2
    out['instance_offset'] = l1.sum() / (n_el if Spec.OFFSET_DIVISOR == 2 else fg.sum()).clamp(min=1)
    if orientation is not None:
        fgo = orientation_foreground.bool()
        v = orientation / orientation.pow(2).sum(1, keepdim=True).clamp(min=Spec.NORM_EPS).sqrt()
        s, c = v[:, 0], v[:, 1]
        cosd = s * torch.sin(orientation_gt) + c * torch.cos(orientation_gt)
        l = (1.0 - torch.exp(kappa * (cosd - 1.0))) * fgo
        out['instance_orientation'] = l.sum() / fgo.sum().clamp(min=1)
    return out
