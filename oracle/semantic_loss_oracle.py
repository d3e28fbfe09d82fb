# -*- coding: utf-8 -*-
"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY PINNED (for this piece).

CPU restatement of the class-weighted semantic cross-entropy of the reference's training step
(`task_helper.training_step`, /root/reference/main.py:131-141; defaults
`--semantic-loss-label-smoothing 0.0`, /root/reference/emsanet/args.py:724-729).  The numerics
are pinned by the reference's own in-tree known-answer class `CrossEntropyLossPrevious`
(/root/reference/emsanet/tests/test_semantic_loss.py:15-48), against which the reference asserts
its current `CrossEntropyLossSemantic(weights, label_smoothing=0.0, weighted_reduction=True)`
(:51-103).  `tests/golden/semantic_ce.npz` holds outputs of THAT class, produced by executing it
from the reference file (tests/golden/make_golden.py); this restatement and the HIP kernels are
checked against them.

Semantics (test_semantic_loss.py:31-48): targets are class indices with 0 = void; per pixel
l = w[t-1] * -log softmax(x)[t-1] (void pixels contribute nothing); the loss of one scale is
sum(l) / sum over non-void pixels of w[t-1].
"""
import torch
import torch.nn.functional as F

# class weights used by the reference's test (test_semantic_loss.py:56-66) -- fixture DATA
NYUV2_TEST_CLASS_WEIGHTS = (
    0.2650426, 0.5533999, 0.42025763, 0.34482047, 0.7993162, 0.49264285, 1.1026958, 0.78996897,
    0.76780474, 0.36996013, 1.6053797, 0.97266424, 0.63303965, 0.73651886, 0.92407864,
    0.59753835, 0.4705898, 1.7916499, 0.61840767, 1.1446692, 1.1642636, 1.081512, 1.8748288,
    0.6763455, 1.0289167, 4.0649543, 1.5289997, 0.42058772, 3.60466, 0.53412074, 1.246997,
    2.2661245, 0.9652696, 3.0297952, 5.316681, 1.0555762, 6.7779245, 1.0640355, 1.2999853,
    1.1953188)


def semantic_ce(logits, target, weights):
    """logits (N,C,H,W) float, target (N,H,W) integer with 0 = void, weights (C,) -> scalar"""
    w = torch.as_tensor(weights, dtype=logits.dtype, device=logits.device)
    t = target.long() - 1
    per_pixel = F.cross_entropy(logits, t, weight=w, reduction='none', ignore_index=-1)
    valid = t >= 0
    divisor = w[t.clamp(min=0)][valid].sum()
    return per_pixel.sum() / divisor


def semantic_ce_multiscale(logits_scales, target_scales, weights):
    return [semantic_ce(x, t, weights) for x, t in zip(logits_scales, target_scales)]
