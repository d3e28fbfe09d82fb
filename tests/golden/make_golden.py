"""Generates the golden fixtures in this directory from the oracle (run in the build container:
`python tests/golden/make_golden.py`).  The reference holds NO golden vectors for this path
(its tests assert shapes/types only and its arithmetic lives in an un-vendored
library), so these fixtures pin the build's own oracle -- "parity unpinned" w.r.t. upstream.

  config1_rgb_semantic_160x128.npz   BASELINE config 1: argmax map (uint8), strided logits sample,
                                     per-output fp64 checksums
  full_rgbd_96x64_eval.npz           full multi-task model, eval: strided samples of every raw
                                     output + fp64 checksums
  full_rgbd_96x64_train.npz          same in train mode (BN batch stats, hash Dropout2d seed 1234):
                                     outputs, side outputs, checksums, a few parameter gradients
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from emsanet_amd import default_args, full_args, nyuv2_config      # noqa: E402
from oracle.emsanet_oracle import (EMSANetOracle, deterministic_state_dict,   # noqa: E402
                                   synthetic_batch)


def flatten(outs):
    flat = []
    for o, sides in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            flat += list(s) if isinstance(s, tuple) else [s]
    return flat


def sample(t, step=4):
    t = t.detach()
    if t.dim() == 4:
        return t[:, :, ::step, ::step].numpy().copy()
    return t.numpy().copy()


def checks(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def main():
    torch.manual_seed(0)
    cfg = nyuv2_config()
    # ---- config 1 --------------------------------------------------------------------------
    a1 = default_args(input_modalities=('rgb',), tasks=('semantic',), input_height=128,
                      input_width=160, no_pretrained_backbone=True)
    o = EMSANetOracle(a1, cfg)
    o.load_state_dict(deterministic_state_dict(o, 0))
    o.eval()
    with torch.no_grad():
        out = o(synthetic_batch(2, 128, 160, modalities=('rgb',)))
    logits = out[0][0]
    np.savez_compressed(os.path.join(HERE, 'config1_rgb_semantic_160x128.npz'),
                        argmax=logits.argmax(1).numpy().astype(np.uint8),
                        logits_sample=sample(logits), checksum=checks(logits))
    # ---- full model, eval ------------------------------------------------------------------
    a2 = full_args(input_height=64, input_width=96)
    o = EMSANetOracle(a2, cfg)
    o.load_state_dict(deterministic_state_dict(o, 0))
    o.eval()
    batch = synthetic_batch(2, 64, 96)
    with torch.no_grad():
        flat = flatten(o(batch))
    d = {f'out{i}': sample(t, 2) for i, t in enumerate(flat)}
    d.update({f'sum{i}': checks(t) for i, t in enumerate(flat)})
    d['semantic_argmax'] = flat[0].argmax(1).numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, 'full_rgbd_96x64_eval.npz'), **d)
    # ---- full model, train -----------------------------------------------------------------
    o.train()
    o.dropout_seed, o.dropout_step = 1234, 0
    flat = flatten(o(batch))
    loss = sum((t * t).mean() for t in flat)
    loss.backward()
    d = {f'out{i}': sample(t, 2) for i, t in enumerate(flat)}
    d.update({f'sum{i}': checks(t) for i, t in enumerate(flat)})
    for k in ('encoder.backbone_rgb.layer1.0.conv3x1_1.weight',
              'encoder.backbone_depth.layer2.0.downsample.0.weight',
              'encoder.fusion_modules.1.se_rgb.fc.0.weight',
              'context_module.final_conv.norm.weight',
              'decoders.semantic_decoder.head.conv.weight',
              'decoders.instance_decoder.head.task_convs.2.weight',
              'decoders.scene_decoder.head.weight'):
        d['grad:' + k] = dict(o.named_parameters())[k].grad.numpy().copy()
    d['running_mean:encoder.backbone_rgb.bn1'] = o.encoder.backbone_rgb.bn1.running_mean.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'full_rgbd_96x64_train.npz'), **d)
    for f in sorted(os.listdir(HERE)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
