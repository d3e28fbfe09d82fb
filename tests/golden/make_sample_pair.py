"""Generates tests/golden/sample_pair.npz (run in the build container:
`python tests/golden/make_sample_pair.py`).

The reference ships ONE real RGB-D frame pair for its inference script, `samples/sample_rgb.png`
(1440x1080 uint8) and `samples/sample_depth.png` (1440x1080 uint16, millimetres, 5.4 % invalid
zeros) -- /root/reference/inference_samples.py:104-136 loads them, scales the depth, and runs the
model on the preprocessed pair (`preprocessing.py:216-226`: NormalizeRGB, NormalizeDepth with the
dataset's depth statistics and raw-depth zero handling, ToTorchTensors).  Every other parity test of
this repo feeds uniform noise; real depth has invalid zeros and long flat regions, real RGB a very
different ReLU sparsity.  The pair is a reference-held INPUT VECTOR: the fixture stores the decoded
frames resized to the network input (480x640; PIL bilinear for RGB, nearest for depth -- the
reference's `Resize` lives in the absent nicr_mt_scene_analysis, SURVEY 0.1, so the resampling filter
is this script's choice and part of the fixture, not a parity claim) as uint8 / uint16 arrays and NO
text of any reference file.

Expected outputs come from the build's oracle (oracle/emsanet_oracle.py, parity unpinned w.r.t.
upstream): deterministic weights (seed 0), BatchNorm running statistics recalibrated on
this frame (one train-mode pass with momentum 1 and Dropout2d seed 2024 over a batch of SIX: the
frame and five augmented twins -- mirrored / flipped, colour channels permuted, depth scaled --
because the PPM's 1x1 bin needs several distinct values per channel for a usable variance; frozen
statistics that do not belong to the weights leave the eval forward un-normalised), then the eval
forward of the frame alone.
Stored: semantic arg-max map (uint8), logits sampled with stride 8, centre / offset / orientation with 4,
scene logits, fp64 checksums of every raw output.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

H, W = 480, 640
DEPTH_STATS = (2841.94, 1417.26)          # NYUv2 train split, refined depth (emsanet_amd.data)
RGB_MEAN = (0.485, 0.456, 0.406)
RGB_STD = (0.229, 0.224, 0.225)
STRIDE = dict(semantic=8, center=4, offset=4, orientation=4, scene=1)     # sampling of the stored maps


def normalise(rgb_u8, depth_u16):
    """numpy restatement of NormalizeRGB / NormalizeDepth(raw_depth=False semantics of the engine's
    staging kernels: invalid zeros stay zero) + HWC -> CHW, fp32"""
    rgb = rgb_u8.astype(np.float32) / np.float32(255.0)
    rgb = (rgb - np.array(RGB_MEAN, np.float32)) / np.array(RGB_STD, np.float32)
    d = depth_u16.astype(np.float32)
    dn = (d - np.float32(DEPTH_STATS[0])) / np.float32(DEPTH_STATS[1])
    dn[depth_u16 == 0] = 0.0
    return (torch.from_numpy(rgb.transpose(2, 0, 1)[None].copy()),
            torch.from_numpy(dn[None, None].copy()))


def calibration_batch(rgb_u8, depth_u16):
    """the frame and five augmented twins (mirrored / flipped, colour channels permuted, depth scaled):
    six samples per channel for the 1x1 bin of the pyramid pooling, whose BatchNorm otherwise sees a
    near-zero variance and amplifies every rounding of its branch by 1/sqrt(eps)"""
    variants = [
        (rgb_u8, depth_u16),
        (rgb_u8[:, ::-1, ::-1], depth_u16[:, ::-1] // 2),
        (rgb_u8[::-1, :, [1, 2, 0]], (depth_u16[::-1].astype(np.uint32) * 3 // 4).astype(np.uint16)),
        (rgb_u8[::-1, ::-1, [2, 0, 1]], np.minimum(depth_u16[::-1, ::-1].astype(np.uint32) * 5 // 4, 65535).astype(np.uint16)),
        (255 - rgb_u8, depth_u16 // 3),
        (np.roll(rgb_u8, 213, axis=1)[:, :, ::-1], np.roll(depth_u16, 213, axis=1)),
    ]
    parts = [normalise(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in variants]
    return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])


def recalibrated_oracle(rgb, depth, dtype=torch.float32, **arg_overrides):
    """rgb / depth: the normalised calibration batch (6, C, H, W)"""
    from emsanet_amd import full_args, nyuv2_config
    from oracle.emsanet_oracle import EMSANetOracle, deterministic_state_dict
    oracle = EMSANetOracle(full_args(input_height=H, input_width=W, **arg_overrides), nyuv2_config())
    oracle.load_state_dict(deterministic_state_dict(oracle, 0))
    bns = [m for m in oracle.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    moms = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    oracle.train()
    oracle.dropout_seed = 2024
    with torch.no_grad():
        oracle({'rgb': rgb, 'depth': depth})
    for m, mom in zip(bns, moms):
        m.momentum = mom
    oracle.eval()
    return oracle.to(dtype)


def flat_eval(outs):
    flat = []
    for o, _ in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
    return flat


def main():
    from PIL import Image
    src = '/root/reference/samples'
    rgb = Image.open(os.path.join(src, 'sample_rgb.png')).convert('RGB').resize((W, H), Image.BILINEAR)
    dep = Image.open(os.path.join(src, 'sample_depth.png')).resize((W, H), Image.NEAREST)
    rgb_u8 = np.asarray(rgb, dtype=np.uint8)
    depth_u16 = np.asarray(dep).astype(np.uint16)
    assert rgb_u8.shape == (H, W, 3) and depth_u16.shape == (H, W)
    x_rgb, x_depth = normalise(rgb_u8, depth_u16)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    oracle = recalibrated_oracle(*calibration_batch(rgb_u8, depth_u16), torch.float64)
    with torch.no_grad():
        out = flat_eval(oracle({'rgb': x_rgb.double(), 'depth': x_depth.double()}))
    names = ['semantic', 'center', 'offset', 'orientation', 'scene']
    assert len(out) == len(names), len(out)
    data = dict(rgb_u8=rgb_u8, depth_u16=depth_u16,
                semantic_argmax=out[0].argmax(1)[0].numpy().astype(np.uint8))
    for n, t in zip(names, out):
        t = t.detach()
        st = STRIDE[n]
        data[n + '_sample'] = (t[:, :, ::st, ::st] if t.dim() == 4 else t).float().numpy().copy()
        data[n + '_checks'] = np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])
    np.savez_compressed(os.path.join(HERE, 'sample_pair.npz'), **data)
    print({k: (v.shape, str(v.dtype)) for k, v in data.items()})
    print('invalid depth pixels: %.2f %%' % (100.0 * (depth_u16 == 0).mean()),
          'classes in the arg-max map:', len(np.unique(data['semantic_argmax'])),
          'centre max %.3f' % out[1].max().item())


if __name__ == '__main__':
    main()
