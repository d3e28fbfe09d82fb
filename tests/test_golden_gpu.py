"""GPU: the HIP engine against the committed golden fixtures (tests/golden/*.npz), i.e. parity on
the GPU box without importing the oracle at all."""
import os

import numpy as np
import pytest
import torch

from util import DEV

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _flatten(outs):
    flat = []
    for o, sides in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            flat += list(s) if isinstance(s, tuple) else [s]
    return flat


def _inputs(bs, h, w, modalities=('rgb', 'depth')):
    # the reference's synthetic generator (inference_time_whole_model.py:519-545), seed 1234
    rng = np.random.default_rng(1234)
    rgb = rng.integers(0, 255, (bs, h, w, 3), dtype=np.uint8)
    depth = rng.integers(0, 40000, (bs, h, w), dtype=np.uint16)
    b = {}
    if 'rgb' in modalities:
        b['rgb'] = torch.from_numpy((rgb.astype(np.float32) / 255).transpose(0, 3, 1, 2).copy()).to(DEV)
    if 'depth' in modalities:
        b['depth'] = torch.from_numpy((depth.astype(np.float32) / 20000)[:, None].copy()).to(DEV)
    return b


def _weights(model, seed=0):
    # same generator as oracle.deterministic_state_dict, restated in tests/util.py so that this
    # test does not touch oracle/ (state-dict order and shapes are identical by construction)
    from util import deterministic_state_dict
    model.load_state_dict(deterministic_state_dict(model, seed))
    return model.to(DEV)


def _close(a, b, tol):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    assert a.shape == b.shape
    err = np.abs(a - b).max()
    assert err <= tol * max(1.0, np.abs(b).max()), f"max abs err {err:.3e}"


def test_config1_golden_gpu():
    from emsanet_amd import default_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    g = np.load(os.path.join(GOLD, 'config1_rgb_semantic_160x128.npz'))
    args = default_args(input_modalities=('rgb',), tasks=('semantic',), input_height=128,
                        input_width=160, no_pretrained_backbone=True)
    model = _weights(EMSANet(args, nyuv2_config())).eval()
    with torch.no_grad():
        logits = model(_inputs(2, 128, 160, ('rgb',)))[0][0]
    _close(logits[:, :, ::4, ::4].cpu().numpy(), g['logits_sample'], 1e-3)
    am = logits.argmax(1).cpu().numpy().astype(np.uint8)
    assert (am != g['argmax']).mean() < 1e-4
    _close(float(logits.double().abs().sum()), g['checksum'][1], 1e-4)


@pytest.mark.parametrize('mode', ['eval', 'train'])
def test_full_model_golden_gpu(mode):
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    g = np.load(os.path.join(GOLD, f'full_rgbd_96x64_{mode}.npz'))
    model = _weights(EMSANet(full_args(input_height=64, input_width=96), nyuv2_config()))
    batch = _inputs(2, 64, 96)
    if mode == 'eval':
        model.eval()
        with torch.no_grad():
            flat = _flatten(model(batch))
    else:
        model.train()
        model.dropout_seed, model.dropout_step = 1234, 0
        flat = _flatten(model(batch))
        sum((t * t).mean() for t in flat).backward()
    tol = 1e-3 if mode == 'eval' else 1e-2   # train: BN over 2-sample batches (ill-conditioned)
    for i, t in enumerate(flat):
        s = t.detach()[:, :, ::2, ::2] if t.dim() == 4 else t.detach()
        _close(s.cpu().numpy(), g[f'out{i}'], tol)
    if mode == 'eval':
        am = flat[0].argmax(1).cpu().numpy().astype(np.uint8)
        assert (am != g['semantic_argmax']).mean() < 1e-4
    else:
        params = dict(model.named_parameters())
        for k in g.files:
            if k.startswith('grad:'):
                _close(params[k[5:]].grad.cpu().numpy(), g[k], 5e-2)
        _close(model.encoder.backbone_rgb.bn1.running_mean.cpu().numpy(),
               g['running_mean:encoder.backbone_rgb.bn1'], 1e-4)
