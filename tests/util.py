"""helpers shared by the GPU parity tests (reference = plain PyTorch on the CPU, fp64/fp32)."""
import numpy as np
import torch

DEV = 'cuda:0'


def rnd(*shape, seed=0, scale=1.0):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def to_act(t_nchw):
    """CPU NCHW tensor -> GPU activation (logical NCHW, NHWC memory)."""
    return t_nchw.to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def close(got, ref, tol=1e-4, what=''):
    """max |got-ref| <= tol * max(1, max|ref|)"""
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    err = (got - ref).abs().max().item()
    lim = tol * max(1.0, ref.abs().max().item())
    assert err <= lim, f"{what}: max abs err {err:.3e} > {lim:.3e}"
    return err


def deterministic_state_dict(model, seed=0):
    """Same generator as oracle.emsanet_oracle.deterministic_state_dict (restated so that
    tests/test_golden_gpu.py runs without importing oracle/): every parameter/buffer from numpy
    default_rng(seed) in state_dict order."""
    from collections import OrderedDict
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for k, v in model.state_dict().items():
        shp = tuple(v.shape)
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith('running_mean'):
            sd[k] = torch.from_numpy(rng.normal(0, 0.1, shp).astype(np.float32))
        elif k.endswith('running_var'):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif v.dim() == 1 and k.endswith('bn2.weight'):
            sd[k] = torch.from_numpy(rng.uniform(0.2, 0.4, shp).astype(np.float32))
        elif v.dim() == 1 and (('bn' in k or 'norm' in k or k.split('.')[-2] == '1')
                               and k.endswith('weight')):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif v.dim() == 1:
            sd[k] = torch.from_numpy(rng.normal(0, 0.05, shp).astype(np.float32))
        else:
            fan_in = int(np.prod(shp[1:]))
            std = np.sqrt(2.0 / fan_in)
            if 'upsampling' in k:
                base = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], np.float32) / 16
                sd[k] = torch.from_numpy((base + rng.normal(0, 0.02, shp)).astype(np.float32))
            else:
                sd[k] = torch.from_numpy(rng.normal(0, std, shp).astype(np.float32))
    return sd
