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
