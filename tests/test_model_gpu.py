"""GPU engine vs the oracle (tests infrastructure) on identical inputs and weights:
forward outputs within 1e-3 (BASELINE.json north_star), argmax-exact class maps, and every
parameter gradient of a multi-task backward pass."""
import pytest
import torch

from util import DEV, close, rnd, to_act

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _pair(args, seed=0):
    from emsanet_amd import nyuv2_config
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import EMSANetOracle, deterministic_state_dict
    cfg = nyuv2_config()
    oracle = EMSANetOracle(args, cfg)
    sd = deterministic_state_dict(oracle, seed)
    oracle.load_state_dict(sd)
    model = EMSANet(args, cfg)
    model.load_state_dict(sd)
    return model.to(DEV), oracle


def _flatten(outs):
    flat = []
    for o, sides in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            flat += list(s) if isinstance(s, tuple) else [s]
    return flat


@pytest.mark.parametrize('cin,cout,stride,p', [(64, 64, 1, 0.0), (64, 64, 1, 0.2),
                                              (64, 128, 2, 0.1), (128, 128, 1, 0.5)])
@pytest.mark.parametrize('mode', ['train', 'eval_grad', 'eval_fast'])
def test_nbt1d_block(cin, cout, stride, p, mode):
    from emsanet_amd.nn import NonBottleneck1D
    from oracle import emsanet_oracle as O
    torch.manual_seed(0)
    ref = O.NonBottleneck1D(cin, cout, stride, p)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    ref.dropout.layer_id = 3
    ref.dropout.seed_fn = lambda: 42
    blk = NonBottleneck1D(cin, cout, stride, p)
    blk.load_state_dict(ref.state_dict())
    blk.dropout.layer_id = 3
    blk.dropout.seed_fn = lambda: 42
    blk.to(DEV)
    x = rnd(2, cin, 12, 16, seed=1)
    if mode == 'train':
        ref.train(), blk.train()
    else:
        ref.eval(), blk.eval()
    if mode == 'eval_fast':
        with torch.no_grad():
            close(blk(to_act(x)), ref(x), tol=2e-4, what='block eval')
        return
    xr = x.clone().requires_grad_(True)
    xg = to_act(x).requires_grad_(True)
    yr, yg = ref(xr), blk(xg)
    close(yg, yr, tol=2e-4, what='block fwd')
    dy = rnd(*yr.shape, seed=2)
    yr.backward(dy)
    yg.backward(to_act(dy))
    close(xg.grad, xr.grad, tol=5e-4, what='block dx')
    rp = dict(ref.named_parameters())
    for k, pg in blk.named_parameters():
        close(pg.grad, rp[k].grad, tol=5e-4, what=f'block grad {k}')
    if mode == 'train':
        rb = dict(ref.named_buffers())
        for k, b in blk.named_buffers():
            if 'running' in k:
                close(b, rb[k], tol=1e-4, what=f'block buffer {k}')


def test_config1_rgb_semantic_eval():
    """BASELINE config 1: R34-NBt1D RGB-only, semantic head only, 160x128, bs=2"""
    from emsanet_amd import default_args
    from oracle.emsanet_oracle import synthetic_batch
    args = default_args(input_modalities=('rgb',), tasks=('semantic',), input_height=128,
                        input_width=160, no_pretrained_backbone=True)
    model, oracle = _pair(args)
    model.eval(), oracle.eval()
    batch = synthetic_batch(2, 128, 160, modalities=('rgb',))
    with torch.no_grad():
        ref = oracle(batch)
        out = model({k: v.to(DEV) for k, v in batch.items()})
    close(out[0][0], ref[0][0], tol=TOL, what='semantic logits')
    assert torch.equal(out[0][0].argmax(1).cpu(), ref[0][0].argmax(1)), "argmax map differs"


@pytest.mark.parametrize('mode', ['eval_fast', 'eval_grad', 'train'])
def test_full_model_small(mode):
    """full RGB-D multi-task model at 128x160, bs=2: outputs, side outputs, all gradients"""
    from emsanet_amd import full_args
    from oracle.emsanet_oracle import synthetic_batch
    args = full_args(input_height=128, input_width=160)
    model, oracle = _pair(args)
    batch = synthetic_batch(2, 128, 160)
    gbatch = {k: v.to(DEV) for k, v in batch.items()}
    if mode == 'train':
        model.train(), oracle.train()
        model.dropout_seed = oracle.dropout_seed = 1234
    else:
        model.eval(), oracle.eval()
    if mode == 'eval_fast':
        with torch.no_grad():
            ref, out = oracle(batch), model(gbatch)
    else:
        ref, out = oracle(batch), model(gbatch)
    fr, fo = _flatten(ref), _flatten(out)
    assert len(fr) == len(fo)
    for i, (a, b) in enumerate(zip(fo, fr)):
        close(a, b, tol=TOL, what=f'{mode} output {i}')
    assert torch.equal(fo[0].argmax(1).cpu(), fr[0].argmax(1)), "semantic argmax differs"
    if mode == 'eval_fast':
        return
    cots = [rnd(*t.shape, seed=100 + i, scale=1e-1) for i, t in enumerate(fr)]
    torch.autograd.backward(fr, cots)
    torch.autograd.backward(fo, [c.to(DEV) for c in cots])
    rp = dict(oracle.named_parameters())
    worst = ('', 0.0)
    for k, p in model.named_parameters():
        if k.endswith('conv1.weight') and 'backbone' in k and 'layer' not in k:
            pass
        assert p.grad is not None, f"no grad for {k}"
        g, r = p.grad.detach().cpu().double(), rp[k].grad.double()
        err = (g - r).abs().max().item() / max(1e-6, r.abs().max().item())
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < 5e-3, f"{mode}: worst relative gradient error {worst}"
    if mode == 'train':
        rb = dict(oracle.named_buffers())
        for k, b in model.named_buffers():
            if 'running' in k:
                close(b, rb[k], tol=1e-3, what=f'buffer {k}')


def test_full_res_eval_bs1():
    """BASELINE config 2 shape (640x480 RGB-D, all heads), bs=1, eval: 1e-3 / argmax-exact"""
    from emsanet_amd import full_args
    from oracle.emsanet_oracle import synthetic_batch
    args = full_args()
    model, oracle = _pair(args)
    model.eval(), oracle.eval()
    batch = synthetic_batch(1, 480, 640)
    with torch.no_grad():
        ref = oracle(batch)
        out = model({k: v.to(DEV) for k, v in batch.items()})
    fr, fo = _flatten(ref), _flatten(out)
    for i, (a, b) in enumerate(zip(fo, fr)):
        close(a, b, tol=TOL, what=f'output {i}')
    assert torch.equal(fo[0].argmax(1).cpu(), fr[0].argmax(1)), "semantic argmax differs"
    assert torch.equal(fo[-1].argmax(1).cpu(), fr[-1].argmax(1)), "scene argmax differs"


def test_missing_gpu_input_fails_loudly():
    from emsanet_amd import _lib, full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import synthetic_batch
    model = EMSANet(full_args(input_height=64, input_width=64), nyuv2_config()).to(DEV)
    with pytest.raises(_lib.EmsaError):
        model(synthetic_batch(1, 64, 64))      # CPU tensors: no CPU fallback
