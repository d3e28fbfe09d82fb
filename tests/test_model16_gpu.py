"""16-bit engine (BASELINE configs[2]: bf16 mixed-precision training; configs[4]: 16-bit whole-model
hipGraph inference) against the fp32 / fp64 CPU oracle on identical inputs and weights.

Stated tolerances (activations are STORED with an 8-bit (bf16) / 11-bit (fp16) mantissa between
~100 layers; all accumulation, BatchNorm statistics, parameters and outputs are fp32):
  * eval outputs: relative L2 error of every raw output vs the fp32 oracle  <= 3e-2 (bf16),
    <= 4e-3 (fp16); semantic / scene class maps: arg-max identical except where the oracle's top-2
    margin is below 4 x the measured max error (no disagreement at a non-tie), and on >= 97.5 %
    (bf16) / 99.7 % (fp16) of the pixels of a RANDOM-WEIGHT network, whose class margins are tiny
    (measured 98.3 % / 99.8 %);
  * train step (bf16): outputs as above against the fp64 oracle replaying the engine's ReLU
    decisions; every parameter gradient within 8e-2 relative L2 (median <= 2e-2) of fp64.
The fp32 engine keeps north_star's 1e-3 (tests/test_model_gpu.py).
"""
import pytest
import torch

from util import DEV, rnd

pytestmark = pytest.mark.gpu

OUT_TOL = {torch.bfloat16: 3e-2, torch.float16: 4e-3}
AGREE = {torch.bfloat16: 0.975, torch.float16: 0.997}
# against the storage-emulating oracle (rounding points reproduced): provisional, see the measured values
EMU_TOL = {torch.bfloat16: 1e-2, torch.float16: 2e-3}


def _flatten(outs):
    flat = []
    for o, sides in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            flat += list(s) if isinstance(s, tuple) else [s]
    return flat


def _rel_l2(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return (a - b).norm().item() / max(1e-30, b.norm().item())


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


def _argmax_gate(got, ref, what, agree):
    g, r = got.detach().cpu().double(), ref.detach().cpu().double()
    ga, ra = g.argmax(1), r.argmax(1)
    same = (ga == ra)
    frac = same.double().mean().item()
    err = (g - r).abs().max().item()
    if not bool(same.all()):
        top = r.max(1).values
        picked = r.gather(1, ga.unsqueeze(1)).squeeze(1)
        margin = (top - picked)[~same]
        assert float(margin.max()) <= 4 * err, \
            f"{what}: arg-max differs at a non-tie (margin {float(margin.max()):.3e}, err {err:.3e})"
    assert frac >= agree, f"{what}: arg-max agreement {frac:.5f} < {agree}"
    return frac


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(4, 96, 128), (1, 480, 640)])
def test_eval_16bit_vs_fp32_oracle(shape, dtype):
    """configs[4] arithmetic: eval forward with 16-bit activation storage"""
    from emsanet_amd import full_args
    from oracle.emsanet_oracle import synthetic_batch
    bs, h, w = shape
    args = full_args(input_height=h, input_width=w)
    model, oracle = _pair(args)
    model.set_compute_dtype(dtype)
    model.eval(), oracle.eval()
    batch = synthetic_batch(bs, h, w)
    with torch.no_grad():
        ref = _flatten(oracle(batch))
        out = _flatten(model({k: v.to(DEV) for k, v in batch.items()}))
    worst = 0.0
    for i, (a, b) in enumerate(zip(out, ref)):
        assert a.dtype == torch.float32, "model outputs stay fp32"
        assert torch.isfinite(a).all()
        e = _rel_l2(a, b)
        worst = max(worst, e)
        assert e <= OUT_TOL[dtype], f"output {i}: rel-L2 {e:.3e} > {OUT_TOL[dtype]:.0e}"
    fs = _argmax_gate(out[0], ref[0], 'semantic', AGREE[dtype])
    print(f"{dtype} eval {shape}: worst output rel-L2 {worst:.2e}, semantic arg-max agreement {fs:.5f}")


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_eval_16bit_vs_storage_emulating_oracle(dtype, monkeypatch):
    """the same eval forward against the fp64 oracle that ROUNDS WHERE THE ENGINE ROUNDS
    (oracle Spec.STORAGE): what remains is fp32 accumulation order and rounding-boundary flips"""
    from emsanet_amd import full_args
    from oracle import emsanet_oracle as O
    args = full_args(input_height=96, input_width=128)
    model, oracle = _pair(args)
    oracle = oracle.double()
    model.set_compute_dtype(dtype)
    model.eval(), oracle.eval()
    monkeypatch.setattr(O.Spec, 'STORAGE', dtype)
    batch = O.synthetic_batch(4, 96, 128)
    with torch.no_grad():
        ref = _flatten(oracle({k: v.double() for k, v in batch.items()}))
        out = _flatten(model({k: v.to(DEV) for k, v in batch.items()}))
    errs = [_rel_l2(a, b) for a, b in zip(out, ref)]
    print(f"{dtype} eval vs emulating oracle: rel-L2 " + ' '.join(f'{e:.1e}' for e in errs))
    assert max(errs) <= EMU_TOL[dtype]


def test_train_bf16_pinned_gradients(monkeypatch):
    """configs[2] arithmetic on one rank: bf16 train step (BatchNorm batch statistics, Dropout2d),
    fwd + bwd at the BASELINE resolution, against the fp64 oracle that (1) replays the engine's
    ReLU decisions and (2) rounds activations, their gradients and the conv weights to bf16 where
    the engine stores them.  (Against the PLAIN fp64 oracle the train-mode outputs of this
    random-weight network differ by 0.2-0.4 relative L2: every BatchNorm with batch statistics
    renormalises signal AND bf16 storage noise, ~0.5 % per block over ~100 layers -- measured with
    tools/stagewise_dtype.py; that is a property of 8-bit mantissas, not of the kernels.)"""
    import torch.nn.functional as F
    from emsanet_amd import full_args, ops
    from oracle import emsanet_oracle as O
    from test_model_gpu import _PinnedRelu
    args = full_args()
    model, oracle = _pair(args)
    oracle = oracle.double()
    model.set_compute_dtype(torch.bfloat16)
    for m in (model, oracle):
        m.train()
        m.dropout_seed = 321
    batch = O.synthetic_batch(2, 480, 640)
    ops.MASK_TRACE = []
    try:
        out = _flatten(model({k: v.to(DEV) for k, v in batch.items()}))
        trace = ops.MASK_TRACE
    finally:
        ops.MASK_TRACE = None
    pinned = _PinnedRelu(trace)
    monkeypatch.setattr(F, 'relu', pinned)
    monkeypatch.setattr(O.Spec, 'STORAGE', torch.bfloat16)
    ref = _flatten(oracle({k: v.double() for k, v in batch.items()}))
    assert pinned.i == len(trace)
    eo = [_rel_l2(a, b) for a, b in zip(out, ref)]
    print("bf16 train outputs rel-L2 vs emulating oracle:", ' '.join(f'{e:.1e}' for e in eo))
    cots = [rnd(*t.shape, seed=100 + i, scale=1e-1) for i, t in enumerate(ref)]
    torch.autograd.backward(out, [c.to(DEV) for c in cots])
    torch.autograd.backward(ref, [c.double() for c in cots])
    monkeypatch.undo()
    pr = dict(oracle.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in pr.values())
    errs, names = [], []
    for k, p in model.named_parameters():
        assert p.grad is not None and p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), k
        r = pr[k].grad
        if k.endswith(('conv1x3_1.bias', 'conv1x3_2.bias')) or r.abs().max().item() < 1e-9 * gmax:
            continue                              # mathematically zero (bias in front of a BN)
        errs.append(_rel_l2(p.grad, r))
        names.append(k)
    e = torch.tensor(errs)
    order = e.argsort(descending=True)[:5]
    print(f"bf16 train: {len(errs)} gradients vs emulating oracle, rel-L2 median {e.median():.2e} p95 "
          f"{e.quantile(0.95):.2e} max {e.max():.2e}; worst: "
          + ', '.join(f'{names[int(i)]} {e[int(i)]:.1e}' for i in order)
          + f"; {pinned.flips} of {pinned.total} ReLU decisions differ from the oracle's own")
    assert max(eo) <= EMU_TOL[torch.bfloat16], eo
    assert e.median().item() <= 2e-2 and e.max().item() <= 8e-2
    assert pinned.flips <= 2e-3 * pinned.total


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
def test_hipgraph_inference_16bit_matches_eager(dtype):
    """BASELINE configs[4]: whole-model hipGraph capture, 640x480, bs=1, 16-bit: the replay is
    bit-identical to the eager forward and depends on the input"""
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.graph import GraphedInference
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import synthetic_batch
    model = EMSANet(full_args(compute_dtype='bfloat16' if dtype == torch.bfloat16 else 'float16'),
                    nyuv2_config()).to(DEV).eval()
    assert model.compute_dtype == dtype
    b1 = {k: v.to(DEV) for k, v in synthetic_batch(1, 480, 640, seed=1).items()}
    b2 = {k: v.to(DEV) for k, v in synthetic_batch(1, 480, 640, seed=2).items()}
    g = GraphedInference(model, b1)
    with torch.no_grad():
        e1 = [t.clone() for t in _flatten(model(b1))]
        e2 = [t.clone() for t in _flatten(model(b2))]
    o1 = [t.clone() for t in _flatten(g(b1))]
    o2 = [t.clone() for t in _flatten(g(b2))]
    torch.cuda.synchronize()
    for a, b in zip(o1 + o2, e1 + e2):
        assert torch.equal(a, b)
    assert not torch.equal(o1[0], o2[0])


def test_bf16_training_step_with_losses_and_sgd():
    """the complete bf16 step: forward, all task losses (fp32, on the fp32 outputs), backward,
    fused SGD on fp32 master weights; finite, and the loss goes down over a few steps"""
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets
    from oracle.emsanet_oracle import synthetic_batch
    args = full_args(input_height=96, input_width=128, compute_dtype='bfloat16')
    torch.manual_seed(0)
    model = EMSANet(args, nyuv2_config()).to(DEV).train()
    batch = {k: v.to(DEV) for k, v in synthetic_batch(4, 96, 128).items()}
    params = [p for p in model.parameters() if p.requires_grad]
    buckets = GradientBuckets(params)
    opt = FusedSGD(buckets, lr=2e-3, momentum=0.9, weight_decay=1e-4)
    g = torch.Generator().manual_seed(3)
    tgt = None
    losses = []
    for _ in range(6):
        buckets.reset()
        flat = _flatten(model(batch))
        if tgt is None:
            tgt = [torch.randn(t.shape, generator=g).to(DEV) * 0.1 for t in flat]
        loss = sum(((a - b) ** 2).mean() for a, b in zip(flat, tgt))
        loss.backward()
        buckets.finish()
        opt.step()
        losses.append(float(loss))
    assert all(l == l for l in losses), losses
    assert losses[-1] < losses[0], losses
    assert all(p.dtype == torch.float32 for p in model.parameters())
