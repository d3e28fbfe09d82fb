"""The BACKWARD pass at the batch sizes `bench.py` times (VERDICT r5 "Missing 3": the fp32 step the
driver times is bs 32 and the bf16 one too, their gradients were verified at bs 2 / bs <= 8; split-K
depths, the multi-job 16-bit weight gradient, the >= 24 MiB bn1-fold rule and the 32-bit-index switch
all change with the batch size; ref backward: /root/reference/main.py:597-599).

The fp64 CPU oracle cannot finish a bs-32 640x480 step in the suite's budget, so the timed size is
checked through properties that do not need it:

  * LINEARITY in the batch (frozen BatchNorm, `model.eval()` with gradients on): every sample's
    forward is independent of the others and the loss cotangents are fixed per sample, so the
    parameter gradients of ONE bs-32 step equal the sum of the gradients of FOUR bs-8 steps on the
    same samples -- the bs-8 steps are the size the oracle-pinned tests cover
    (tests/test_model16_gpu.py:173), their split-K plans / index widths / tile rounds differ from the
    bs-32 ones;
  * TWO KERNEL FAMILIES at bs 32 in TRAIN mode (batch statistics, Dropout2d): the Winograd kernels
    (`conv1d_wino_kernel`, `conv_wgrad1d_wino_kernel`, bn1 folded into the loaders by the >= 24 MiB
    rule) against the implicit-GEMM family (`conv_igemm_kernel`, `conv_wgrad1d_kernel`, separate
    BatchNorm passes): different tilings, different split-K, different epilogues, same mathematics;
  * OPERATOR level against fp64 `F.conv2d`: the 3-tap weight gradient (fp32 Winograd form and the
    16-bit multi-job form) at n = 32, 120x160, C = 64 -- the 614 k-pixel reduction of the /4 stage;
  * configs[3] (R101, 960x736) at bs 16: batch consistency of the eval forward against bs 1 and
    bit-reproducibility of the train forward (twin of tests/test_model_gpu.py:429-470).
"""
import pytest
import torch
import torch.nn.functional as F

from util import DEV, close, rnd, to_act, deterministic_state_dict

pytestmark = pytest.mark.gpu


def _flatten(outs):
    flat = []
    for o, sides in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            flat += list(s) if isinstance(s, tuple) else [s]
    return flat


def _inputs(bs, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return {'rgb': torch.randn(bs, 3, h, w, generator=g).to(DEV),
            'depth': torch.randn(bs, 1, h, w, generator=g).to(DEV)}


def _model(args, dtype=None):
    from emsanet_amd import nyuv2_config
    from emsanet_amd.model import EMSANet
    model = EMSANet(args, nyuv2_config())
    model.load_state_dict(deterministic_state_dict(model))
    model.to(DEV)
    if dtype is not None:
        model.set_compute_dtype(dtype)
    return model


def _recalibrate(model, batch):
    """the deterministic state dict draws the running statistics at random; frozen statistics that do
    not belong to the weights leave the eval forward un-normalised (head logits saturate sigmoid /
    tanh).  One train-mode pass with momentum 1 writes the batch statistics into the buffers."""
    bns = [m for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    moms = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(batch)
    for m, mom in zip(bns, moms):
        m.momentum = mom
    model.eval()


def _grads(model, batch, cots, with_outputs=False):
    for p in model.parameters():
        p.grad = None
    out = _flatten(model(batch))
    assert len(out) == len(cots)
    torch.autograd.backward(out, cots)
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().double() for k, p in model.named_parameters() if p.grad is not None}
    return (grads, [t.detach() for t in out]) if with_outputs else grads


@pytest.fixture
def batch_invariant():
    """`emsa_set_batch_invariant(1)`: the SE / channel reductions partition a sample's pixels by the
    map size alone, so a sample's eval forward is the same bits in every batch (the default rule
    re-partitions launches with <= 64 workgroups for latency: bs 8 and bs 32 differ at the /16 stage,
    and a 1e-7 difference there flips ~1e-5 of the ReLU decisions downstream, which moves fp32
    gradients by ~5e-3 rel-L2 and decorrelates bf16 ones -- measured with the default rule: fp32
    worst rel-L2 9.2e-3, bf16 cosine down to 0.64)."""
    from emsanet_amd import _lib
    prev = _lib.lib().emsa_set_batch_invariant(1)
    yield
    _lib.lib().emsa_set_batch_invariant(prev)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_frozen_bn_gradients_are_linear_in_the_batch_at_the_timed_size(dtype, batch_invariant):
    """configs[1] (fp32) / configs[2] (bf16), 640x480, all heads: gradients of one bs-32 step ==
    sum over four bs-8 (bf16: two bs-16) steps on the same samples.  With batch-invariant reductions the partial forwards
    reproduce the bs-32 forward BIT FOR BIT per sample (asserted first), so both sides take the same
    ReLU / max-pool decisions and what is left is the summation order of the weight-gradient
    reductions over the batch (bf16 since round 6 too: the pyramid pooling's bilinear backward is a
    gather now, its fp32 atomics were the one non-reproducible link of the activation-gradient chain):
    every tensor within 1e-4 relative L2 and 2e-5 of the largest gradient, both storage types.  A lost split-K partial, a wrong launch plan
    or a 32-bit index overflow at the bs-32 sizes is an O(1/splits) .. O(1) error in its tensor."""
    from emsanet_amd import full_args
    # (bf16: two bs-16 steps -- below 4,096 pixels at the /32 stage conv_rs.hip plans 32-pixel tiles
    #  with two accumulator chains instead of 64-pixel tiles with one: a different summation order)
    bs, part, h, w = 32, (8 if dtype == torch.float32 else 16), 480, 640
    REL_TOL, ABS_TOL = 1e-4, 2e-5          # measured: fp32 3.6e-6 / 1.3e-6, bf16 1.3e-6 / 4.1e-7
    model = _model(full_args(), None if dtype == torch.float32 else dtype)
    batch = _inputs(bs, h, w, seed=11)
    _recalibrate(model, batch)
    with torch.no_grad():
        full = [t.clone() for t in _flatten(model(batch))]
        for i in range(0, bs, part):
            sub = _flatten(model({k: v[i:i + part].contiguous() for k, v in batch.items()}))
            for j, (a, b) in enumerate(zip(sub, full)):
                assert torch.equal(a, b[i:i + part]), \
                    f"output {j}, samples {i}..{i + part - 1}: the bs-{part} forward differs from bs-{bs}"
    shapes = [t.shape for t in full]
    del full
    g = torch.Generator().manual_seed(4321)
    cots = [(torch.randn(s, generator=g) * 1e-1).to(DEV) for s in shapes]
    big, out_big = _grads(model, batch, cots, with_outputs=True)
    acc = None
    for i in range(0, bs, part):
        sub = {k: v[i:i + part].contiguous() for k, v in batch.items()}
        gi, out_i = _grads(model, sub, [c[i:i + part].contiguous() for c in cots], with_outputs=True)
        for j, (a, b) in enumerate(zip(out_i, out_big)):     # (the autograd-mode forward: its own path)
            assert torch.equal(a, b[i:i + part]), f"grad-mode forward, output {j}, samples from {i}"
        acc = gi if acc is None else {k: acc[k] + gi[k] for k in acc}
    assert set(big) == set(acc) and len(big) > 700
    gmax = max(float(v.abs().max()) for v in big.values())
    rows, bad = [], []
    for k, a in big.items():
        b = acc[k]
        assert torch.isfinite(a).all(), k
        if float(b.abs().max()) < 1e-9 * gmax:
            assert float(a.abs().max()) <= 1e-6 * gmax, k
            continue
        e_abs = float((a - b).abs().max()) / gmax
        e_rel = float((a - b).norm() / b.norm())
        rows.append((e_rel, e_abs, float(a.norm() / b.norm()), k))
        if e_abs > ABS_TOL or (e_rel > REL_TOL and e_abs > 0.05 * ABS_TOL):
            bad.append(rows[-1])
    rows.sort(reverse=True)
    for r in rows[:6]:
        print("  rel-L2 %.2e  |diff|/gmax %.2e  ratio %.6f  %s" % r)
    print(f"bs-32 vs {bs // part} x bs-{part} ({dtype}): {len(rows)} gradients, worst rel-L2 {rows[0][0]:.2e}, "
          f"worst |diff| / gmax {max(r[1] for r in rows):.2e}")
    assert not bad, bad[:10]


def test_train_step_gradients_two_kernel_families_at_the_timed_size(monkeypatch):
    """configs[1] in TRAIN mode at bs 32 (batch statistics, Dropout2d, side outputs; the launch plan
    `bench.py` times): the default kernel set (1-D Winograd forward / data / weight gradients, bn1
    folded into the conv loaders where the tensor is >= 24 MiB, phased strided data gradients)
    against the implicit-GEMM set with separate BatchNorm passes.  Outputs within 3e-4 of their
    magnitude, every gradient within 1e-3 relative L2 (+ 2e-4 of the largest gradient for the
    tensors that are roundoff-sized)."""
    from emsanet_amd import full_args, functional as Fn
    bs, h, w = 32, 480, 640
    batch = _inputs(bs, h, w, seed=5)
    res = []
    for family in ('default', 'igemm'):
        if family == 'igemm':
            monkeypatch.setenv('EMSA_WINO', '0')
            monkeypatch.setenv('EMSA_WGRAD_WINO', '0')
            monkeypatch.setattr(Fn, 'BN1_FOLD', False)
            monkeypatch.setattr(Fn, 'DGRAD_PHASES', False, raising=False)
        model = _model(full_args())
        model.train()
        model.dropout_seed = 17
        model.dropout_step = 0
        with torch.no_grad():
            shapes = [t.shape for t in _flatten(model(batch))]
        model.load_state_dict(deterministic_state_dict(model))      # (running statistics moved)
        model.dropout_step = 0
        g = torch.Generator().manual_seed(99)
        cots = [(torch.randn(s, generator=g) * 1e-1).to(DEV) for s in shapes]
        for p in model.parameters():
            p.grad = None
        out = _flatten(model(batch))
        torch.autograd.backward(out, cots)
        torch.cuda.synchronize()
        res.append(([t.detach().double() for t in out],
                    {k: p.grad.detach().double() for k, p in model.named_parameters()}))
        del model, out
    (oa, ga), (ob, gb) = res
    for i, (a, b) in enumerate(zip(oa, ob)):
        e = float((a - b).abs().max()) / max(1.0, float(b.abs().max()))
        assert e <= 3e-4, f"output {i}: {e:.3e}"
    gmax = max(float(v.abs().max()) for v in gb.values())
    rows = []
    for k, a in ga.items():
        b = gb[k]
        assert torch.isfinite(a).all(), k
        if k.endswith(('conv1x3_1.bias', 'conv1x3_2.bias')) or float(b.abs().max()) < 1e-7 * gmax:
            assert float((a - b).abs().max()) <= 2e-4 * gmax, k       # mathematically zero
            continue
        rows.append((float((a - b).norm() / b.norm()), float(a.norm() / b.norm()),
                     float(torch.dot(a.flatten(), b.flatten()) / (a.norm() * b.norm())), k))
    rows.sort(reverse=True)
    print(f"train bs 32, Winograd + folds vs implicit GEMM: {len(rows)} gradients, worst rel-L2 "
          f"{rows[0][0]:.2e} ({rows[0][3]}), norm ratio {min(r[1] for r in rows):.4f} .. "
          f"{max(r[1] for r in rows):.4f}, cosine min {min(r[2] for r in rows):.5f}")
    # the two families' forwards agree to fp32 roundoff (asserted above), which flips ~1e-5 of the
    # ReLU decisions; with batch statistics every flip moves all upstream gradients (measured: stem
    # weight 1.6e-2 rel-L2 between the families, the same figure two eager runs of one family reach
    # in train mode at small sizes, tests/test_model_gpu.py::test_hipgraph_train_step_matches_eager).
    # The gates catch what a wrong launch plan at bs 32 would do (a lost partial, a wrong index
    # width: O(1 / splits) .. O(1) in its tensor), not roundoff:
    for e, ratio, cos, k in rows:
        assert e <= 5e-2 and abs(ratio - 1.0) <= 5e-2 and cos >= 0.998, (k, e, ratio, cos)


def _conv_ref(x, dy, k, p, dtype=None):
    """fp64 weight / bias gradient of a stride-1 conv (operands rounded to `dtype` first)"""
    from test_ops16_gpu import q
    c = x.shape[1]
    wt = torch.zeros(dy.shape[1], c, *k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(dy.shape[1], dtype=torch.float64, requires_grad=True)
    xs = x.double() if dtype is None else q(x, dtype)
    dys = dy.double() if dtype is None else q(dy, dtype)
    F.conv2d(xs, wt, b, padding=p).backward(dys)
    return wt.grad, b.grad


def test_wgrad_3tap_at_the_timed_reduction_length():
    """/4-stage NBt1D weight gradient at bs 32: 64 -> 64 channels, 120x160, 614,400 pixels per
    reduction (the largest operator test before had 38 k) against fp64 `F.conv2d`; the fp32 Winograd
    two-pass form (bit-reproducible) and the direct form"""
    from emsanet_amd import functional as Fn
    n, c, h, w = 32, 64, 120, 160
    for k, p, seed in (((1, 3), (0, 1), 1), ((3, 1), (1, 0), 2)):
        x = rnd(n, c, h, w, seed=seed)
        dy = rnd(n, c, h, w, seed=seed + 10, scale=0.1)
        rw, rb = _conv_ref(x, dy, k, p)
        spec = Fn.ConvSpec(c, c, k, (1, 1), p)
        like = torch.empty(c, c, *k, device=DEV)
        xa, da = to_act(x), to_act(dy)
        dw, db, packed = Fn.conv_wgrad(xa, da, spec, True, like=like, two_pass=True)
        assert not packed
        close(dw, rw, tol=2e-5, what=f'wgrad {k} two-pass')
        close(db, rb, tol=2e-5, what=f'dbias {k} two-pass')
        dw2, db2, _ = Fn.conv_wgrad(xa, da, spec, True, like=like, two_pass=True)
        assert torch.equal(dw, dw2) and torch.equal(db, db2)
        dwp, db3, packed = Fn.conv_wgrad(xa, da, spec, True)
        dw3 = Fn.unpack_wgrad(dwp, like) if packed else dwp
        close(dw3, rw, tol=2e-5, what=f'wgrad {k} atomics')
        close(db3, rb, tol=2e-5, what=f'dbias {k} atomics')


@pytest.mark.parametrize('c,h,w', [(64, 120, 160), (128, 60, 80)])
def test_wgrad16_multi_job_at_the_timed_reduction_length(c, h, w):
    """the bf16 NBt1D block's four weight gradients at the /4- and /8-stage sizes of the timed step
    (n = 32; 614,400 / 153,600 pixels per reduction), single launches and -- where the library has the
    form -- the one multi-job launch (emsa_conv_wgrad_multi_t), against fp64 on the rounded operands"""
    from emsanet_amd import functional as Fn
    from test_ops16_gpu import act16
    dtype = torch.bfloat16
    n = 32
    kinds = [((3, 1), (1, 0)), ((1, 3), (0, 1)), ((3, 1), (1, 0)), ((1, 3), (0, 1))]
    jobs, refs = [], []
    for j, (k, p) in enumerate(kinds):
        x = rnd(n, c, h, w, seed=30 + j)
        dy = rnd(n, c, h, w, seed=40 + j, scale=0.1)
        refs.append(_conv_ref(x, dy, k, p, dtype))
        spec = Fn.ConvSpec(c, c, k, (1, 1), p)
        jobs.append((act16(x, dtype), act16(dy, dtype), spec, torch.empty(c, c, *k, device=DEV),
                     None, None, True))
    out = Fn.conv_wgrad_multi(jobs)
    assert out is not None and len(out) == 4
    torch.cuda.synchronize()
    for j, (rw, rb) in enumerate(refs):
        x, dy, spec, like, _, _, _ = jobs[j]
        dw1, db1, packed = Fn.conv_wgrad(x, dy, spec, True, like=like, two_pass=True)
        assert not packed
        close(dw1, rw, tol=5e-5, what=f'wgrad16 job {j}')
        close(db1, rb, tol=5e-5, what=f'dbias16 job {j}')
        if out is not None:
            dw, db = out[j]
            close(dw, rw, tol=5e-5, what=f'multi wgrad16 job {j}')
            close(db, rb, tol=5e-5, what=f'multi dbias16 job {j}')
            close(dw, dw1.double().cpu(), tol=2e-5, what=f'multi vs single job {j}')


def test_config3_r101_bs16_batch_consistency_and_determinism():
    """BASELINE configs[3] at its own batch size (ResNet-101-NBt1D x2, 960x736, bs 16; no test ran it
    above bs 2): eval outputs of a sample do not depend on the batch it sits in (bs 16 vs bs 1, which
    test_config4_r101_highres_eval checks against the oracle), the train-mode forward is
    bit-reproducible, and a train step's gradients are finite and reproducible to atomics jitter."""
    from emsanet_amd import full_args
    h, w, bs = 736, 960, 16
    args = full_args(input_height=h, input_width=w, rgb_encoder_backbone='resnet101',
                     depth_encoder_backbone='resnet101')
    model = _model(args)
    batch = _inputs(bs, h, w, seed=3)
    _recalibrate(model, batch)
    with torch.no_grad():
        big = model(batch)
        for i in (0, 9, 15):
            one = model({k: v[i:i + 1].contiguous() for k, v in batch.items()})
            for (ob, _), (o1, _) in zip(big, one):
                obs = ob if isinstance(ob, tuple) else (ob,)
                o1s = o1 if isinstance(o1, tuple) else (o1,)
                for a, b in zip(obs, o1s):
                    ref = b[0].float()
                    err = (a[i].float() - ref).abs().max().item()
                    assert err <= 2e-4 * max(1.0, ref.abs().max().item()), (i, tuple(a.shape), err)
    del big
    model.train()
    outs = []
    for _ in range(2):
        model.dropout_step = 3
        with torch.no_grad():
            outs.append([t.clone() for t in _flatten(model(batch))])
    for a, b in zip(*outs):
        assert torch.equal(a, b), 'train-mode forward is not bit-reproducible'
    g = torch.Generator().manual_seed(1)
    cots = [(torch.randn(t.shape, generator=g) * 1e-1).to(DEV) for t in outs[0]]
    del outs
    grads = []
    for _ in range(2):
        model.dropout_step = 3
        grads.append(_grads(model, batch, cots))
    gmax = max(float(v.abs().max()) for v in grads[0].values())
    assert gmax > 0
    for k, a in grads[0].items():
        assert torch.isfinite(a).all(), k
        assert float((a - grads[1][k]).abs().max()) <= 1e-4 * gmax, k


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_train_step_gradients_are_reproducible_run_to_run(dtype):
    """TRAIN mode (batch statistics, Dropout2d at a fixed step, side outputs), 640x480 bs 8: two runs
    of the identical step.  The activation-gradient chain has no atomics since round 6 (the pyramid
    pooling's bilinear backward is a gather), the 1-D / 3x3 weight gradients are two-pass: what is left
    is the fp32-atomics jitter of the stem / 1x1 / strided / merged-head weight gradients and of the
    up-sampling weights, which feeds nothing.  Before the gather the bf16 step differed by 1.5e-2 on
    480 of its 742 gradients (tools/grad_repeat_probe.py): bf16 rounding turns a 1e-7 jitter of one
    activation gradient into a different noise realisation of everything upstream."""
    from emsanet_amd import full_args
    model = _model(full_args(), None if dtype == torch.float32 else dtype)
    model.train()
    model.dropout_seed = 23
    batch = _inputs(8, 480, 640, seed=2)
    model.dropout_step = 0
    with torch.no_grad():
        shapes = [t.shape for t in _flatten(model(batch))]
    g = torch.Generator().manual_seed(77)
    cots = [(torch.randn(s, generator=g) * 1e-1).to(DEV) for s in shapes]
    runs = []
    for _ in range(2):
        model.dropout_step = 5
        runs.append(_grads(model, batch, cots, with_outputs=True))
    (g0, o0), (g1, o1) = runs
    for a, b in zip(o0, o1):
        assert torch.equal(a, b), 'train-mode forward is not bit-reproducible'
    worst, n_diff = 0.0, 0
    gmax = max(float(v.abs().max()) for v in g0.values())
    for k, a in g0.items():
        b = g1[k]
        if torch.equal(a, b):
            continue
        n_diff += 1
        # (a conv bias in front of a train-mode BatchNorm has a mathematically ZERO gradient: what the
        #  atomics form returns for it is a cancelled sum, roundoff-sized and different every run --
        #  measured 0.15 relative on layer2.0.conv1x3_1.bias at 1e-7 of the largest gradient)
        if float((a - b).abs().max()) <= 1e-6 * gmax and float(b.abs().max()) <= 1e-4 * gmax:
            continue
        e = float((a - b).norm() / max(float(b.norm()), 1e-30))
        worst = max(worst, e)
        assert e <= 1e-5, f"{k}: run-to-run rel-L2 {e:.3e}"
    print(f"train step {dtype}: {n_diff} of {len(g0)} gradients differ run to run, worst rel-L2 {worst:.2e}")
    assert n_diff <= len(g0) // 4
