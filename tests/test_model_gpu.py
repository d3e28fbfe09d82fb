"""GPU engine vs the oracle (tests infrastructure) on identical inputs and weights.

Criteria
  * eval mode (BASELINE.json north_star): every raw output within 1e-3 of the fp32 CPU oracle
    (relative to the tensor's max magnitude); class maps argmax-exact except at numerical ties
    (pixels whose top-2 margin in the fp64 oracle is below the measured error), which must stay
    below 1e-4 of all pixels.
  * train-mode outputs (BatchNorm batch statistics over tiny test batches are ill-conditioned):
    error against the fp64 oracle <= max(1e-3, 4 x the error of the fp32 CPU oracle).
  * gradients of the ~100-layer ReLU network under random cotangents are chaotic at the 1e-3
    level for ANY fp32 implementation (a 1e-7 forward perturbation flips a few ReLU masks, each
    flip moves a cancellation-heavy gradient sum by ~1e-3; the fp32 CPU oracle shows it against
    fp64 too).  The tight gradient gates are therefore the op-level tests (test_ops_gpu.py,
    1e-4) and the block-level tests below (5e-4); the whole-model test checks that the engine's
    relative-L2 gradient error against fp64 is of the same class as the fp32 CPU oracle's, as a
    distribution over all parameter tensors whose gradient is not identically ~0:
    median <= max(2e-3, 4 x cpu median, 2 x chaos median), same for the 95th percentile with
    (1e-2, 8 x, 2 x), max <= 0.1 -- where "chaos" is the change of the fp64 oracle's OWN
    gradients when its input is perturbed by 1e-5 relative (measured: median 5e-3, max 1.2e-2:
    one flipped ReLU at the 3x4-pixel /32 stage shifts every upstream gradient by that much; the
    engine's fp32 forward error of ~2e-6 happens to cross that flip, the CPU's 5e-7 does not).
"""
import copy

import pytest
import torch

from util import DEV, close, rnd, to_act

pytestmark = pytest.mark.gpu

TOL = 1e-3


def _triple(args, seed=0):
    from emsanet_amd import nyuv2_config
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import EMSANetOracle, deterministic_state_dict
    cfg = nyuv2_config()
    o32 = EMSANetOracle(args, cfg)
    sd = deterministic_state_dict(o32, seed)
    o32.load_state_dict(sd)
    o64 = copy.deepcopy(o32).double()
    model = EMSANet(args, cfg)
    model.load_state_dict(sd)
    return model.to(DEV), o32, o64


def _flatten(outs):
    flat = []
    for o, sides in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            flat += list(s) if isinstance(s, tuple) else [s]
    return flat


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return (a - b).abs().max().item() / max(1.0, b.abs().max().item())


def _argmax_check(gpu_logits, ref64, what):
    g = gpu_logits.detach().cpu().double()
    ga, ra = g.argmax(1), ref64.argmax(1)
    diff = ga != ra
    n_diff = int(diff.sum())
    if n_diff == 0:
        return
    err = (g - ref64).abs().max().item()
    top = ref64.max(1).values
    picked = ref64.gather(1, ga.unsqueeze(1)).squeeze(1)
    margin = (top - picked)[diff]
    assert float(margin.max()) <= 4 * err + 1e-12, \
        f"{what}: argmax differs at a non-tie (margin {float(margin.max()):.3e}, err {err:.3e})"
    assert n_diff <= 1e-4 * diff.numel(), f"{what}: {n_diff} of {diff.numel()} argmax ties flipped"


@pytest.mark.parametrize('cin,cout,stride,p', [(64, 64, 1, 0.0), (64, 64, 1, 0.2),
                                              (64, 128, 2, 0.1), (128, 128, 1, 0.5)])
@pytest.mark.parametrize('mode', ['train', 'train_unfolded', 'eval_grad', 'eval_fast'])
def test_nbt1d_block(cin, cout, stride, p, mode, monkeypatch):
    """'train' runs bn1 folded into the loaders of conv3x1_2 and of its weight gradient (the fp32
    default), 'train_unfolded' with the separate normalise + ReLU pass"""
    from emsanet_amd import functional as Fn
    from emsanet_amd.nn import NonBottleneck1D
    from oracle import emsanet_oracle as O
    monkeypatch.setattr(Fn, 'BN1_FOLD', mode != 'train_unfolded')
    if mode == 'train_unfolded':
        mode = 'train'
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
    model, o32, o64 = _triple(args)
    model.eval(), o32.eval(), o64.eval()
    batch = synthetic_batch(2, 128, 160, modalities=('rgb',))
    with torch.no_grad():
        ref = o32(batch)
        ref64 = o64({k: v.double() for k, v in batch.items()})
        out = model({k: v.to(DEV) for k, v in batch.items()})
    close(out[0][0], ref[0][0], tol=TOL, what='semantic logits')
    _argmax_check(out[0][0], ref64[0][0], 'semantic')


@pytest.mark.parametrize('mode', ['eval_fast', 'eval_grad', 'train'])
def test_full_model_small(mode):
    """full RGB-D multi-task model at 96x128, bs=4: outputs, side outputs, all gradients"""
    from emsanet_amd import full_args
    from oracle.emsanet_oracle import synthetic_batch
    h, w, bs = 96, 128, 4
    args = full_args(input_height=h, input_width=w)
    model, o32, o64 = _triple(args)
    batch = synthetic_batch(bs, h, w)
    batch64 = {k: v.double() for k, v in batch.items()}
    gbatch = {k: v.to(DEV) for k, v in batch.items()}
    train = mode == 'train'
    for m in (model, o32, o64):
        m.train(train)
        m.dropout_seed = 1234
    ctx = torch.no_grad() if mode == 'eval_fast' else torch.enable_grad()
    with ctx:
        r32, r64, out = o32(batch), o64(batch64), model(gbatch)
    f32, f64, fo = _flatten(r32), _flatten(r64), _flatten(out)
    assert len(f32) == len(fo) == len(f64)
    for i, (a, b, c) in enumerate(zip(fo, f32, f64)):
        if train:
            lim = max(TOL, 4 * _rel(b, c))
            e = _rel(a, c)
            assert e <= lim, f"train output {i}: err vs fp64 {e:.3e} > {lim:.3e}"
        else:
            close(a, b, tol=TOL, what=f'{mode} output {i}')
    _argmax_check(fo[0], f64[0], 'semantic')
    if mode == 'eval_fast':
        return
    cots = [rnd(*t.shape, seed=100 + i, scale=1e-1) for i, t in enumerate(f32)]
    torch.autograd.backward(f32, cots)
    torch.autograd.backward(f64, [c.double() for c in cots])
    torch.autograd.backward(fo, [c.to(DEV) for c in cots])
    p32, p64 = dict(o32.named_parameters()), dict(o64.named_parameters())
    # chaos yardstick: fp64 oracle again with a 1e-5 relative input perturbation
    buffers64 = {k: b.clone() for k, b in o64.named_buffers()}   # (the second pass updates them)
    g64 = {k: p.grad.clone() for k, p in p64.items() if p.grad is not None}
    for p in o64.parameters():
        p.grad = None
    o64.dropout_step = 0
    gen = torch.Generator().manual_seed(1)
    pert = {k: v * (1 + 1e-5 * torch.randn(v.shape, generator=gen, dtype=torch.float64))
            for k, v in batch64.items()}
    torch.autograd.backward(_flatten(o64(pert)), [c.double() for c in cots])
    chaos = torch.tensor([(p64[k].grad - g64[k]).norm().item() / max(1e-30, g64[k].norm().item())
                          for k in g64])
    for k in g64:
        p64[k].grad = g64[k]
    e_gpu_all, e_cpu_all, names = [], [], []
    gmax = max(p64[k].grad.abs().max().item() for k in p64 if p64[k].grad is not None)
    for k, p in model.named_parameters():
        if not train and 'side_output_heads' in k:
            continue      # side heads are evaluated in training mode only
        assert p.grad is not None, f"no grad for {k}"
        assert torch.isfinite(p.grad).all(), f"non-finite grad for {k}"
        r = p64[k].grad
        before_bn = k.endswith(('conv1x3_1.bias', 'conv1x3_2.bias'))   # bias feeding a train-mode BN
        if before_bn or r.abs().max().item() < 1e-9 * gmax:
            continue      # mathematically zero (e.g. conv bias in front of a train-mode BN)
        den = max(1e-30, r.norm().item())
        e_gpu = (p.grad.detach().cpu().double() - r).norm().item() / den
        e_cpu = (p32[k].grad.double() - r).norm().item() / den
        e_gpu_all.append(e_gpu), e_cpu_all.append(e_cpu), names.append(k)
    eg, ec = torch.tensor(e_gpu_all), torch.tensor(e_cpu_all)
    med_g, med_c = eg.median().item(), ec.median().item()
    p95_g, p95_c = eg.quantile(0.95).item(), ec.quantile(0.95).item()
    worst = names[int(eg.argmax())]
    msg = (f"{mode}: rel-L2 grad error vs fp64  gpu median {med_g:.2e} p95 {p95_g:.2e} max "
           f"{eg.max().item():.2e} ({worst}) | cpu-fp32 median {med_c:.2e} p95 {p95_c:.2e} max "
           f"{ec.max().item():.2e}")
    print(msg)
    msg += f" | fp64 chaos(1e-5) median {chaos.median().item():.2e} p95 " \
           f"{chaos.quantile(0.95).item():.2e}"
    print(msg)
    assert med_g <= max(2e-3, 4 * med_c, 2 * chaos.median().item()), msg
    assert p95_g <= max(1e-2, 8 * p95_c, 2 * chaos.quantile(0.95).item()), msg
    assert eg.max().item() <= 0.1, msg
    if train:
        rb = buffers64
        for k, b in model.named_buffers():
            if 'running' in k:
                close(b, rb[k], tol=1e-3, what=f'buffer {k}')


def test_full_res_eval_bs1():
    """BASELINE config 2 shape (640x480 RGB-D, all heads), bs=1, eval: 1e-3 / argmax"""
    from emsanet_amd import full_args
    from oracle.emsanet_oracle import synthetic_batch
    args = full_args()
    model, o32, o64 = _triple(args)
    model.eval(), o32.eval(), o64.eval()
    batch = synthetic_batch(1, 480, 640)
    with torch.no_grad():
        ref = o32(batch)
        ref64 = o64({k: v.double() for k, v in batch.items()})
        out = model({k: v.to(DEV) for k, v in batch.items()})
    fr, f64, fo = _flatten(ref), _flatten(ref64), _flatten(out)
    for i, (a, b) in enumerate(zip(fo, fr)):
        close(a, b, tol=TOL, what=f'output {i}')
    _argmax_check(fo[0], f64[0], 'semantic')
    _argmax_check(fo[-1], f64[-1], 'scene')


def test_config4_r101_highres_eval():
    """BASELINE config 4 shape: ResNet-101-NBt1D dual encoder at 960x736 (json says 960x720,
    which is not divisible by 32 -- SURVEY.md 0.3), bs=1, eval"""
    from emsanet_amd import full_args
    from oracle.emsanet_oracle import synthetic_batch
    args = full_args(input_height=736, input_width=960, rgb_encoder_backbone='resnet101',
                     depth_encoder_backbone='resnet101')
    model, o32, o64 = _triple(args)
    model.eval(), o32.eval(), o64.eval()
    batch = synthetic_batch(1, 736, 960)
    with torch.no_grad():
        ref = o32(batch)
        ref64 = o64({k: v.double() for k, v in batch.items()})
        out = model({k: v.to(DEV) for k, v in batch.items()})
    fr, f64, fo = _flatten(ref), _flatten(ref64), _flatten(out)
    # 2 x 33 NBt1D blocks deep: fp32 roundoff of BOTH fp32 paths reaches the 1e-3 class on the
    # tanh/sigmoid-bounded outputs, so the fp64 oracle is the reference and the CPU fp32 path
    # the yardstick (same rule as the train-mode outputs)
    for i, (a, b, c) in enumerate(zip(fo, fr, f64)):
        lim = max(TOL, 4 * _rel(b, c))
        assert _rel(a, c) <= lim, f"output {i}: err vs fp64 {_rel(a, c):.3e} > {lim:.3e}"
        close(a, b, tol=2 * TOL, what=f'output {i} vs fp32 oracle')
    _argmax_check(fo[0], f64[0], 'semantic')


def test_resnet18_rgbd_train_step():
    """smaller backbone variant (reference test matrix uses resnet18, test_interface_model.py)"""
    from emsanet_amd import full_args
    from oracle.emsanet_oracle import synthetic_batch
    args = full_args(input_height=64, input_width=96, rgb_encoder_backbone='resnet18',
                     depth_encoder_backbone='resnet18')
    model, o32, o64 = _triple(args)
    for m in (model, o32, o64):
        m.train()
        m.dropout_seed = 7
    batch = synthetic_batch(4, 64, 96)
    r32, r64 = o32(batch), o64({k: v.double() for k, v in batch.items()})
    out = model({k: v.to(DEV) for k, v in batch.items()})
    for i, (a, b, c) in enumerate(zip(_flatten(out), _flatten(r32), _flatten(r64))):
        lim = max(TOL, 4 * _rel(b, c))
        assert _rel(a, c) <= lim, f"output {i}: {_rel(a, c):.3e} > {lim:.3e}"
    sum((t * t).mean() for t in _flatten(out)).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


def test_hipgraph_inference_matches_eager():
    """BASELINE config 5 shape (640x480, bs=1): whole-model hipGraph replay == eager forward"""
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.graph import GraphedInference
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import synthetic_batch
    model = EMSANet(full_args(), nyuv2_config()).to(DEV).eval()
    b1 = {k: v.to(DEV) for k, v in synthetic_batch(1, 480, 640, seed=1).items()}
    b2 = {k: v.to(DEV) for k, v in synthetic_batch(1, 480, 640, seed=2).items()}
    g = GraphedInference(model, b1)
    with torch.no_grad():
        e1 = [t.clone() for t in _flatten(model(b1))]
        e2 = [t.clone() for t in _flatten(model(b2))]
    o1 = [t.clone() for t in _flatten(g(b1))]
    o2 = [t.clone() for t in _flatten(g(b2))]
    torch.cuda.synchronize()
    for a, b in zip(o1, e1):
        assert torch.equal(a, b)
    for a, b in zip(o2, e2):
        assert torch.equal(a, b)
    assert not torch.equal(o1[0], o2[0])
    # weights changed after the capture (a fine-tuning step, load_state_dict): the graph holds
    # pointers to the packed weights of its warm-up runs -> it notices and re-captures
    assert g.captures == 1
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.01)
        e3 = [t.clone() for t in _flatten(model(b1))]
    o3 = [t.clone() for t in _flatten(g(b1))]
    torch.cuda.synchronize()
    assert g.captures == 2
    for a, b in zip(o3, e3):
        assert torch.equal(a, b)
    assert not torch.equal(o3[0], o1[0])


def test_missing_gpu_input_fails_loudly():
    from emsanet_amd import _lib, full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import synthetic_batch
    model = EMSANet(full_args(input_height=64, input_width=64), nyuv2_config()).to(DEV)
    with pytest.raises(_lib.EmsaError):
        model(synthetic_batch(1, 64, 64))      # CPU tensors: no CPU fallback


def test_library_loaded_before_torch_use_subprocess():
    """`__graft_entry__.build()` resolves the C-ABI before anything touched the GPU and `smoke()`
    may follow in the same process: the HIP runtime serving the library must still be torch's
    (the library is linked against /opt/rocm's libamdhip64, torch ships its own)"""
    import os
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import __graft_entry__ as g\n"
            "from emsanet_amd import _lib\n"
            "L = _lib.lib()\n"
            "import torch\n"
            "from emsanet_amd import functional as Fn\n"
            "x = torch.randn(2, 64, 8, 8, device='cuda')\n"
            "y, _ = Fn.maxpool_fwd(Fn.as_act(x))\n"
            "torch.cuda.synchronize()\n"
            "assert torch.equal(y, torch.nn.functional.max_pool2d(x, 3, 2, 1))\n"
            "print('ORDER_OK')\n")
    r = subprocess.run([sys.executable, '-c', code], cwd=ROOT, capture_output=True, text=True,
                       timeout=300)
    assert 'ORDER_OK' in r.stdout, r.stderr[-2000:]


def test_pack_plan_matches_per_layer_transforms():
    """the one-launch weight transform of the whole model == the per-layer kernels, bit for bit;
    it re-runs when a weight changes and when requires_grad flips"""
    from emsanet_amd import functional as Fn
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from util import deterministic_state_dict
    model = EMSANet(full_args(input_height=64, input_width=96), nyuv2_config())
    model.load_state_dict(deterministic_state_dict(model))
    model.to('cuda:0').train()
    plan = model._pack_plan
    assert len(plan.rts) > 200
    plan.refresh()
    n_wino = 0
    for rt in plan.rts:
        w = rt.conv.weight.detach()
        if rt.wino:
            u, ud = Fn.pack_wino(w, fwd=True, dgrad=True)
            assert torch.equal(rt._u, u) and torch.equal(rt._ud, ud)
            n_wino += 1
        else:
            wp, wpd = Fn.pack_weight_pair(w)
            assert torch.equal(rt._wp, wp) and torch.equal(rt._wpd, wpd)
    assert n_wino > 150
    # the merged / channel-padded head convs ride in the same launch: == their own re-pack path
    import copy as _copy
    from emsanet_amd import ops
    assert len(plan.multi) >= 9
    for m in plan.multi:
        twin = ops.MultiConvRT(m.placements, m.spec.cout, m.spec.cin, (m.spec.kh, m.spec.kw),
                               (m.spec.ph, m.spec.pw))
        wp, bias = twin.packed()
        if m.wino:
            assert m._wp is None and torch.equal(m._u, twin._u)
            ud = Fn.pack_wino_packed(twin.packed_dgrad(), m.spec.cin, m.spec.cout,
                                     Fn.wino_rows(m.spec), flip=True)
            assert torch.equal(m._hd[torch.float32][2], ud)
        else:
            assert torch.equal(m._wp, wp) and torch.equal(m._hd[torch.float32][1], twin.packed_dgrad())
        assert (bias is None) == (m._bias is None) and (bias is None or torch.equal(m._bias, bias))
        assert m._key == m._key_now()
    plan.refresh(torch.bfloat16)
    for m in plan.multi:
        twin = ops.MultiConvRT(m.placements, m.spec.cout, m.spec.cin, (m.spec.kh, m.spec.kw),
                               (m.spec.ph, m.spec.pw))
        wp16, _ = twin.packed_t(torch.bfloat16)
        assert torch.equal(m._h[torch.bfloat16][1], wp16)
        assert torch.equal(m._hd[torch.bfloat16][1], twin.packed_dgrad(torch.bfloat16))
    plan.refresh()
    rt = plan.rts[5]
    before = rt._u.clone() if rt.wino else rt._wp.clone()
    with torch.no_grad():
        rt.conv.weight.mul_(2.0)
    plan.refresh()
    after = rt._u if rt.wino else rt._wp
    assert torch.equal(after, before * 2.0)
    for p in model.parameters():
        p.requires_grad_(False)
    plan.refresh()
    assert all((r._ud if r.wino else None) is None for r in plan.rts if r.wino)


def test_full_size_batch_consistency_and_determinism():
    """BASELINE.json's bench size (bs=32, 640x480 RGB-D, all heads) through size-independent
    properties: (1) eval outputs of a sample do not depend on the batch it sits in -- sample i of
    the bs=32 forward equals the bs=1 forward of that sample (which test_full_res_eval_bs1 checks
    against the oracle); (2) the train-mode forward (batch statistics, hash dropout) is
    bit-reproducible run to run."""
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from util import deterministic_state_dict
    dev = 'cuda:0'
    args = full_args()
    model = EMSANet(args, nyuv2_config())
    model.load_state_dict(deterministic_state_dict(model))
    model.to(dev).eval()
    g = torch.Generator().manual_seed(7)
    rgb = torch.randn(32, 3, 480, 640, generator=g).to(dev)
    depth = torch.randn(32, 1, 480, 640, generator=g).to(dev)
    with torch.no_grad():
        big = model({'rgb': rgb, 'depth': depth})
        for i in (0, 17, 31):
            one = model({'rgb': rgb[i:i + 1].contiguous(), 'depth': depth[i:i + 1].contiguous()})
            for (ob, _), (o1, _) in zip(big, one):
                obs = ob if isinstance(ob, tuple) else (ob,)
                o1s = o1 if isinstance(o1, tuple) else (o1,)
                for a, b in zip(obs, o1s):
                    ref = b[0].float()
                    err = (a[i].float() - ref).abs().max().item()
                    # reductions split differently with the batch size (conv tile choice, SE pooling:
                    # 64-pixel chunks at batch 1, 512-pixel chunks at batch 32 since round 4 -- measured
                    # 1.02e-4 on the tanh offset map, <= 4e-5 on the other outputs; north_star's bound
                    # for the bs-1 forward against the oracle is 1e-3, test_full_res_eval_bs1)
                    assert err <= 2e-4 * max(1.0, ref.abs().max().item()), (i, tuple(a.shape), err)
    del big
    model.train()
    outs = []
    for _ in range(2):
        model.dropout_step = 3
        with torch.no_grad():
            o = model({'rgb': rgb, 'depth': depth})
        outs.append([t.clone() for t in (o[0][0], *o[1][0], o[2][0])])
    for a, b in zip(*outs):
        assert torch.equal(a, b), 'train-mode forward is not bit-reproducible'


# ---------------------------------------------------------------------------------------------
# mask-pinned gradient parity: the oracle replays the ENGINE's ReLU sign decisions
# ---------------------------------------------------------------------------------------------
class _PinnedRelu:
    """Replacement for torch.nn.functional.relu inside the oracle: call i multiplies by the i-th
    sign mask the engine recorded in its own forward (emsanet_amd.ops.MASK_TRACE).  Both sides
    then sit on the SAME piecewise-linear branch of the network, so gradients can be compared at
    a tight tolerance instead of through the chaos yardstick above.  Counts how many decisions
    the oracle would have taken differently (fp32 roundoff at |x| ~ 1e-6)."""

    def __init__(self, trace):
        self.trace, self.i, self.flips, self.total = trace, 0, 0, 0

    def __call__(self, x, inplace=False):
        tag, m = self.trace[self.i]
        self.i += 1
        m = m.cpu()
        assert m.numel() == x.numel(), f"mask {self.i - 1} ({tag}): {tuple(m.shape)} vs {tuple(x.shape)}"
        m = m.reshape(x.shape)
        self.flips += int(((x.detach() > 0) != m).sum())
        self.total += m.numel()
        return x * m.to(x.dtype)


def _pinned_grad_parity(args, bs, seed, monkeypatch, tol_out, tol_grad, oracle_dtype=torch.float64,
                        check_buffers=True, cfg=None, flips_rel=1e-5):
    """train-mode forward + backward of the engine vs the oracle on the engine's ReLU branch:
    every raw output within tol_out, EVERY parameter gradient within tol_grad (relative L2;
    gradients that are mathematically ~0 are compared against the global scale)"""
    import torch.nn.functional as F
    from emsanet_amd import nyuv2_config, ops
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import EMSANetOracle, deterministic_state_dict, synthetic_batch
    cfg = cfg if cfg is not None else nyuv2_config()
    oracle = EMSANetOracle(args, cfg)
    sd = deterministic_state_dict(oracle, 0)
    oracle.load_state_dict(sd)
    oracle = oracle.to(oracle_dtype)
    model = EMSANet(args, cfg)
    model.load_state_dict(sd)
    model.to(DEV)
    for m in (model, oracle):
        m.train()
        m.dropout_seed = seed
    batch = synthetic_batch(bs, args.input_height, args.input_width)
    ops.MASK_TRACE = []
    try:
        out = model({k: v.to(DEV) for k, v in batch.items()})
        trace = ops.MASK_TRACE
    finally:
        ops.MASK_TRACE = None
    pinned = _PinnedRelu(trace)
    monkeypatch.setattr(F, 'relu', pinned)
    ref = oracle({k: v.to(oracle_dtype) for k, v in batch.items()})
    assert pinned.i == len(trace), f"oracle made {pinned.i} ReLU calls, engine {len(trace)}"
    fo, fr = _flatten(out), _flatten(ref)
    assert len(fo) == len(fr)
    for i, (a, b) in enumerate(zip(fo, fr)):
        e = _rel(a, b)
        assert e <= tol_out, f"output {i}: rel err {e:.3e} > {tol_out:.1e}"
    _argmax_check(fo[0], fr[0].double(), 'semantic (train)')
    cots = [rnd(*t.shape, seed=100 + i, scale=1e-1) for i, t in enumerate(fr)]
    torch.autograd.backward(fo, [c.to(DEV) for c in cots])
    torch.autograd.backward(fr, [c.to(oracle_dtype) for c in cots])
    monkeypatch.undo()
    pr = dict(oracle.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in pr.values() if p.grad is not None)
    worst, worst_k, n, n_zero = 0.0, None, 0, 0
    for k, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
        r = pr[k].grad.double()
        g = p.grad.detach().cpu().double()
        before_bn = k.endswith(('conv1x3_1.bias', 'conv1x3_2.bias'))   # bias feeding a train-mode BN
        if before_bn or r.abs().max().item() < 1e-9 * gmax:
            # mathematically zero (a conv bias in front of a train-mode BatchNorm): the engine
            # must return fp32-roundoff-sized values (a cancelled sum over all pixels), not a
            # gradient
            assert g.abs().max().item() <= 1e-4 * gmax, f"{k}: should vanish, max {g.abs().max():.3e}"
            n_zero += 1
            continue
        e = (g - r).norm().item() / r.norm().item()
        n += 1
        if e > worst:
            worst, worst_k = e, k
    print(f"pinned parity: {n} gradients (+{n_zero} mathematically zero), worst rel-L2 {worst:.2e} ({worst_k}); "
          f"{pinned.flips} of {pinned.total} ReLU decisions differ from the oracle's own")
    assert worst <= tol_grad, f"gradient {worst_k}: rel-L2 {worst:.3e} > {tol_grad:.1e}"
    assert pinned.flips <= flips_rel * pinned.total + 2, "engine forward disagrees on too many signs"
    if check_buffers:
        rb = dict(oracle.named_buffers())
        for k, b in model.named_buffers():
            if 'running' in k:
                close(b, rb[k], tol=1e-3, what=f'buffer {k}')
    return worst


@pytest.mark.parametrize('fold', [True, False])
def test_pinned_gradients_small(fold, monkeypatch):
    """96x128 bs 4, all heads: all 766 gradients at 2e-3 relative L2 (was: distribution gate with
    max <= 0.1).  Measured 9.5e-4: at this size the /32 BatchNorms see 48 samples per channel and
    are ill-conditioned; the BASELINE-resolution test below holds 1e-3 (measured 4.6e-4).
    fold: bn1 of all 50 NBt1D blocks folded into the conv loaders / the data-gradient epilogue
    (the default rule only folds tensors >= 24 MiB: forced here) vs the separate passes."""
    from emsanet_amd import full_args, functional as Fn
    monkeypatch.setattr(Fn, 'BN1_FOLD', fold)
    _pinned_grad_parity(full_args(input_height=96, input_width=128), 4, 1234, monkeypatch,
                        tol_out=TOL, tol_grad=2e-3)


def test_pinned_gradients_five_tasks_with_normal(monkeypatch):
    """the reference's widest task set (tests/test_interface_model.py:129-132: semantic, instance,
    orientation, scene, normal): outputs and all gradients vs the fp64 oracle, three dense decoders"""
    from emsanet_amd import full_args
    args = full_args(input_height=96, input_width=128,
                     tasks=('semantic', 'instance', 'orientation', 'scene', 'normal'))
    _pinned_grad_parity(args, 4, 99, monkeypatch, tol_out=TOL, tol_grad=2e-3)


@pytest.mark.parametrize('fusion', ['add-rgbd', 'add'])
def test_pinned_gradients_rgbd_single_encoder(fusion, monkeypatch):
    """input modality 'rgbd' (/root/reference/emsanet/model.py:76-92,195-199): ONE ResNet-NBt1D over
    cat(rgb, depth) (4-channel stem), no encoder fusion modules, the decoders take the 'rgbd'
    skips; outputs and all gradients vs the fp64 oracle"""
    from emsanet_amd import full_args
    args = full_args(input_height=96, input_width=128, input_modalities=('rgbd',),
                     semantic_encoder_decoder_fusion=fusion, instance_encoder_decoder_fusion=fusion)
    _pinned_grad_parity(args, 4, 31, monkeypatch, tol_out=TOL, tol_grad=2e-3)


@pytest.mark.parametrize('backbone,block', [('resnet18', 'basicblock'), ('resnet34', 'basicblock'),
                                            ('resnet50', 'bottleneck')])
def test_pinned_gradients_other_resnet_blocks(backbone, block, monkeypatch):
    """`--*-encoder-backbone-resnet-block basicblock | bottleneck` (/root/reference/emsanet/args.py:159-166;
    ResNet-50 bottleneck: inference_time.bash:8,13, tests/test_interface_model.py:133 -- 256 ... 2048-channel
    stages: fp32 tensors with more than 1024 channels take the *_wide reduction kernels):
    train-mode outputs and all gradients vs the fp64 oracle, then the no-grad eval path (BatchNorm
    folded into the conv epilogues, residual added there) vs the oracle's eval forward."""
    from emsanet_amd import full_args
    from oracle.emsanet_oracle import synthetic_batch
    kw = dict(input_height=96, input_width=128, rgb_encoder_backbone=backbone,
              depth_encoder_backbone=backbone, rgb_encoder_backbone_resnet_block=block,
              depth_encoder_backbone_resnet_block=block)
    # (ResNet-50: 291 of 25.2 M ReLU decisions differ from the fp64 oracle's own, 1.2e-5 -- 16 + 33 layers
    #  deep per encoder, BatchNorm over 48 ... 768 samples at this size; every gradient within 1.5e-3)
    _pinned_grad_parity(full_args(**kw), 4, 17, monkeypatch, tol_out=TOL, tol_grad=2e-3,
                        flips_rel=3e-5 if block == 'bottleneck' else 1e-5)
    model, o32, o64 = _triple(full_args(**kw), seed=3)
    batch = synthetic_batch(2, 96, 128)
    for m in (model, o64):
        m.eval()
    with torch.no_grad():
        ref = o64({k: v.double() for k, v in batch.items()})
        out = model({k: v.to(DEV) for k, v in batch.items()})
    for i, (a, b) in enumerate(zip(_flatten(out), _flatten(ref))):
        assert _rel(a, b) <= TOL, f"eval output {i}: {_rel(a, b):.3e}"


@pytest.mark.parametrize('n_sem,n_scene', [(37, 21), (19, 5), (41, 3)])
def test_pinned_gradients_other_class_counts(n_sem, n_scene, monkeypatch):
    """the reference's other datasets change the head widths only (`dataset_config.semantic_label_list_
    without_void`, /root/reference/emsanet/model.py:39-43: SUNRGB-D 37 classes, Cityscapes 19, ...): the
    semantic head is then a channel-PADDED conv (37 -> 40, 19 -> 24, 41 -> 48) and its two learned
    up-samplings run on channel counts without a fused backward tile -- outputs and every gradient vs
    the fp64 oracle, as for NYUv2's 40"""
    from emsanet_amd import full_args
    from emsanet_amd.data import DatasetConfig
    _pinned_grad_parity(full_args(input_height=96, input_width=128), 3, 55, monkeypatch, tol_out=TOL,
                        tol_grad=2e-3, cfg=DatasetConfig(n_sem, n_scene))


@pytest.mark.parametrize('modes', [('nearest', 'nearest', 'nearest'), ('bilinear', 'bilinear', 'bilinear'),
                                   ('bilinear', 'learned-3x3-zeropad', 'nearest')])
def test_pinned_gradients_weight_free_decoder_upsampling(modes, monkeypatch):
    """`--semantic-decoder-upsampling / --instance-decoder-upsampling / --upsampling-prediction` with the
    weight-free modes (/root/reference/emsanet/args.py:280-298,363-372,439-448; handed to the decoders at
    emsanet/decoder.py:55-57,78,123): five tasks incl. the normal decoder, outputs and every gradient vs
    the fp64 oracle (F.interpolate); the state dict has no `upsampling.conv` keys where a mode has no
    weights; the twin-launch eval path falls back to separate launches"""
    from emsanet_amd import full_args
    sem, inst, pred = modes
    args = full_args(input_height=96, input_width=128,
                     tasks=('semantic', 'instance', 'orientation', 'scene', 'normal'),
                     semantic_decoder_upsampling=sem, instance_decoder_upsampling=inst,
                     normal_decoder_upsampling=sem, upsampling_prediction=pred)
    # (Dropout2d seed 24: with seed 23 and bilinear everywhere ONE squeeze-excitation linear's gradient
    #  nearly cancels -- |g| 2.6 beside siblings of 70..160 -- and its rel-L2 reads 7e-3 for an absolute
    #  error of the usual size; tools/upsampling_grad_probe.py lists the worst tensors per mode)
    _pinned_grad_parity(args, 3, 24, monkeypatch, tol_out=TOL, tol_grad=2e-3)
    from emsanet_amd import nyuv2_config
    from emsanet_amd.model import EMSANet
    keys = list(EMSANet(args, nyuv2_config()).state_dict())
    has = lambda frag: any(frag in k and 'upsampling.conv' in k or
                           (frag in k and '.upsampling.' in k and '.conv.' in k) for k in keys)   # noqa: E731
    assert has('instance_decoder.decoder_modules') == (inst == 'learned-3x3-zeropad')
    assert not has('semantic_decoder.decoder_modules') and not has('.head.upsampling')


def test_pinned_gradients_basicblock_decoders(monkeypatch):
    """`--semantic-decoder-block / --instance-decoder-block basicblock` (/root/reference/emsanet/args.py:
    325-331,401-407; `get_block_class(...)` at emsanet/decoder.py:68-71,100-103 -- the argument used to be
    ignored): the decoder modules chain basic blocks instead of NBt1D blocks; outputs and every gradient
    vs the fp64 oracle, and the state dict carries conv1 / bn1 / conv2 / bn2 under `decoder_modules.*.blocks`"""
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    args = full_args(input_height=96, input_width=128, semantic_decoder_block='basicblock',
                     instance_decoder_block='basicblock')
    _pinned_grad_parity(args, 3, 29, monkeypatch, tol_out=TOL, tol_grad=2e-3)
    m = EMSANet(args, nyuv2_config())
    keys = [k for k in m.state_dict() if 'semantic_decoder.decoder_modules.1.blocks.0.' in k]
    assert any(k.endswith('conv2.weight') for k in keys) and not any('conv3x1' in k for k in keys)
    # zero_residual_initialization (ref emsanet/model.py:188-190) reaches the basic blocks' last BatchNorm
    assert float(m.decoders['instance_decoder'].decoder_modules[2].blocks[1].bn2.weight.abs().sum()) == 0.0


def test_pinned_gradients_nearest_context_upsampling(monkeypatch):
    """`--upsampling-context-module nearest` (/root/reference/emsanet/args.py:250-256, handed to the
    context module at model.py:109-119; the engine used to ignore the argument): outputs and every
    gradient vs the fp64 oracle, which up-samples with F.interpolate(mode='nearest')"""
    from emsanet_amd import full_args
    args = full_args(input_height=96, input_width=128, upsampling_context_module='nearest')
    _pinned_grad_parity(args, 3, 21, monkeypatch, tol_out=TOL, tol_grad=2e-3)
    # and the two modes differ (the argument is honoured)
    from emsanet_amd import nyuv2_config
    from emsanet_amd.model import EMSANet
    from util import deterministic_state_dict
    outs = []
    for mode in ('bilinear', 'nearest'):
        m = EMSANet(full_args(input_height=96, input_width=128, upsampling_context_module=mode),
                    nyuv2_config())
        m.load_state_dict(deterministic_state_dict(m))
        m.to(DEV).eval()
        g = torch.Generator().manual_seed(1)
        b = {'rgb': torch.randn(1, 3, 96, 128, generator=g).to(DEV),
             'depth': torch.randn(1, 1, 96, 128, generator=g).to(DEV)}
        with torch.no_grad():
            outs.append(m(b)[0][0].clone())
    assert not torch.equal(outs[0], outs[1])


def test_cityscapes_like_geometry(monkeypatch):
    """a wide frame with 19 classes (the reference trains Cityscapes at 512x1024,
    /root/reference/README.md): eval forward at 512x1024 bs 1 vs the fp32 / fp64 oracle at north_star's
    1e-3 (other tile tails: 128 / 64 / 32 / 16 rows of 256 ... 32 pixels), then a train step at 256x512
    bs 2 with every gradient against the fp64 oracle on the engine's ReLU branch"""
    from emsanet_amd import full_args
    from emsanet_amd.data import DatasetConfig
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import EMSANetOracle, deterministic_state_dict, synthetic_batch
    cfg = DatasetConfig(19, 3)
    args = full_args(input_height=512, input_width=1024)
    o32 = EMSANetOracle(args, cfg)
    sd = deterministic_state_dict(o32, 0)
    o32.load_state_dict(sd)
    model = EMSANet(args, cfg)
    model.load_state_dict(sd)
    model.to(DEV).eval()
    o32.eval()
    batch = synthetic_batch(1, 512, 1024)
    with torch.no_grad():
        ref = _flatten(o32(batch))
        out = _flatten(model({k: v.to(DEV) for k, v in batch.items()}))
    assert out[0].shape == (1, 19, 512, 1024)
    for i, (a, b) in enumerate(zip(out, ref)):
        close(a, b, tol=TOL, what=f'output {i}')
    _argmax_check(out[0], ref[0].double(), 'semantic (512x1024)')
    del model, o32
    _pinned_grad_parity(full_args(input_height=256, input_width=512), 2, 8, monkeypatch, tol_out=TOL,
                        tol_grad=2e-3, cfg=cfg)


def test_pinned_gradients_baseline_resolution(monkeypatch):
    """BASELINE configs[1] shape: 640x480 RGB-D, all heads, train mode, bs=2 (what the fp64 CPU
    oracle finishes in seconds): every output and every one of the 742 gradients vs fp64; bn1 folded
    in every block (at bs=32 the default rule folds the /4, /8 and /16 stages)"""
    from emsanet_amd import full_args, functional as Fn
    monkeypatch.setattr(Fn, 'BN1_FOLD', True)
    _pinned_grad_parity(full_args(), 2, 77, monkeypatch, tol_out=TOL, tol_grad=1e-3)


def test_pinned_gradients_config4_r101_train(monkeypatch):
    """BASELINE configs[3]: ResNet-101-NBt1D dual encoder at 960x736, TRAIN step, bs=2, vs the
    fp32 CPU oracle on the engine's ReLU branch (fp64 at this size takes minutes)"""
    from emsanet_amd import full_args
    args = full_args(input_height=736, input_width=960, rgb_encoder_backbone='resnet101',
                     depth_encoder_backbone='resnet101')
    _pinned_grad_parity(args, 2, 5, monkeypatch, tol_out=2 * TOL, tol_grad=5e-3,
                        oracle_dtype=torch.float32)


@pytest.mark.parametrize('name,val', [('STEM_BIAS', True), ('DW_UPSAMPLE_BIAS', False),
                                      ('SIDE_OUTPUT_KERNEL', 3), ('SKIP_FUSION_1X1', 'always'),
                                      ('ORIENTATION_L2_NORMALIZE', True)])
def test_spec_switch_flips(name, val, monkeypatch):
    """each [U] switch of SURVEY App. A flipped on BOTH sides: engine == oracle, fwd + bwd"""
    from emsanet_amd import full_args, nn as enn
    from oracle import emsanet_oracle as O
    monkeypatch.setattr(enn.Spec, name, val)
    monkeypatch.setattr(O.Spec, name, val)
    args = full_args(input_height=96, input_width=128)
    if name == 'SKIP_FUSION_1X1':
        args.semantic_decoder_n_channels = (256, 128, 64)
        args.instance_decoder_n_channels = (256, 128, 64)
    # (96x128: the /32 BatchNorms and the SE squeeze see 48 / 12 samples; the tiny SE bias gradients are
    #  the worst tensors, measured up to 4.4e-3.  The L2-normalised orientation output divides by
    #  the vector norm, which a random-weight head leaves near zero at some of the 12 pixels of the
    #  /32 side output: fp32 roundoff of the summation order shows up amplified there -- measured
    #  1.6e-3 .. 2.8e-3 across kernel versions, a conditioning effect, not a kernel error)
    tol_out = 1e-2 if name == 'ORIENTATION_L2_NORMALIZE' else 2 * TOL
    _pinned_grad_parity(args, 4, 11, monkeypatch, tol_out=tol_out, tol_grad=1e-2)


def test_load_weights_surgery_then_forward(monkeypatch):
    """f-2 on the device: a checkpoint WITH orientation and 37 semantic classes goes through
    `load_weights` (orientation removal + 37 -> 40 class reuse, weights.py:28-56,95-107) into an
    engine without orientation; its eval forward equals the oracle loaded with the same
    surgically altered state dict"""
    from emsanet_amd import DatasetConfig, full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from emsanet_amd.weights import load_weights
    from oracle.emsanet_oracle import EMSANetOracle, deterministic_state_dict, synthetic_batch
    src_args = full_args(input_height=64, input_width=96)
    src = EMSANetOracle(src_args, DatasetConfig(37, 10))
    ckpt = deterministic_state_dict(src, 3)
    ckpt = {k.replace('encoder.', 'fused_encoders.', 1) if k.startswith('encoder.') else k: v
            for k, v in ckpt.items()}
    args = full_args(input_height=64, input_width=96, tasks=('semantic', 'scene', 'instance'))
    args.dataset = 'nyuv2'
    torch.manual_seed(0)
    model = EMSANet(args, nyuv2_config())
    load_weights(args, model, {k: v.clone() for k, v in ckpt.items()}, verbose=False)
    oracle = EMSANetOracle(args, nyuv2_config())
    oracle.load_state_dict(model.state_dict())          # strict: same keys and shapes
    own = model.state_dict()
    k = 'decoders.semantic_decoder.head.conv.weight'
    assert own[k].shape[0] == 40 and torch.equal(own[k][:37], ckpt[k])
    k = 'decoders.instance_decoder.head.shared_conv.conv.weight'
    assert own[k].shape[0] == 64 and torch.equal(own[k], ckpt[k][:64])
    model.to(DEV).eval(), oracle.eval()
    batch = synthetic_batch(2, 64, 96)
    with torch.no_grad():
        out = model({k: v.to(DEV) for k, v in batch.items()})
        ref = oracle(batch)
    for i, (a, b) in enumerate(zip(_flatten(out), _flatten(ref))):
        close(a, b, tol=TOL, what=f'output {i} after load_weights')


def test_hipgraph_train_step_matches_eager():
    """SURVEY 7 step 7 / VERDICT item 8: the whole training step (forward, backward, fused SGD)
    captured in a hipGraph == the eager step.  Two EAGER runs of this tiny configuration already
    drift apart by 1e-3 within three optimizer steps (the stem / 1x1 / strided / merged-head weight
    gradients are summed with fp32 atomics and BatchNorm over 48 samples amplifies the 1e-6
    jitter), so the comparison is arranged to be exact where the step is deterministic:
      * learning rate 0: the forward pass is bit-reproducible -> the losses of the replays equal the
        eager twin's BIT FOR BIT over several steps on changing batches, which needs fresh Dropout2d
        masks (device-side seed / step counter) and identical BatchNorm running statistics;
      * gradients of every parameter within 1e-4 of the eager ones (atomics jitter);
      * then one step with a learning rate set through the schedule between replays: the parameter
        UPDATE equals the eager update to 1e-3 (the kernel read the new lr from device memory)."""
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.graph import GraphedTrainStep
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets
    from oracle.emsanet_oracle import synthetic_batch
    args = full_args(input_height=96, input_width=128)

    def build():
        torch.manual_seed(0)
        m = EMSANet(args, nyuv2_config()).to(DEV).train()
        m.dropout_seed = 99
        params = [p for p in m.parameters() if p.requires_grad]
        b = GradientBuckets(params)
        o = FusedSGD(b, lr=0.0, momentum=0.9, weight_decay=0.0)
        return m, b, o
    batches = [{k: v.to(DEV) for k, v in synthetic_batch(4, 96, 128, seed=s).items()}
               for s in (1, 2, 3, 4)]

    def loss_of(out):
        return sum((t * t).mean() for t in _flatten(out))

    def eager_step(m, b, o, batch):
        b.reset()
        loss = loss_of(m(batch))
        loss.backward()
        b.finish()
        o.step()
        return float(loss.detach())

    # the constructor runs 3 eager warm-up steps on batches[0], takes their effects back
    # (parameters, momentum, BatchNorm statistics / counters, Dropout2d step: ADVICE r2) and
    # RECORDS (does not run) one step: the eager twin starts from the same initial state
    m2, b2, o2 = build()
    m3, b3, o3 = build()
    sd0 = {k: v.clone() for k, v in m2.state_dict().items()}
    g = GraphedTrainStep(m2, batches[0], b2, o2, loss_fn=loss_of, warmup=3)
    torch.cuda.synchronize()
    assert m2.dropout_step == m3.dropout_step == 0
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd0[k]), f"building the graph changed {k}"
    for batch in batches[1:]:
        l2, _ = g.replay(batch)
        l3 = eager_step(m3, b3, o3, batch)
        torch.cuda.synchronize()
        assert float(l2) == l3, (float(l2), l3)          # bit for bit
    gmax = max(float(p.grad.abs().max()) for p in m3.parameters())
    for (k, p2), (_, p3) in zip(m2.named_parameters(), m3.named_parameters()):
        d = float((p2.grad - p3.grad).abs().max())
        assert d <= 1e-4 * gmax, f"grad {k}: {d:.3e} vs max {gmax:.3e}"
    sd2, sd3 = m2.state_dict(), m3.state_dict()
    for k in sd2:
        assert torch.equal(sd2[k], sd3[k]), k            # parameters untouched, statistics equal
    assert m2.dropout_step == m3.dropout_step == 3
    k = next(k for k in sd2 if k.endswith('num_batches_tracked'))
    assert int(sd2[k]) == int(sd3[k]) == 3
    # replaying the same batch twice: new Dropout2d masks, different loss (as in eager mode)
    la = float(g.replay(batches[1])[0])
    lb = float(g.replay(batches[1])[0])
    assert la != lb
    eager_step(m3, b3, o3, batches[1])
    eager_step(m3, b3, o3, batches[1])
    # the schedule between replays: lr 0 -> 1e-4
    before = {k: v.clone() for k, v in m2.named_parameters()}
    o2.set_schedule(1e-4, 0.9)
    o3.set_schedule(1e-4, 0.9)
    g.replay(batches[2])
    eager_step(m3, b3, o3, batches[2])
    torch.cuda.synchronize()
    moved = 0.0
    for (k, p2), (_, p3) in zip(m2.named_parameters(), m3.named_parameters()):
        upd = float((p2 - before[k]).abs().max())
        moved = max(moved, upd)
        # (1e-3 of the update + one fp32 ulp of the parameter: the gradients carry atomics jitter)
        assert float((p2 - p3).abs().max()) <= 1e-3 * upd + 2.5e-7 * float(p3.abs().max()) + 1e-9, k
    assert moved > 0.0



def test_hipgraph_train_replay_then_eval_sees_new_weights():
    """a replay moves the parameters through raw pointers; every packed-weight
    cache of the engine keys on `_version`, which the captured `FusedSGD.step()` bumped only once,
    at capture time.  Flow of INTEGRATION.md: replay per batch, validate in between -- each eval
    forward (eager AND through GraphedInference) must see the weights of the LAST replay.  Checked
    against an eager twin at lr > 0: replay, eval, replay, eval."""
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.graph import GraphedInference, GraphedTrainStep
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets
    from oracle.emsanet_oracle import synthetic_batch
    args = full_args(input_height=96, input_width=128, dropout_p=0.0)

    def build():
        torch.manual_seed(0)
        m = EMSANet(args, nyuv2_config()).to(DEV).train()
        b = GradientBuckets([p for p in m.parameters() if p.requires_grad])
        o = FusedSGD(b, lr=0.01, momentum=0.9, weight_decay=0.0)
        return m, b, o
    batches = [{k: v.to(DEV) for k, v in synthetic_batch(4, 96, 128, seed=s).items()}
               for s in (1, 2, 3)]
    val = {k: v.to(DEV) for k, v in synthetic_batch(2, 96, 128, seed=9).items()}

    def loss_of(out):
        return sum((t * t).mean() for t in _flatten(out))

    def evaluate(m):
        m.eval()
        with torch.no_grad():
            out = [t.clone() for t in _flatten(m(val))]
        m.train()
        return out

    m2, b2, o2 = build()
    m3, b3, o3 = build()
    g = GraphedTrainStep(m2, batches[0], b2, o2, loss_fn=loss_of)
    m2.eval()
    gi = GraphedInference(m2, val)
    m2.train()
    evals2, evals3, graphed2 = [], [], []
    for batch in batches:
        g.replay(batch)
        b3.reset()
        loss_of(m3(batch)).backward()
        b3.finish()
        o3.step()
        evals2.append(evaluate(m2))
        m2.eval()
        graphed2.append([t.clone() for t in _flatten(gi(val))])
        m2.train()
        evals3.append(evaluate(m3))
    torch.cuda.synchronize()
    assert gi.captures == len(batches) + 1            # re-captured after every update
    def dist(xs, ys):
        num = sum(float((x.double() - y.double()).pow(2).sum()) for x, y in zip(xs, ys))
        den = sum(float(y.double().pow(2).sum()) for y in ys)
        return (num / den) ** 0.5

    for step, (e2, e3, q2) in enumerate(zip(evals2, evals3, graphed2)):
        for a, c in zip(e2, q2):
            assert torch.equal(a, c), f"step {step}: GraphedInference replayed stale weights"
        if step:
            # self-calibrating: the graphed model's eval must sit on the eager twin's CURRENT step,
            # far from the twin's previous one (the two runs differ by atomics jitter only; a stale
            # pack would reproduce the previous step's outputs)
            d_cur, d_prev = dist(e2, e3), dist(e2, evals3[step - 1])
            assert d_prev > 0.0 and d_cur <= 0.25 * d_prev, (step, d_cur, d_prev)


def _train_losses(a, bs, h, w):
    """TrainingLosses (HIP kernels, all supervised scales) + synthetic targets, as bench.py --losses"""
    from emsanet_amd.loss import TrainingLosses
    a.tasks_weighting = (1.0, 0.25, 3.0, 0.5)
    g = torch.Generator(device='cpu').manual_seed(99)
    crit = TrainingLosses(a, torch.rand(40, generator=g) * 2 + 0.3, 10).to(DEV)
    sem, inst = [], []
    for hh, ww in [(h, w)] + [(h // s, w // s) for s in (32, 16, 8)]:
        sem.append(torch.randint(0, 41, (bs, hh, ww), generator=g).to(DEV))
        fg = torch.rand(bs, hh, ww, generator=g) > 0.5
        inst.append({'center': (torch.rand(bs, 1, hh, ww, generator=g) ** 4).to(DEV),
                     'offset': (torch.rand(bs, 2, hh, ww, generator=g) * 2 - 1).to(DEV),
                     'foreground': fg.to(DEV),
                     'orientation': ((torch.rand(bs, hh, ww, generator=g) * 2 - 1) * 3.1415).to(DEV),
                     'orientation_foreground': (fg & (torch.rand(bs, hh, ww, generator=g) > 0.5)).to(DEV)})
    targets = {'semantic': sem, 'instance': inst,
               'scene': torch.randint(0, 11, (bs,), generator=g).to(DEV)}
    return crit, targets


@pytest.mark.parametrize('panoptic', [False, True])
def test_losses_on_the_merged_training_dictionary(panoptic):
    """the reference's training step hands `model(batch, do_postprocessing=True)` -- in TRAIN mode, the
    merged dictionary -- to its task helpers (/root/reference/main.py:126-141): TrainingLosses takes that
    dictionary as well as the raw list, same total and per-task losses bit for bit (same tensors, same
    kernels), also under --enable-panoptic, whose raw list is nested but whose dictionary is flat"""
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import synthetic_batch
    args = full_args(input_height=96, input_width=128, enable_panoptic=panoptic)
    torch.manual_seed(0)
    model = EMSANet(args, nyuv2_config()).to(DEV).train()
    crit, targets = _train_losses(args, 2, 96, 128)
    batch = {k: v.to(DEV) for k, v in synthetic_batch(2, 96, 128).items()}
    model.dropout_step = 0
    raw = model(batch)
    t_raw, l_raw = crit(raw, targets)
    model.dropout_step = 0
    merged = model(batch, do_postprocessing=True)
    assert isinstance(merged, dict) and len(merged['semantic_side_outputs']) == 3
    t_m, l_m = crit(merged, targets)
    assert float(t_raw) == float(t_m) and sorted(l_raw) == sorted(l_m)
    for k in l_raw:
        assert float(l_raw[k]) == float(l_m[k]), k
    t_m.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize('variant', ['bf16', 'losses', 'r101_960x736'])
def test_hipgraph_train_step_variants(variant):
    """the captured training step == the eager step beyond the fp32 96x128 case:
    bf16 storage, the complete step with all task losses on the device, ResNet-101 at 960x736
    (configs[3] shape).  Learning rate 0 and fresh Dropout2d masks per step: the losses of three
    replays on changing batches equal the eager twin's bit for bit (every forward reduction and the
    loss kernels are atomics-free), with eager work -- the twin's steps, host reads -- running
    between the replays."""
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.graph import GraphedTrainStep
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets
    from oracle.emsanet_oracle import synthetic_batch
    h, w, bs = (736, 960, 1) if variant.startswith('r101') else (96, 128, 4)
    kw = dict(input_height=h, input_width=w)
    if variant.startswith('r101'):
        kw.update(rgb_encoder_backbone='resnet101', depth_encoder_backbone='resnet101')
    if variant == 'bf16':
        kw.update(compute_dtype='bfloat16')
    args = full_args(**kw)
    crit = targets = None
    if variant == 'losses':
        crit, targets = _train_losses(args, bs, h, w)

    def loss_of(out):
        if crit is not None:
            return crit(out, targets)[0]
        return sum((t * t).mean() for t in _flatten(out))

    def build():
        torch.manual_seed(0)
        m = EMSANet(args, nyuv2_config()).to(DEV).train()
        m.dropout_seed = 17
        b = GradientBuckets([p for p in m.parameters() if p.requires_grad])
        return m, b, FusedSGD(b, lr=0.0, momentum=0.9, weight_decay=0.0)
    batches = [{k: v.to(DEV) for k, v in synthetic_batch(bs, h, w, seed=s).items()}
               for s in (1, 2, 3, 4)]
    m2, b2, o2 = build()
    m3, b3, o3 = build()
    g = GraphedTrainStep(m2, batches[0], b2, o2, loss_fn=loss_of, warmup=2)
    assert g.graph_info['memset_nodes'] == g.graph_info['replaced']
    for batch in batches[1:]:
        l2, _ = g.replay(batch)
        b3.reset()
        l3 = loss_of(m3(batch))
        l3.backward()
        b3.finish()
        o3.step()
        torch.cuda.synchronize()
        assert float(l2) == float(l3), (variant, float(l2), float(l3))
    gmax = max(float(p.grad.abs().max()) for p in m3.parameters())
    for (k, p2), (_, p3) in zip(m2.named_parameters(), m3.named_parameters()):
        if variant != 'bf16':
            assert float((p2.grad - p3.grad).abs().max()) <= 1e-4 * gmax, k
        else:
            # bf16 backward is not run-to-run identical: the few fp32-atomics kernels (bilinear /
            # pooling scatter, merged-head weight gradients) jitter in the last bit, the next bf16
            # store turns that into 1-ulp (0.4 %) flips, and ~100 layers later the first layers'
            # gradients of two EAGER runs differ by percents too -- direction and size must agree
            if k.endswith(('conv1x3_1.bias', 'conv1x3_2.bias')):
                continue              # mathematically zero (bias in front of a BatchNorm): noise
            a, b = p2.grad.double().flatten(), p3.grad.double().flatten()
            if float(b.norm()) > 1e-3 * gmax * b.numel() ** 0.5:
                cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
                # (measured over several boxes: cosine >= 0.9997 on the tiny SE tensors, norm ratio up
                #  to 1.10 there; bounds leave room for that run-to-run spread, a sign / scale error
                #  gives -1 / 0.5 / 2)
                assert cos >= 0.95 and 0.75 <= float(a.norm() / b.norm()) <= 1.33, (k, cos)


@pytest.mark.parametrize('panoptic', [False, True])
def test_hipgraph_inference_with_postprocessing(panoptic):
    """eval graph with `do_postprocessing=True` (+ `enable_panoptic`): every entry of the merged
    dict -- raw outputs, arg-max maps, instance centres / ids, panoptic ids -- replayed from the
    hipGraph == the eager forward, on two different inputs"""
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.graph import GraphedInference
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import synthetic_batch
    torch.manual_seed(0)
    model = EMSANet(full_args(input_height=192, input_width=256, enable_panoptic=panoptic),
                    nyuv2_config()).to(DEV).eval()
    b1 = {k: v.to(DEV) for k, v in synthetic_batch(2, 192, 256, seed=1).items()}
    b2 = {k: v.to(DEV) for k, v in synthetic_batch(2, 192, 256, seed=2).items()}
    g = GraphedInference(model, b1, do_postprocessing=True)
    assert g.graph_info['memset_nodes'] == g.graph_info['replaced']

    def flat(d):
        out = []
        for k in sorted(d):
            v = d[k]
            vs = [v] if torch.is_tensor(v) else [t for t in (v if isinstance(v, (list, tuple)) else [])
                                                 if torch.is_tensor(t)]
            out += [(k, t) for t in vs]
        return out
    with torch.no_grad():
        e1 = [(k, t.clone()) for k, t in flat(model(b1, do_postprocessing=True))]
        e2 = [(k, t.clone()) for k, t in flat(model(b2, do_postprocessing=True))]
    o1 = [(k, t.clone()) for k, t in flat(g(b1))]
    o2 = [(k, t.clone()) for k, t in flat(g(b2))]
    torch.cuda.synchronize()
    assert len(o1) == len(e1) and len(o1) >= 6
    for (k, a), (_, b) in zip(o1 + o2, e1 + e2):
        assert a.shape == b.shape and torch.equal(a, b), k
    assert any(not torch.equal(a, b) for (_, a), (_, b) in zip(o1, o2))


def test_graph_memset_repair_handles_chained_memsets():
    """two (three) zero-fills back to back on one stream are memset -> memset chains in the captured
    graph: the repair must re-read a node's edges when its turn comes (with edges taken up
    front, the second memset's dependency list named the already destroyed first one and the
    ordering kernel -> memset2 -> kernel was lost).  Replays must reproduce the eager result."""
    from emsanet_amd.graph import _new_graph, _repair_and_instantiate, _CAPTURE_MODE
    a = torch.ones(1 << 16, device=DEV)
    b = torch.ones(1 << 16, device=DEV)
    src = torch.arange(1 << 16, device=DEV, dtype=torch.float32)
    out = torch.empty(1 << 16, device=DEV)

    from emsanet_amd import _lib
    L = _lib.lib()

    def memset(t):          # (torch's own zero_() is a fill KERNEL; hipMemsetAsync captures as a MEMSET node)
        _lib.check(L.emsa_memset_async(t.data_ptr(), 0, t.numel() * t.element_size(),
                                       torch.cuda.current_stream().cuda_stream), 'emsa_memset_async')

    def work():
        a.add_(src)                 # kernel in front of the chain
        memset(a)                   # memset 1
        memset(b)                   # memset 2, chained behind memset 1
        memset(a)                   # memset 3, same buffer again
        b.add_(src)                 # kernels behind the chain
        torch.add(a, b, out=out)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        work()
    torch.cuda.current_stream().wait_stream(side)
    g, kept = _new_graph()
    if not kept:
        pytest.skip("this torch has no CUDAGraph(keep_graph=True)")
    with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
        work()
    info = _repair_and_instantiate(g, kept)
    assert info['memset_nodes'] >= 3 and info['replaced'] == info['memset_nodes'], info
    for _ in range(3):
        a.fill_(7.0)
        b.fill_(9.0)
        out.fill_(-1.0)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, src) and torch.equal(b, src) and float(a.abs().max()) == 0.0


def test_eval_after_stats_only_train_forward_sees_new_running_stats():
    """eval -> train-mode forward under no_grad (BatchNorm recalibration: no parameter changes, the
    running statistics are rewritten through raw pointers by emsa_bn_finalize) -> eval: the cached
    frozen-BatchNorm fold must be re-derived (the cache was keyed on tensor version
    counters that raw-pointer writes do not move).  Oracle: the same three calls in plain PyTorch."""
    from emsanet_amd.nn import NonBottleneck1D
    from oracle import emsanet_oracle as O
    torch.manual_seed(0)
    ref = O.NonBottleneck1D(64, 64, 1, 0.0)
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.momentum = 0.5
    blk = NonBottleneck1D(64, 64, 1, 0.0)
    blk.load_state_dict(ref.state_dict())
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = 0.5
    blk.to(DEV)
    x0, x1 = rnd(2, 64, 12, 16, seed=1), rnd(2, 64, 12, 16, seed=2, scale=3.0) + 1.0
    with torch.no_grad():
        ref.eval(), blk.eval()
        close(blk(to_act(x0)), ref(x0), tol=2e-4, what='eval before')
        before = blk(to_act(x0)).clone()
        ref.train(), blk.train()
        ref(x1), blk(to_act(x1))                       # statistics only
        ref.eval(), blk.eval()
        after = blk(to_act(x0))
        close(after, ref(x0), tol=2e-4, what='eval after recalibration')
        assert float((after - before).abs().max()) > 1e-2      # (the statistics did move)


def test_sanity_check_flow_of_main_py_leaves_running_statistics_alone():
    """/root/reference/main.py:486-498: before training, the reference forwards one batch in TRAIN mode
    with `track_running_stats = False` on every module that has the attribute, then switches it back on.
    The engine never calls nn.BatchNorm2d.forward, so it has to honour the flag itself: no running
    statistic and no step counter moves while it is off, batch statistics are still used (outputs ==
    the tracked pass's), and tracking resumes afterwards."""
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from util import deterministic_state_dict
    model = EMSANet(full_args(input_height=64, input_width=96), nyuv2_config())
    model.load_state_dict(deterministic_state_dict(model))
    model.to(DEV).train()
    model.dropout_seed = 3
    g = torch.Generator().manual_seed(9)
    batch = {'rgb': torch.randn(2, 3, 64, 96, generator=g).to(DEV),
             'depth': torch.randn(2, 1, 64, 96, generator=g).to(DEV)}
    before = {k: v.clone() for k, v in model.state_dict().items()}
    for m in model.modules():
        if hasattr(m, 'track_running_stats'):
            m.track_running_stats = False
    model.dropout_step = 0
    with torch.no_grad():
        out_off = [t.clone() for t in _flatten(model(batch))]
    after = model.state_dict()
    for k in before:
        assert torch.equal(before[k], after[k]), f"{k} moved with track_running_stats = False"
    for m in model.modules():
        if hasattr(m, 'track_running_stats'):
            m.track_running_stats = True
    model.dropout_step = 0
    with torch.no_grad():
        out_on = _flatten(model(batch))
    for a, b in zip(out_off, out_on):
        assert torch.equal(a, b)                       # batch statistics either way
    after = model.state_dict()
    moved = [k for k in before if 'running_mean' in k and not torch.equal(before[k], after[k])]
    assert len(moved) > 100
    k = next(k for k in after if k.endswith('num_batches_tracked'))
    assert int(after[k]) == int(before[k]) + 1
