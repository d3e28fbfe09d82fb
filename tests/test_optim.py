"""Optimizer step and LR schedule.  The reference configures torch's own
SGD / Adam / AdamW / RAdam and OneCycleLR (/root/reference/emsanet/optimizer.py:29-57,
lr_scheduler.py:23-31), so torch IS the pinned oracle here."""
import pytest
import torch


@pytest.mark.parametrize('total,max_lr', [(500, 0.04), (20, 0.01), (7, 1.0)])
def test_one_cycle_matches_torch_scheduler(total, max_lr):
    from emsanet_amd.optim import one_cycle
    p = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.SGD([p], lr=0.01, momentum=0.9, nesterov=True, weight_decay=1e-4)
    sch = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=[max_lr], total_steps=total,
                                              div_factor=25, pct_start=0.1, anneal_strategy='cos',
                                              final_div_factor=1e4)
    for step in range(total):
        lr, mom = one_cycle(step, total, max_lr)
        g = opt.param_groups[0]
        assert abs(lr - g['lr']) <= 1e-12 * max_lr and abs(mom - g['momentum']) <= 1e-12
        opt.step()
        if step < total - 1:
            sch.step()
    with pytest.raises(ValueError):
        one_cycle(total, total, max_lr)


@pytest.mark.gpu
def test_fused_sgd_matches_torch_sgd():
    from emsanet_amd.optim import FusedSGD, one_cycle
    from emsanet_amd.parallel import GradientBuckets
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 64, 3, 1), (64,), (7,), (128, 64, 1, 1), (5, 3), (1,), (40, 128, 3, 3)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]
    topt = torch.optim.SGD(ref, lr=0.01, momentum=0.9, weight_decay=1e-4, nesterov=True)
    buckets = GradientBuckets(mine, bucket_bytes=20000)        # several buckets
    assert len(buckets.buckets) > 1
    opt = FusedSGD(buckets, lr=0.01, momentum=0.9, weight_decay=1e-4)
    for p, q in zip(mine, ref):                                  # parameters are views now
        assert torch.equal(p.detach().cpu(), q.detach())
        assert p.data_ptr() % 16 == 0                            # float4 kernels read them
    for step in range(4):
        lr, mom = one_cycle(step, 20, 0.05)
        for grp in topt.param_groups:
            grp['lr'], grp['momentum'] = lr, mom
        opt.set_schedule(lr, mom)
        buckets.reset()
        for i, (p, q) in enumerate(zip(mine, ref)):
            gr = torch.randn(p.shape, generator=g)
            q.grad = gr.clone()
            if not (step == 2 and i == 2):      # one parameter without a gradient in one step
                p.grad = gr.to(dev)
            else:
                q.grad = torch.zeros_like(q)
        buckets.finish()
        opt.step()
        topt.step()
        for p, q in zip(mine, ref):
            err = (p.detach().cpu() - q.detach()).abs().max().item()
            assert err <= 2e-6 * max(1.0, q.detach().abs().max().item()), (step, p.shape, err)
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    assert len(sd['momentum_buffers']) == len(buckets.buckets)


@pytest.mark.gpu
def test_fused_sgd_invalidates_packed_weight_caches():
    """FusedSGD writes the parameters through raw pointers; the engine's packed / Winograd weight
    caches (keyed on the parameters' version counters) must be refreshed afterwards: after one
    step the forward output equals that of an identical model stepped with torch.optim.SGD."""
    import copy
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from util import deterministic_state_dict
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets
    dev = 'cuda:0'
    args = full_args(input_height=64, input_width=96)
    m1 = EMSANet(args, nyuv2_config())
    m1.load_state_dict(deterministic_state_dict(m1))
    m1.to(dev).train()
    m2 = copy.deepcopy(m1)
    g = torch.Generator().manual_seed(0)
    batch = {'rgb': torch.randn(2, 3, 64, 96, generator=g).to(dev),
             'depth': torch.randn(2, 1, 64, 96, generator=g).to(dev)}

    def fwd_bwd(m):
        m.dropout_step = 0
        outs = m(batch)
        flat = [outs[0][0], *outs[1][0], outs[2][0]]
        torch.autograd.backward(flat, [torch.ones_like(t) * 1e-4 for t in flat])
        return [t.detach().clone() for t in flat]

    b1 = GradientBuckets(list(m1.parameters()))
    o1 = FusedSGD(b1, lr=1e-5, momentum=0.9, weight_decay=1e-4)
    o2 = torch.optim.SGD(m2.parameters(), lr=1e-5, momentum=0.9, weight_decay=1e-4, nesterov=True)
    b1.reset()
    before = fwd_bwd(m1)
    fwd_bwd(m2)
    v0 = next(m1.parameters())._version
    b1.finish()
    o1.step()
    o2.step()
    assert next(m1.parameters())._version > v0
    with torch.no_grad():
        m1.eval(); m2.eval()
        a = m1(batch)[0][0]
        b = m2(batch)[0][0]
    assert torch.isfinite(b).all() and (a - b).abs().max() <= 1e-4 * b.abs().max()
    # the cached Winograd / packed weights are those of the UPDATED parameters, bit for bit
    from emsanet_amd import functional as Fn
    changed = 0
    for rt in m1._pack_plan.rts[:40]:
        w = rt.conv.weight.detach()
        if rt.wino:
            assert torch.equal(rt._u, Fn.pack_wino(w)[0])
        else:
            assert torch.equal(rt._wp, Fn.pack_weight(w, 'fwd'))
    for p1, p2 in zip(m1.parameters(), m2.parameters()):
        assert (p1 - p2).abs().max() <= 2e-6 * max(1.0, float(p2.abs().max()))
        changed += int(p1.grad is not None)
    assert changed > 700 and before[0].isfinite().all()


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['adam', 'adamw', 'radam'])
@pytest.mark.parametrize('wd', [1e-4, 0.0])
def test_fused_adam_family_matches_torch(mode, wd):
    """the three other optimizers of the reference's factory (optimizer.py:37-57) under its one-cycle
    schedule, which cycles beta1 for them (torch's OneCycleLR, cycle_momentum left on,
    lr_scheduler.py:23-31): 9 steps -- RAdam's rectification switches on at step 6 -- several buckets,
    one parameter without a gradient in one step, state_dict round trip into a second optimizer"""
    from emsanet_amd.optim import FusedAdam
    from emsanet_amd.parallel import GradientBuckets
    dev = 'cuda:0'
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 64, 3, 1), (64,), (7,), (128, 64, 1, 1), (5, 3), (1,), (40, 128, 3, 3)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref]
    cls = {'adam': torch.optim.Adam, 'adamw': torch.optim.AdamW, 'radam': torch.optim.RAdam}[mode]
    topt = cls(ref, lr=0.01, weight_decay=wd, betas=(0.9, 0.999))
    total = 12
    sch = torch.optim.lr_scheduler.OneCycleLR(topt, max_lr=[0.01], total_steps=total, div_factor=25,
                                              pct_start=0.1, anneal_strategy='cos', final_div_factor=1e4)
    from emsanet_amd.optim import one_cycle
    buckets = GradientBuckets(mine, bucket_bytes=20000)
    assert len(buckets.buckets) > 1
    opt = FusedAdam(buckets, lr=0.01, weight_decay=wd, mode=mode)
    for step in range(9):
        lr, b1 = one_cycle(step, total, 0.01)
        grp = topt.param_groups[0]
        assert abs(grp['lr'] - lr) <= 1e-15 and abs(grp['betas'][0] - b1) <= 1e-15   # torch cycles beta1
        opt.set_schedule(lr, b1)
        buckets.reset()
        for i, (p, q) in enumerate(zip(mine, ref)):
            gr = torch.randn(p.shape, generator=g) * (10.0 ** (i - 3))         # gradient scales 1e-3 .. 1e3
            q.grad = gr.clone()
            if not (step == 2 and i == 2):
                p.grad = gr.to(dev)
            else:
                q.grad = torch.zeros_like(q)
        buckets.finish()
        opt.step()
        topt.step()
        sch.step()
        for p, q in zip(mine, ref):
            err = (p.detach().cpu() - q.detach()).abs().max().item()
            assert err <= 2e-6 * max(1.0, q.detach().abs().max().item()), (mode, step, p.shape, err)
    assert opt.step_count == 9
    # resume: a second optimizer over the same parameters continues identically
    sd = opt.state_dict()
    twin_params = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    tb = GradientBuckets(twin_params, bucket_bytes=20000)
    twin = FusedAdam(tb, lr=0.5, weight_decay=0.3, mode=mode)
    twin.load_state_dict(sd)
    buckets.reset(); tb.reset()
    for p, q in zip(mine, twin_params):
        gr = torch.randn(p.shape, generator=g).to(dev)
        p.grad, q.grad = gr, gr.clone()
    buckets.finish(); tb.finish()
    opt.step(); twin.step()
    assert twin.step_count == 10
    for p, q in zip(mine, twin_params):
        assert torch.equal(p.detach(), q.detach())
    with pytest.raises(ValueError):
        twin.load_state_dict({**sd, 'mode': 'sgd'})


def test_optimizer_and_schedule_factories_refuse_unknown_names():
    """same names and error as /root/reference/emsanet/optimizer.py:13,27-28 / lr_scheduler.py:8,19-20"""
    from emsanet_amd import default_args
    from emsanet_amd.optim import KNOWN_OPTIMIZERS, get_lr_schedule, get_optimizer, one_cycle
    assert KNOWN_OPTIMIZERS == ('adam', 'adamw', 'radam', 'sgd')
    a = default_args()
    assert (a.optimizer, a.learning_rate, a.momentum, a.weight_decay, a.learning_rate_scheduler,
            a.n_epochs) == ('sgd', 0.01, 0.9, 1e-4, 'onecycle', 500)
    with pytest.raises(ValueError, match="Unknown optimizer"):
        get_optimizer(default_args(optimizer='lamb'), None)
    with pytest.raises(ValueError, match="Unknown learning rate scheduler"):
        get_lr_schedule(default_args(learning_rate_scheduler='step'))
    sched = get_lr_schedule(default_args(n_epochs=20, learning_rate=0.03))
    assert sched(5) == one_cycle(5, 20, 0.03)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['sgd', 'adamw', 'radam'])
def test_factory_optimizers_in_a_captured_training_step(name):
    """`get_optimizer(args, buckets)` hands `GraphedTrainStep` either family: building the graph (eager
    warm-up steps + capture) leaves parameters, moments and the DEVICE-side step count untouched;
    every replay counts one step; the first replay moves the weights as the optimizer's first step
    does (Adam family: by ~lr per element whatever the gradient's scale)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.graph import GraphedTrainStep
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedAdam, FusedSGD, get_optimizer
    from emsanet_amd.parallel import GradientBuckets
    from oracle.emsanet_oracle import synthetic_batch
    dev = 'cuda:0'
    lr = 1e-4
    args = full_args(input_height=64, input_width=96, optimizer=name, learning_rate=lr, weight_decay=0.0)
    torch.manual_seed(0)
    m = EMSANet(args, nyuv2_config()).to(dev).train()
    b = GradientBuckets([p for p in m.parameters() if p.requires_grad])
    o = get_optimizer(args, b)
    assert isinstance(o, FusedSGD if name == 'sgd' else FusedAdam)
    batches = [{k: v.to(dev) for k, v in synthetic_batch(2, 64, 96, seed=s).items()} for s in (1, 2, 3)]

    def loss_of(out):
        flat = [out[0][0], *out[1][0], out[2][0]]
        return sum((t * t).mean() for t in flat)
    p0 = [p.detach().clone() for p in o.flat_params]
    g = GraphedTrainStep(m, batches[0], b, o, loss_fn=loss_of, warmup=2)
    torch.cuda.synchronize()
    for a, c in zip(o.flat_params, p0):
        assert torch.equal(a, c)
    if name != 'sgd':
        assert o.step_count == 0 and all(float(t.abs().max()) == 0 for t in o.exp_avg + o.exp_avg_sq)
    g.replay(batches[1])
    torch.cuda.synchronize()
    if name != 'sgd':
        assert o.step_count == 1
        d = torch.cat([(a - c).abs().reshape(-1) for a, c in zip(o.flat_params, p0)])
        if name == 'adamw':                 # first Adam step: lr * g / (|g| + eps) ~ lr wherever g != 0
            assert float(d.max()) <= lr * 1.001 and float(d.median()) >= 0.5 * lr
        else:                               # RAdam's first five steps are plain momentum steps: lr * g
            assert 0 < float(d.max()) < 1.0
    o.set_schedule(lr * 0.5, 0.93)
    g.replay(batches[2])
    torch.cuda.synchronize()
    if name != 'sgd':
        assert o.step_count == 2
        assert abs(float(o._hyper[0]) - lr * 0.5) < 1e-18 and abs(float(o._hyper[1]) - 0.93) < 1e-15
    assert all(torch.isfinite(t).all() for t in o.flat_params)
