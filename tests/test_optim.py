"""Optimizer step and LR schedule.  The reference configures torch's own
SGD / OneCycleLR (/root/reference/emsanet/optimizer.py:29-36, lr_scheduler.py:23-31), so torch IS
the pinned oracle here."""
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
