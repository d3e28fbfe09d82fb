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
