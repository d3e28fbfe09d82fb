"""CPU, world_size 2, gloo: the data-parallel gradient buckets average gradients across ranks
exactly like a single process on the concatenated batch (the only exchange step of the path,
SURVEY.md §8e)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net():
    torch.manual_seed(0)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1),
                         nn.ReLU(), nn.Flatten(), nn.Linear(8 * 6 * 6, 5))


def _worker(rank, world, port, bucket_bytes, ret):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from emsanet_amd.parallel import GradientBuckets, broadcast_parameters
    net = _net()
    if rank != 0:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)              # replicas start different; broadcast must fix it
    broadcast_parameters(net)
    buckets = GradientBuckets(list(net.parameters()), bucket_bytes=bucket_bytes)
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 3, 6, 6, generator=g)
    y = torch.randn(8, 5, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for _ in range(2):                   # second iteration checks reset()
        buckets.reset()
        loss = ((net(xs) - ys) ** 2).mean()
        loss.backward()
        buckets.finish()
    if rank == 0:
        ret['grads'] = [p.grad.clone() for p in net.parameters()]
        flat = buckets.buckets[0][0]
        ret['views'] = all(p.grad.data_ptr() >= buckets.buckets[buckets._bucket_of[p]][0].data_ptr()
                           for p in net.parameters())
        ret['n_buckets'] = len(buckets.buckets)
    dist.destroy_process_group()


@pytest.mark.parametrize('bucket_bytes', [256, 1 << 20])
def test_gradient_buckets_average_like_single_process(bucket_bytes):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, bucket_bytes, ret), nprocs=2, join=True)
    net = _net()
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 3, 6, 6, generator=g)
    y = torch.randn(8, 5, generator=g)
    # mean over the two shards' mean losses == mean over the concatenated batch
    loss = ((net(x) - y) ** 2).mean()
    loss.backward()
    for a, p in zip(ret['grads'], net.parameters()):
        assert torch.allclose(a, p.grad, rtol=1e-5, atol=1e-7)
    assert ret['n_buckets'] == (1 if bucket_bytes > 4096 else ret['n_buckets'])
    if bucket_bytes == 256:
        assert ret['n_buckets'] > 1
    assert ret['views']      # after finish() the gradients live in the reduced flat buffers


def test_single_process_is_passthrough():
    """world size 1: no copies, no collectives -- gradients are what autograd produced"""
    sys.path.insert(0, ROOT)
    from emsanet_amd.parallel import GradientBuckets
    net = _net()
    b = GradientBuckets(list(net.parameters()), bucket_bytes=1 << 20)
    x = torch.randn(2, 3, 6, 6)
    for _ in range(2):
        b.reset()
        assert all(p.grad is None for p in net.parameters())
        net(x).sum().backward()
        b.finish()
    ref = _net()
    ref(x).sum().backward()
    # every tensor starts on a 16-byte boundary inside its flat bucket
    assert b.n_bytes() == sum((p.numel() + 3) // 4 * 4 for p in net.parameters()) * 4
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.equal(p.grad, q.grad)


def _worker_modes(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from emsanet_amd.parallel import GradientBuckets
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 3, 6, 6, generator=g)
    y = torch.randn(8, 5, generator=g)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]

    # (1) a second backward pass without reset() must raise instead of silently diverging
    net = _net()
    b = GradientBuckets(list(net.parameters()), bucket_bytes=256)
    b.reset()
    ((net(xs) - ys) ** 2).mean().backward()
    try:
        ((net(xs) - ys) ** 2).mean().backward()
        raised = False
    except RuntimeError as e:
        raised = 'already all-reduced' in str(e)
    b.finish()

    # (2) average=False leaves the world SUM (FusedSGD folds 1/world into its update)
    net2 = _net()
    b2 = GradientBuckets(list(net2.parameters()), bucket_bytes=1 << 20, average=False)
    b2.reset()
    ((net2(xs) - ys) ** 2).mean().backward()
    b2.finish()
    sums = [p.grad.clone() for p in net2.parameters()]

    # (3) bf16 buckets on the wire
    net3 = _net()
    b3 = GradientBuckets(list(net3.parameters()), bucket_bytes=1 << 20, comm_dtype=torch.bfloat16)
    b3.reset()
    ((net3(xs) - ys) ** 2).mean().backward()
    b3.finish()
    if rank == 0:
        ret['raised'] = raised
        ret['sums'] = sums
        ret['bf16'] = [p.grad.clone() for p in net3.parameters()]
        ret['stats'] = dict(b3.stats)
        ret['bytes_f32'] = b2.stats['bytes']
    dist.destroy_process_group()


def test_bucket_modes_two_ranks():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_modes, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret['raised'], "second backward without reset() must raise"
    net = _net()
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 3, 6, 6, generator=g)
    y = torch.randn(8, 5, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    for s, a, p in zip(ret['sums'], ret['bf16'], net.parameters()):
        assert torch.allclose(s, 2.0 * p.grad, rtol=1e-5, atol=1e-7)         # SUM of 2 shard means
        assert torch.allclose(a, p.grad, rtol=2e-2, atol=1e-3 * float(p.grad.abs().max()))
    assert ret['stats']['collectives'] == 1 and ret['stats']['steps'] == 1
    assert ret['stats']['bytes'] * 2 == ret['bytes_f32']                       # half the bytes


def _worker_order(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from emsanet_amd.parallel import GradientBuckets, agree_on_order, record_arrival_order
    net = _net()
    params = list(net.parameters())
    pos = {id(p): i for i, p in enumerate(params)}
    # rank 1 proposes a different order: every rank must end up with rank 0's
    local = list(reversed(params)) if rank == 0 else params[1:] + params[:1]
    agreed = agree_on_order(params, local)
    ret[f'order{rank}'] = [pos[id(p)] for p in agreed]

    def run():
        ((net(torch.ones(2, 3, 6, 6)) ** 2).mean()).backward()
    measured = record_arrival_order(params, run)
    ret[f'measured{rank}'] = [pos[id(p)] for p in measured]
    buckets = GradientBuckets(params, order=measured, bucket_bytes=1024, tail_bytes=256)
    ret[f'layout{rank}'] = [[pos[id(p)] for p in b[1]] for b in buckets.buckets]
    dist.destroy_process_group()


def test_arrival_order_is_rank_zeros_on_every_rank():
    """bucket layouts must agree across ranks: the measured arrival order is broadcast from rank 0"""
    ret = mp.Manager().dict()
    mp.spawn(_worker_order, args=(2, _free_port(), ret), nprocs=2, join=True)
    n = len(list(_net().parameters()))
    assert ret['order0'] == list(reversed(range(n))) and ret['order1'] == ret['order0']
    assert ret['measured0'] == ret['measured1'] and sorted(ret['measured0']) == list(range(n))
    assert ret['layout0'] == ret['layout1'] and len(ret['layout0']) > 1


def test_grad_target_hands_out_each_view_once_per_step():
    sys.path.insert(0, ROOT)
    from emsanet_amd.parallel import GradientBuckets, grad_target
    net = _net()
    ps = list(net.parameters())
    assert grad_target(ps[0]) is None                  # no buckets: kernels allocate
    b = GradientBuckets(ps, bucket_bytes=1 << 20)
    v = grad_target(ps[0])
    assert v is not None and v.shape == ps[0].shape and v.data_ptr() % 16 == 0
    assert grad_target(ps[0]) is None                  # only once per step
    b.reset()
    assert grad_target(ps[0]) is not None
    ps[1].grad = torch.zeros_like(ps[1])
    assert grad_target(ps[1]) is None                  # accumulation: autograd must add


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` outside a launcher re-executes itself through
    torch.distributed.run (one process per GPU) instead of exiting"""
    sys.path.insert(0, ROOT)
    import argparse
    import subprocess
    import bench
    seen = {}
    monkeypatch.setattr(subprocess, 'call', lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 8)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '3'])
    assert bench.self_launch(argparse.Namespace(gpus=8)) == 0
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert '--nproc-per-node=8' in cmd and '127.0.0.1' in cmd
    assert cmd[-4:] == ['--gpus', '8', '--steps', '3'] and cmd[-5].endswith('bench.py')
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    monkeypatch.setattr(torch.cuda, 'device_count', lambda: 1)
    monkeypatch.delenv('EMSA_DIST_BACKEND', raising=False)
    with pytest.raises(SystemExit):
        bench.self_launch(argparse.Namespace(gpus=8))
