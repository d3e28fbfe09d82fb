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
