"""Two ranks, real kernels: the data-parallel step of SURVEY.md §8(e) rehearsed on ONE GPU.

RCCL refuses two ranks on one device, so the ranks exchange through gloo (which accepts device
tensors) while every kernel of the engine runs on cuda:0 -- the flow is exactly the one
`bench.py --gpus N` runs under torch.distributed.run: broadcast of the start parameters, one
bs/GPU shard per rank, bucketed asynchronous all-reduce fired from the autograd hooks while the
backward pass is still running, fused SGD on the flat buckets."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, BS = 64, 96, 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flatten(outs):
    flat = []
    for o, sides in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            flat += list(s) if isinstance(s, tuple) else [s]
    return flat


def _batch(rank, dev):
    rng = np.random.default_rng(1234 + rank)
    rgb = rng.integers(0, 255, (BS, H, W, 3), dtype=np.uint8).astype(np.float32) / 255
    depth = rng.integers(0, 40000, (BS, H, W), dtype=np.uint16).astype(np.float32) / 20000
    return {'rgb': torch.from_numpy(rgb.transpose(0, 3, 1, 2).copy()).to(dev),
            'depth': torch.from_numpy(depth[:, None].copy()).to(dev)}


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['EMSA_DETERMINISTIC'] = '1'       # two-pass weight gradients: run-to-run identical
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets, broadcast_parameters

    torch.manual_seed(rank)                      # replicas start DIFFERENT; broadcast must fix it
    model = EMSANet(full_args(input_height=H, input_width=W), nyuv2_config()).to(dev)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('bn2.weight'):
                p.fill_(0.3)
    broadcast_parameters(model)
    model.train()
    batch = _batch(rank, dev)
    params = [p for p in model.parameters() if p.requires_grad]

    def backward():
        model.dropout_step = 0
        flat = _flatten(model(batch))
        g = torch.Generator().manual_seed(4321)
        cots = [(torch.randn(t.shape, generator=g) * 1e-2).to(dev).contiguous(
            memory_format=torch.channels_last if t.dim() == 4 else torch.contiguous_format)
            for t in flat]
        torch.autograd.backward(flat, cots)

    # (1) this rank's own gradients, no exchange; the expected result is their mean over ranks.
    # Taken twice: what differs between the two is the run-to-run rounding noise of the kernels
    # that accumulate with atomics, the yardstick for (2)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    backward()
    local = [p.grad.detach().clone() for p in params]
    for p in params:
        p.grad = None
    backward()
    again = [p.grad.detach().clone() for p in params]

    def mean_over_ranks(ts):
        out = []
        for g in ts:
            t = g.clone()
            dist.all_reduce(t)
            out.append(t / world)
        return out

    expect, expect2 = mean_over_ranks(local), mean_over_ranks(again)
    gmax = max(e.abs().max().item() for e in expect)

    def rel(a, b):
        # (biases in front of a train-mode BatchNorm have a mathematically zero gradient: only
        # rounding noise, so the yardstick has a floor relative to the largest gradient)
        return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-2 * gmax)

    noise = max(rel(a, b) for a, b in zip(expect2, expect))

    # (2) the product flow: hooks -> bucket gather -> async all-reduce -> average
    buckets = GradientBuckets(params, bucket_bytes=8 << 20)
    opt = FusedSGD(buckets, lr=1e-3, momentum=0.9, weight_decay=1e-4)
    buckets.reset()
    backward()
    buckets.finish()
    errs = [rel(p.grad, e) for p, e in zip(params, expect)]
    worst = max(errs)
    worst_name = names[errs.index(worst)]
    in_bucket = all(p.grad.data_ptr() == v.data_ptr()
                    for _, ps, views in buckets.buckets for p, v in zip(ps, views))
    opt.step()
    # (3) replicas stay bit-identical after the update
    digest = torch.stack([p.detach().double().sum() for p in params])
    both = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(both, digest)
    same = bool((both[0] == both[1]).all().item())
    differs = any((a - b).abs().max().item() > 0 for a, b in zip(local, expect))
    if rank == 0:
        ret.update(worst=worst, noise=noise, worst_name=worst_name, in_bucket=in_bucket, same=same, differs=differs,
                   n_buckets=len(buckets.buckets))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_training_step_on_one_gpu():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret['n_buckets'] > 1
    assert ret['differs'], "ranks saw the same data: the test would not notice a missing exchange"
    assert ret['in_bucket'], "gradients must live in the reduced flat buffers after finish()"
    # all-reduced bucket == mean of the ranks' own gradients, up to the run-to-run rounding noise
    print(f"bucket vs mean-of-ranks: worst {ret['worst']:.2e} ({ret['worst_name']}), "
          f"run-to-run noise {ret['noise']:.2e}")
    assert ret['worst'] <= max(1e-5, 4 * ret['noise']), (ret['worst'], ret['noise'],
                                                          ret['worst_name'])
    assert ret['same'], "replicas diverged after the fused SGD step"


def _seg_worker(rank, world, port, ret):
    """the multi-rank step as a chain of hipGraphs (one per backward segment,
    all-reduce issued eagerly between the replays) == the mean of the ranks' own gradients, the
    replicas stay identical, and every bucket but the last segment's is on the wire BEFORE the
    backward pass ends."""
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['EMSA_DETERMINISTIC'] = '1'
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.graph import SegmentedGraphedTrainStep, segment_parameter_groups
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets, broadcast_parameters

    torch.manual_seed(rank)
    model = EMSANet(full_args(input_height=H, input_width=W), nyuv2_config()).to(dev)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('bn2.weight'):
                p.fill_(0.3)
    broadcast_parameters(model)
    model.train()
    model.dropout_seed = 7
    batch = _batch(rank, dev)
    params = [p for p in model.parameters() if p.requires_grad]
    flat0 = _flatten(model(batch))
    g = torch.Generator().manual_seed(4321)
    cots = [(torch.randn(t.shape, generator=g) * 1e-2).to(dev).contiguous(
        memory_format=torch.channels_last if t.dim() == 4 else torch.contiguous_format)
        for t in flat0]
    del flat0
    model.dropout_step = 0

    # this rank's own gradients through the ORDINARY (uncut) backward pass
    torch.autograd.backward(_flatten(model(batch)), cots)
    local = [p.grad.detach().clone() for p in params]
    for p in params:
        p.grad = None
    model.dropout_step = 0
    expect = []
    for t in local:
        t = t.clone()
        dist.all_reduce(t)
        expect.append(t / world)
    gmax = max(e.abs().max().item() for e in expect)

    groups = segment_parameter_groups(model, (2, 1), decoder_cut=True)
    buckets = GradientBuckets(params, bucket_bytes=8 << 20, groups=groups, manual=True,
                              average=False, tail_bytes=4 << 20)
    opt = FusedSGD(buckets, lr=0.0, momentum=0.9, weight_decay=0.0)
    step = SegmentedGraphedTrainStep(model, batch, buckets, opt, cotangents=cots, cut_stages=(2, 1),
                                     decoder_cut=True)
    assert model.dropout_step == 0              # the warm-up was taken back
    step.replay(batch)
    torch.cuda.synchronize()
    # buckets hold the world SUM (average=False: FusedSGD folds 1/world into its update)
    errs = [(p.grad / world - e).abs().max().item() / max(e.abs().max().item(), 1e-2 * gmax)
            for p, e in zip(params, expect)]
    lead = step.bucket_launch_ms_before_backward_end()
    last_seg = set(buckets.group_buckets[-1])
    early = sum(1 for bi, ms in enumerate(lead) if bi not in last_seg and ms > 0.0)
    tail_bytes = sum(buckets.buckets[bi][0].numel() * 4 for bi in last_seg)
    # a second replay with lr > 0: replicas identical afterwards
    opt.set_schedule(1e-3, 0.9)
    step.replay(batch)
    torch.cuda.synchronize()
    digest = torch.stack([p.detach().double().sum() for p in params])
    both = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(both, digest)
    if rank == 0:
        seg_lead = [max(lead[bi] for bi in g) for g in buckets.group_buckets]
        ret.update(worst=max(errs), n_buckets=len(buckets.buckets), early=early, lead=lead, seg_lead=seg_lead,
                   n_last=len(last_seg), tail_bytes=tail_bytes,
                   same=bool((both[0] == both[1]).all().item()),
                   graphs=[i['nodes'] for i in step.graph_info])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_segmented_graph_step_on_one_gpu():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_seg_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    print(f"segmented graph step: {ret['n_buckets']} buckets, lead times (ms before backward end) "
          f"{ret['lead']}, graph nodes {ret['graphs']}, last segment {ret['tail_bytes']} bytes")
    # (same yardstick as the hook-driven test: a few weight gradients carry fp32-atomics jitter)
    assert ret['worst'] <= 1e-4, ret['worst']
    assert ret['same'], "replicas diverged after the graph-replayed update"
    # every bucket outside the LAST backward segment is launched before the backward pass ends;
    # the last segment (stem + layer1: the only gradients that cannot overlap) is small
    assert ret['early'] == ret['n_buckets'] - ret['n_last'] and ret['early'] >= ret['n_buckets'] - 1
    assert ret['tail_bytes'] <= 4 << 20
    # decoder_cut: five backward segments (decoder heads + later modules | first decoder modules +
    # context module | encoder stages 4-3 | stage 2 | stages 1-0), their buckets leave one segment
    # after the other: strictly decreasing lead times
    sl = ret['seg_lead']
    assert len(sl) == 5 and all(a > b for a, b in zip(sl, sl[1:])), sl


@pytest.mark.gpu
@pytest.mark.parametrize('cuts', [(2, 1), (3, 2, 1)])
@pytest.mark.parametrize('decoder_cut', [False, True])
@pytest.mark.parametrize('inst_fusion', ['add-rgb', 'add-depth'])
def test_segmented_step_equals_plain_step_single_process(inst_fusion, decoder_cut, cuts):
    """one process, no collectives: the segmented backward (cuts at the decoder boundary and
    behind encoder stages 2 and 1) gives the gradients of the ordinary backward pass, eagerly and
    replayed from its graphs; a second replay draws fresh Dropout2d masks.  'add-depth' for the
    instance decoder (emsanet/decoder.py:94-139 takes the fusion per decoder): the depth skips are
    cut too, so the decoder's skip gradients reach the depth encoder.  decoder_cut: a
    further cut behind the first module of both dense decoders, the loss is a root of two segments"""
    sys.path.insert(0, ROOT)
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.graph import SegmentedGraphedTrainStep, segment_parameter_groups
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = EMSANet(full_args(input_height=H, input_width=W,
                              instance_encoder_decoder_fusion=inst_fusion),
                    nyuv2_config()).to(dev).train()
    assert model._skip_streams_read() == ({'rgb'} if inst_fusion == 'add-rgb' else {'rgb', 'depth'})
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith('bn2.weight'):
                p.fill_(0.3)
    model.dropout_seed = 5
    batch = _batch(0, dev)
    params = [p for p in model.parameters() if p.requires_grad]

    def loss_of(out):
        return sum((t * t).mean() for t in _flatten(out))

    loss_of(model(batch)).backward()
    ref = [p.grad.detach().clone() for p in params]
    for p in params:
        p.grad = None
    model.dropout_step = 0
    # (cuts (3, 2, 1): bench.py's default for the multi-rank step since round 6 -- six / seven segments)
    groups = segment_parameter_groups(model, cuts, decoder_cut=decoder_cut)
    assert sum(len(g) for g in groups) == len(params) and len(groups) == 2 + len(cuts) + decoder_cut
    buckets = GradientBuckets(params, groups=groups, manual=True, tail_bytes=4 << 20)
    opt = FusedSGD(buckets, lr=0.0, momentum=0.9, weight_decay=0.0)
    step = SegmentedGraphedTrainStep(model, batch, buckets, opt, loss_fn=loss_of, cut_stages=cuts,
                                     decoder_cut=decoder_cut)
    gmax = max(float(r.abs().max()) for r in ref)
    loss_e, _ = step.eager_step(batch)
    for p, r in zip(params, ref):
        assert float((p.grad - r).abs().max()) <= 2e-5 * gmax
    model.dropout_step = 0
    model._sync_dropout_state()
    loss_g, _ = step.replay(batch)
    torch.cuda.synchronize()
    assert float(loss_g) == float(loss_e)
    for p, r in zip(params, ref):
        assert float((p.grad - r).abs().max()) <= 2e-5 * gmax
    l2 = float(step.replay(batch)[0])
    assert l2 != float(loss_e)                   # next step's masks
    assert all(i['memset_nodes'] == i['replaced'] for i in step.graph_info)


@pytest.mark.gpu
def test_segmented_step_falls_back_to_its_eager_twin_when_capture_fails(monkeypatch):
    """a runtime that refuses the capture (simulated: graph creation raises) must not cost a
    multi-GPU run: with eager_fallback=True the object survives, replay() runs the eager segmented
    step -- same gradients -- and capture_error says why; without it the error propagates"""
    sys.path.insert(0, ROOT)
    from emsanet_amd import full_args, graph as G, nyuv2_config
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = EMSANet(full_args(input_height=H, input_width=W), nyuv2_config()).to(dev).train()
    model.dropout_seed = 5
    batch = _batch(0, dev)
    params = [p for p in model.parameters() if p.requires_grad]

    def loss_of(out):
        return sum((t * t).mean() for t in _flatten(out))

    loss_of(model(batch)).backward()
    ref = [p.grad.detach().clone() for p in params]
    for p in params:
        p.grad = None
    model.dropout_step = 0

    def refuse(*a, **k):
        raise RuntimeError('capture refused (test)')
    monkeypatch.setattr(G.torch.cuda, 'graph', refuse)
    groups = G.segment_parameter_groups(model, (2, 1))
    buckets = GradientBuckets(params, groups=groups, manual=True, tail_bytes=4 << 20)
    opt = FusedSGD(buckets, lr=0.0, momentum=0.9, weight_decay=0.0)
    with pytest.raises(RuntimeError, match='capture refused'):
        G.SegmentedGraphedTrainStep(model, batch, buckets, opt, loss_fn=loss_of, cut_stages=(2, 1))
    model.dropout_step = 0
    with pytest.warns(RuntimeWarning, match='capture failed'):
        step = G.SegmentedGraphedTrainStep(model, batch, buckets, opt, loss_fn=loss_of,
                                           cut_stages=(2, 1), eager_fallback=True)
    assert step.graphs is None and 'capture refused' in step.capture_error
    model.dropout_step = 0
    model._sync_dropout_state()
    step.replay(batch)
    torch.cuda.synchronize()
    gmax = max(float(r.abs().max()) for r in ref)
    for p, r in zip(params, ref):
        assert float((p.grad - r).abs().max()) <= 2e-5 * gmax
    assert step.replays == 1 and step.bucket_launch_ms_before_backward_end() is None


@pytest.mark.gpu
def test_bench_self_launch_two_ranks_gloo():
    """`python bench.py --gpus 2` through its own launcher (torch.distributed.run on 127.0.0.1, the
    command line the driver uses), two ranks sharing this GPU over gloo (RCCL refuses two ranks on
    one device): the default N > 1 path is the segmented-hipGraph step, both ranks report, the JSON
    line carries the whole-job value and the kernels' roofline"""
    import json
    import subprocess
    env = dict(os.environ, EMSA_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--batch-size', '2',
                        '--steps', '2', '--warmup', '1', '--roofline-steps', '1', '--no-cpu-baseline'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['value'] > 0 and d['steps'] == 2
    c = d['comm']
    assert c['ranks_seen'] == [0, 1] and c['backend'] == 'gloo'
    assert c['path'] == 'segmented-graph' and len(c['graphs']) >= 2
    assert c['allreduce_bytes_per_step'] > 200e6        # ~63.5 M fp32 gradients
    assert d['roofline'] is not None and d['roofline']['frac'] > 0


@pytest.mark.gpu
def test_segmented_step_capture_and_fallback_leave_identical_parameters(monkeypatch):
    """the N > 1 code path of bench.py (segmented step, cuts (3, 2, 1) + decoder cut, manual buckets,
    fused SGD at lr > 0) once with its hipGraphs captured and once with the capture REFUSED (eager
    twin through `eager_fallback`): after a step both models hold the same parameters and BatchNorm
    statistics, to the fp32-atomics jitter of a few weight gradients."""
    sys.path.insert(0, ROOT)
    from emsanet_amd import full_args, graph as G, nyuv2_config
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets
    dev = torch.device('cuda', 0)
    batches = [_batch(i, dev) for i in range(3)]

    def loss_of(out):
        return sum((t * t).mean() for t in _flatten(out))

    def run(refuse):
        torch.manual_seed(0)
        model = EMSANet(full_args(input_height=H, input_width=W), nyuv2_config()).to(dev).train()
        model.dropout_seed = 5
        params = [p for p in model.parameters() if p.requires_grad]
        groups = G.segment_parameter_groups(model, (3, 2, 1), decoder_cut=True)
        buckets = GradientBuckets(params, groups=groups, manual=True, tail_bytes=4 << 20)
        opt = FusedSGD(buckets, lr=1e-3, momentum=0.9, weight_decay=1e-4)
        if refuse:
            def no(*a, **k):
                raise RuntimeError('capture refused (test)')
            monkeypatch.setattr(G.torch.cuda, 'graph', no)
            with pytest.warns(RuntimeWarning, match='capture failed'):
                step = G.SegmentedGraphedTrainStep(model, batches[0], buckets, opt, loss_fn=loss_of,
                                                   cut_stages=(3, 2, 1), decoder_cut=True,
                                                   eager_fallback=True)
            monkeypatch.undo()
            assert step.graphs is None
        else:
            step = G.SegmentedGraphedTrainStep(model, batches[0], buckets, opt, loss_fn=loss_of,
                                               cut_stages=(3, 2, 1), decoder_cut=True)
            assert step.graphs is not None and len(step.graphs) == 7
        before = {k: v.detach().clone() for k, v in model.state_dict().items()}
        step.replay(batches[1])
        torch.cuda.synchronize()
        return before, {k: v.detach().clone() for k, v in model.state_dict().items()}
    (b0, a), (b1, b) = run(False), run(True)
    assert set(a) == set(b)
    moved = 0.0
    for k in a:
        assert torch.equal(b0[k], b1[k]), f"{k}: building the step changed the state"
        if a[k].dtype.is_floating_point:
            upd = float((a[k] - b0[k]).abs().max())
            moved = max(moved, upd)
            d = float((a[k] - b[k]).abs().max())
            # ONE step (a second one would compound the chaos of train-mode BatchNorm on this tiny
            # configuration, cf. test_hipgraph_train_step_matches_eager): 1e-3 of the update + one fp32
            # ulp of the tensor -- the weight gradients of heads / 1x1 / strided convs carry atomics jitter
            assert d <= 1e-3 * upd + 2.5e-7 * float(b[k].abs().max()) + 1e-9, (k, d, upd)
        else:
            assert torch.equal(a[k], b[k]), k
    assert moved > 0.0
