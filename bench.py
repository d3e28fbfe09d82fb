#!/usr/bin/env python
# -*- coding: utf-8 -*-
"""
bench.py -- EMSANet forward+backward throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one training step of the full RGB-D multi-task EMSANet-R34-NBt1D on a synthetic
bs=32/GPU batch at 640x480, fp32: forward, backward driven by fixed output cotangents
(the task losses live in an un-vendored library), gradient all-reduce across ranks
(N>1, overlapped with backward) and the SGD-nesterov update of /root/reference/emsanet/
optimizer.py:29-36.  Inputs are resident in HBM before the timed region.  Weights: deterministic
random init; data: the reference's own synthetic generator (inference_time_whole_model.py:519-545).

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events around every 4th launch
(per kernel class) of the MFMA convolution kernels inside the timed steps (emsa_prof_* C-ABI;
bracketing every launch costs 3 % of the step, every 4th < 1 %); `cpu_baseline` times the
oracle (plain PyTorch CPU restatement, tests infrastructure) on the host cores for a bounded
sample of the same workload.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np    # noqa: E402
import torch          # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MFMA_BF16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16, dense
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec (6.3 TB/s measured copy)
FWD_GFLOP_PER_IMAGE = 119.672609792    # FlopCounterMode on the oracle at 480x640


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch-size', type=int, default=32, help='per GPU')
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-replicas', action='store_true',
                    help='cpu_baseline: additionally run the host-saturating form -- R pinned replicas of the '
                         'oracle step (R = usable CPUs / fastest thread count).  Opt-in: on the GPU boxes of '
                         'round 6 it took 5 minutes and reached 0.32 images/s with 15 x 16 threads where ONE '
                         '16-thread process reaches 2.3 (profiles/r06_l_f32_driver_cmd_with_cpu_replicas.json)')
    ap.add_argument('--cpu-replica-worker', type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-replica-threads', type=int, default=16, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-replica-start', type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-replica-seconds', type=float, default=12.0, help=argparse.SUPPRESS)
    ap.add_argument('--no-kernel-timing', action='store_true',
                    help='do not bracket conv launches with HIP events (A/B of the overhead)')
    ap.add_argument('--timing-every', type=int, default=5,
                    help='bracket every n-th launch of each conv kernel class with HIP events '
                         '(5: coprime to the 4 launches an NBt1D block issues per class, so every '
                         'position of the block is sampled)')
    ap.add_argument('--cpu-seconds', type=float, default=45.0,
                    help='bound of the timed part of the CPU baseline (3 warm-up + up to 10 timed '
                         'iterations, SURVEY 8d; at least 3 timed)')
    ap.add_argument('--eval', action='store_true', help='inference-only forward (not the metric)')
    ap.add_argument('--force-dist', action='store_true',
                    help='validation: RCCL process group + bucketed all-reduce path with ONE rank')
    ap.add_argument('--backbone', default='resnet34', choices=('resnet18', 'resnet34', 'resnet50', 'resnet101'),
                    help='ResNet of both encoders (BASELINE config 4: resnet101 at 960x736)')
    ap.add_argument('--resnet-block', default='nonbottleneck1d',
                    choices=('nonbottleneck1d', 'basicblock', 'bottleneck'),
                    help='block of both encoders (ref --*-encoder-backbone-resnet-block; every BASELINE '
                         'config: nonbottleneck1d; resnet50 + bottleneck = inference_time.bash:8,13)')
    ap.add_argument('--torch-optimizer', action='store_true',
                    help='A/B: torch.optim.SGD (foreach) instead of the fused bucket-wise SGD kernel')
    ap.add_argument('--losses', action='store_true',
                    help='complete training step: all task losses on device (semantic / scene CE, '
                         'instance MSE / L1 / von Mises, multi-scale, reference weights) and '
                         'loss.backward() instead of fixed output cotangents')
    ap.add_argument('--dtype', default='f32', choices=('f32', 'bf16', 'f16'),
                    help='storage type of the activations: f32 = the reference arithmetic and the '
                         'headline metric (BASELINE configs[1]); bf16 = configs[2] mixed precision '
                         '(fp32 master weights, statistics, accumulation, outputs); f16: --eval only')
    ap.add_argument('--grad-dtype', default=None, choices=('f32', 'bf16'),
                    help='dtype of the gradient all-reduce buckets on the wire (bf16: half the xGMI '
                         'bytes, SURVEY 8e).  Default: the storage type of the run -- f32 for --dtype '
                         'f32, bf16 for --dtype bf16 (mixed precision: the gradients were computed '
                         'from bf16 activations anyway; the fp32 master copy and the optimizer state '
                         'stay fp32)')
    ap.add_argument('--h2d', action='store_true',
                    help='also time the same steps with every batch staged from pinned host memory '
                         '(raw uint8/uint16 frames, overlapped copy, on-device normalisation); '
                         'reported as h2d_staged beside the HBM-resident value')
    ap.add_argument('--roofline-steps', type=int, default=5,
                    help='single-stream steps behind the timed region for the per-kernel roofline '
                         '(the timed region overlaps launches on two streams)')
    ap.add_argument('--graph', action='store_true',
                    help='replay the step from a hipGraph: with --eval the whole-model forward '
                         '(BASELINE config 5 shape), otherwise the whole training step')
    ap.add_argument('--protocol', default='resident', choices=('resident', 'reference'),
                    help='--eval only.  resident (default): inputs in HBM before the timed region, '
                         'K forward passes between two synchronisations.  reference: the timing '
                         'protocol of /root/reference/inference_time_whole_model.py:297-347 -- per run '
                         'one event pair around {host->device copy of the inputs, forward, '
                         'device->host copy of every main output}, synchronised per run, '
                         '--warmup (reference: 20) + --steps (80) runs, FPS = mean(1/t) +- std; '
                         'reported as `reference_protocol` beside the resident value')
    ap.add_argument('--cut-stages', default='3,2,1',
                    help='multi-rank segmented-graph step: encoder stages AFTER which the backward pass is '
                         'cut (one hipGraph per segment, the buckets of a finished segment all-reduced while '
                         'the next one runs)')
    ap.add_argument('--protocol-reps', type=int, default=1,
                    help='--protocol reference: repeat every form this many times (same process, same '
                         'box) and report the median repetition beside the list (VERDICT r5: one '
                         'repetition per box scattered 230 .. 304 FPS)')
    ap.add_argument('--eager', action='store_true',
                    help='launch the training step kernel by kernel from the host instead of replaying it '
                         'from a hipGraph (the hipGraph step is the default for training at every N: the '
                         'eager 16-bit step is host-bound, the fp32 replay is +0.6 %% ahead of its eager step)')
    return ap.parse_args()


def synthetic_batch_device(bs, h, w, seed, device):
    # same generator as oracle.synthetic_batch / the reference's timing script
    rng = np.random.default_rng(seed)
    rgb = rng.integers(0, 255, (bs, h, w, 3), dtype=np.uint8)
    depth = rng.integers(0, 40000, (bs, h, w), dtype=np.uint16)
    return {
        'rgb': torch.from_numpy((rgb.astype(np.float32) / 255).transpose(0, 3, 1, 2).copy()).to(device),
        'depth': torch.from_numpy((depth.astype(np.float32) / 20000)[:, None].copy()).to(device),
    }


def deterministic_init_(model, seed=0):
    """random-init weights of the architecture (no checkpoints available): keep the reference's
    constructor init but give BatchNorm non-trivial statistics and the decoder residual gammas a
    non-zero value so that no branch of the backward pass is multiplied by exact zeros."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('bn2.weight'):
                p.copy_(torch.empty(p.shape).uniform_(0.2, 0.4, generator=g))


def training_losses_and_targets(a, bs, h, w, dev):
    """--losses: TrainingLosses with the reference's published task weighting (README.md:621-622)
    and synthetic targets of every supervised scale (full resolution, /32, /16, /8)"""
    from emsanet_amd.loss import TrainingLosses
    a.tasks_weighting = (1.0, 0.25, 3.0, 0.5)
    g = torch.Generator(device='cpu').manual_seed(99)
    class_weights = torch.rand(40, generator=g) * 2 + 0.3
    crit = TrainingLosses(a, class_weights, 10).to(dev)
    sizes = [(h, w)] + [(h // s, w // s) for s in (32, 16, 8)]
    sem, inst = [], []
    for hh, ww in sizes:
        sem.append(torch.randint(0, 41, (bs, hh, ww), generator=g).to(dev))
        fg = torch.rand(bs, hh, ww, generator=g) > 0.5
        inst.append({'center': (torch.rand(bs, 1, hh, ww, generator=g) ** 4).to(dev),
                     'offset': (torch.rand(bs, 2, hh, ww, generator=g) * 2 - 1).to(dev),
                     'foreground': fg.to(dev),
                     'orientation': ((torch.rand(bs, hh, ww, generator=g) * 2 - 1) * 3.1415).to(dev),
                     'orientation_foreground': (fg & (torch.rand(bs, hh, ww, generator=g) > 0.5)).to(dev)})
    targets = {'semantic': sem, 'instance': inst,
               'scene': torch.randint(0, 11, (bs,), generator=g).to(dev)}
    return crit, targets


def flatten_outputs(outs):
    flat = []
    for o, sides in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
        for s in sides:
            flat += list(s) if isinstance(s, tuple) else [s]
    return flat


def cpu_replica_worker(args):
    """one replica of the host-saturating CPU baseline (`cpu_replicas`): pin to its core range, one
    warm-up step, then timed steps; prints {t0, t1, n} (wall-clock window of the timed steps)"""
    from emsanet_amd import full_args, nyuv2_config
    from oracle.emsanet_oracle import EMSANetOracle, deterministic_state_dict, synthetic_batch
    slot, nt = args.cpu_replica_worker, args.cpu_replica_threads
    try:
        os.sched_setaffinity(0, range(slot * nt, (slot + 1) * nt))
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(nt)
    a = full_args(input_height=args.height, input_width=args.width,
                  rgb_encoder_backbone=args.backbone, depth_encoder_backbone=args.backbone,
                  rgb_encoder_backbone_resnet_block=args.resnet_block,
                  depth_encoder_backbone_resnet_block=args.resnet_block)
    o = EMSANetOracle(a, nyuv2_config())
    o.load_state_dict(deterministic_state_dict(o, 0))
    o.train()
    batch = synthetic_batch(2, args.height, args.width)

    def step():
        for p in o.parameters():
            p.grad = None
        flat = flatten_outputs(o(batch))
        torch.autograd.backward(flat, [torch.full_like(t, 1e-3) for t in flat])
    step()
    # (start together: every replica waits for the wall-clock second the parent named)
    while time.time() < args.cpu_replica_start:
        time.sleep(0.005)
    t0 = time.time()
    n = 0
    while n < 3 or time.time() - t0 < args.cpu_replica_seconds:
        step()
        n += 1
    print(json.dumps({'t0': t0, 't1': time.time(), 'n': n}), flush=True)


def usable_cpus():
    """CPUs this process may actually use: the scheduler affinity mask capped by the cgroup's CPU
    quota (a container on a 256-core host is usually given far fewer)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != 'max':
                n = min(n, max(1, int(int(quota) / int(period))))
        except Exception:                               # noqa: BLE001
            pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0:
            n = min(n, max(1, q // per))
    except Exception:                                   # noqa: BLE001
        pass
    return n


def cpu_replicas(args, threads, all_cores, bs):
    """the host-saturating form of the CPU baseline: R = cores / threads independent
    replicas of the oracle step, each pinned to its own `threads` cores (the fastest single-process
    thread count), started together; value = images of all replicas / the wall-clock window in which
    ALL of them were inside their timed steps' span (min start .. max end).  Bounded: at most 16
    replicas, ~8 GB of free host memory per replica, 12 s of timed steps."""
    import subprocess
    try:
        import psutil
        free_gb = psutil.virtual_memory().available / 2 ** 30
    except Exception:                                   # noqa: BLE001
        free_gb = 64.0
    reps = int(min(all_cores // max(1, threads), 16, free_gb // 8))
    if reps < 2:
        return {'value': None, 'replicas': reps,
                'sample': f'not run: {all_cores} usable CPUs / {threads} threads, {free_gb:.0f} GB free'}
    start = time.time() + 45.0          # (imports + oracle construction + one warm-up step per replica)
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-replica-threads', str(threads),
           '--cpu-replica-start', repr(start), '--cpu-replica-seconds', '12', '--height', str(args.height),
           '--width', str(args.width), '--backbone', args.backbone, '--resnet-block', args.resnet_block]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads),
               HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    procs = [subprocess.Popen(cmd + ['--cpu-replica-worker', str(i)], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, env=env, text=True) for i in range(reps)]
    rows = []
    for p_ in procs:
        try:
            out, _ = p_.communicate(timeout=240)
            rows.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:                               # noqa: BLE001
            p_.kill()
    if len(rows) < 2:
        return None
    span = max(r['t1'] for r in rows) - min(r['t0'] for r in rows)
    late = max(r['t0'] for r in rows) - start
    return {'value': round(sum(r['n'] for r in rows) * bs / span, 4), 'unit': 'images/s',
            'replicas': len(rows), 'threads_each': threads, 'cores': len(rows) * threads,
            'sample': f'{len(rows)} replicas x {threads} threads (pinned core ranges) of the same oracle step, '
                      f'started together, {sum(r["n"] for r in rows)} fwd+bwd iterations of bs={bs} in a '
                      f'{span:.1f} s window (slowest replica started {late:.1f} s after the common start)'}


def cpu_baseline(args):
    """oracle (port of the reference path) fwd+bwd on the host cores, bounded sample"""
    from emsanet_amd import full_args, nyuv2_config
    from oracle.emsanet_oracle import EMSANetOracle, deterministic_state_dict, synthetic_batch
    cores = torch.get_num_threads()
    a = full_args(input_height=args.height, input_width=args.width,
                  rgb_encoder_backbone=args.backbone, depth_encoder_backbone=args.backbone,
                  rgb_encoder_backbone_resnet_block=args.resnet_block,
                  depth_encoder_backbone_resnet_block=args.resnet_block)
    o = EMSANetOracle(a, nyuv2_config())
    o.load_state_dict(deterministic_state_dict(o, 0))
    o.train()
    bs = 2      # train-mode BatchNorm of the PPM 1x1 branch needs > 1 sample per channel
    batch = synthetic_batch(bs, args.height, args.width)

    def step():
        for p in o.parameters():
            p.grad = None
        flat = flatten_outputs(o(batch))
        torch.autograd.backward(flat, [torch.full_like(t, 1e-3) for t in flat])
    # thread count: torch's default (all host cores) is the WORST choice for this workload on a
    # many-core box -- 128 threads on the ~300 small convolutions of a step take 14x longer than 16
    # (profiles/r05_oracle_threads.txt) -- so the baseline is timed with the fastest of a short
    # sweep, two steps each (the first one of the sweep also warms allocator / oneDNN caches)
    all_cores = os.cpu_count() or cores
    sweep = {}
    for nt in sorted({t for t in (8, 16, 32) if t <= all_cores} | ({all_cores} if all_cores <= 32 else set())):
        torch.set_num_threads(nt)
        step()
        t1 = time.time()
        step()
        sweep[nt] = time.time() - t1
    cores = min(sweep, key=sweep.get)
    torch.set_num_threads(cores)
    # 3 warm-up + 10 timed iterations; bounded so that the default run stays within
    # minutes on a slow host (warm-up: at most 30 s after the first iteration, timed: at least
    # 3 iterations, then until --cpu-seconds)
    t0 = time.time()
    n_warm = 0
    while n_warm < 3 and (n_warm == 0 or time.time() - t0 < 30.0):
        step()                  # warm-up (allocator, oneDNN primitive cache, thread pool)
        n_warm += 1
    warm = time.time() - t0
    times = []
    t0 = time.time()
    while len(times) < 10 and (len(times) < 3 or time.time() - t0 < args.cpu_seconds):
        t1 = time.time()
        step()
        times.append(time.time() - t1)
    dt = time.time() - t0
    n = len(times)
    limit = usable_cpus()
    saturating = cpu_replicas(args, cores, min(all_cores, limit), bs) if args.cpu_replicas else None
    return {'value': round(n * bs / dt, 4), 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'host_saturating': saturating, 'usable_cpus': limit,
            'best_iteration_value': round(bs / min(times), 4),
            'sample': f'{n} timed fwd+bwd iteration(s) of the PyTorch-CPU oracle, bs={bs}, '
                      f'{args.width}x{args.height} RGB-D, all heads, train mode, fp32 oneDNN, '
                      f'{cores} threads (fastest of a sweep over '
                      + ', '.join(f'{k}: {v:.2f} s/step' for k, v in sweep.items())
                      + f'; the host has {all_cores} cores), after {n_warm} warm-up iteration(s) ({warm:.1f}s)'}


def reference_protocol(args, model, graphed, bs, dev):
    """/root/reference/inference_time_whole_model.py:297-347 (`time_inference_pytorch`), run for run:
    start event; inputs host -> device; forward; every main output (side outputs ignored, eval mode
    has none) device -> host; end event; synchronise; t = elapsed.  Inputs are the reference's
    random frames (`:519-545`): --warmup + --steps DIFFERENT samples, fps = mean(1 / t) +- std
    (`:592`; `inference_time.bash:15-18` uses 20 + 80).  Two forms:
      as_reference   pageable float32 NCHW host tensors (normalised on the host, as the reference's
                     `.to(device)` sees them) and pageable `.cpu()` outputs -- the reference's loop
                     with this engine's forward in the middle;
      pinned_raw     what a deployment of this engine does: raw uint8 / uint16 frames in pinned host
                     memory (1/4 .. 1/2 of the bytes), normalisation as device kernels, outputs into
                     pinned host buffers;
      pinned_compact the same with what the reference's consumers actually READ crossing PCIe instead
                     of the raw maps (`inference_samples.py:128-136` / `visualization.py` take the
                     semantic class index + score and the instance maps): semantic arg-max (uint8) +
                     softmax score (fp16) formed on the device (`postprocessing.softmax_argmax`),
                     centre / offset / orientation as fp16, scene logits fp32 -- 3.7 MB per 640x480
                     frame instead of 55 MB.  An OPTION beside the raw-output number, not a
                     replacement: the protocol's letter is "every main output".
    The forward is the whole-model hipGraph when --graph is given, else the eager forward.
    --protocol-reps N: every form N times in this process; `fps_mean` etc. are the MEDIAN repetition
    (by fps_mean), `reps_fps_mean` lists them all."""
    from emsanet_amd.postprocessing import normalize_depth, normalize_rgb
    n_warm, n_runs = args.warmup, args.steps
    h, w = args.height, args.width
    rng = np.random.default_rng(4242)

    def main_outputs(out):
        flat = []
        for o, _ in out:
            for t in (o if isinstance(o, tuple) else (o,)):
                flat += list(t) if isinstance(t, tuple) else [t]
        return flat

    def forward(b):
        if graphed is not None:
            return graphed(b)
        with torch.no_grad():
            return model(b)

    def run(kind):
        frames = []
        for _ in range(n_warm + n_runs):
            rgb = rng.integers(0, 255, (bs, h, w, 3), dtype=np.uint8)
            depth = rng.integers(0, 40000, (bs, h, w), dtype=np.uint16)
            if kind == 'as_reference':
                frames.append({'rgb': torch.from_numpy(np.ascontiguousarray(
                    (rgb / 255).astype('float32').transpose(0, 3, 1, 2))),
                    'depth': torch.from_numpy((depth.astype('float32') / 20000)[:, None].copy())})
            else:
                frames.append({'rgb': torch.from_numpy(rgb).pin_memory(),
                               'depth': torch.from_numpy(depth).pin_memory()})
        host_out = None
        times, nbytes_in, nbytes_out = [], 0, 0
        for i, f in enumerate(frames):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if kind == 'as_reference':
                # (the graph's static input buffers are the destination of the host -> device copy)
                b = f if graphed is not None else {k: v.to(dev) for k, v in f.items()}
            else:
                b = {'rgb': normalize_rgb(f['rgb'].to(dev, non_blocking=True),
                                          mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0)),
                     'depth': normalize_depth(f['depth'].to(dev, non_blocking=True), 0.0, 20000.0,
                                              keep_invalid_zero=False)}
            outs = main_outputs(forward(b))
            if kind == 'pinned_compact':
                from emsanet_amd.postprocessing import softmax_argmax
                score, idx = softmax_argmax(outs[0])
                outs = [idx.to(torch.uint8), score.to(torch.float16)] + \
                    [t.to(torch.float16) if t.dim() == 4 else t for t in outs[1:]]
            if kind == 'as_reference':
                cpu = [t.cpu() for t in outs]
            else:
                if host_out is None:
                    host_out = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for t in outs]
                for d, t in zip(host_out, outs):
                    d.copy_(t, non_blocking=True)
                cpu = host_out
            e1.record()
            torch.cuda.synchronize()
            if i >= n_warm:
                times.append(e0.elapsed_time(e1) / 1e3)
            nbytes_in = sum(v.numel() * v.element_size() for v in f.values())
            nbytes_out = sum(t.numel() * t.element_size() for t in cpu)
        t = np.array(times)
        return {'fps_mean': round(float(np.mean(1 / t)) * bs, 2), 'fps_std': round(float(np.std(1 / t)) * bs, 2),
                'ms_mean': round(float(t.mean()) * 1e3, 4), 'ms_min': round(float(t.min()) * 1e3, 4),
                'host_to_device_bytes': nbytes_in, 'device_to_host_bytes': nbytes_out}
    reps = max(1, int(getattr(args, 'protocol_reps', 1)))

    def repeated(kind):
        rs = sorted((run(kind) for _ in range(reps)), key=lambda r: r['fps_mean'])
        med = dict(rs[len(rs) // 2])
        if reps > 1:
            med['reps_fps_mean'] = [r['fps_mean'] for r in rs]
            med['reps_ms_mean'] = [r['ms_mean'] for r in rs]
            med['quoted'] = f'median of {reps} repetitions in one process (by fps_mean)'
        return med
    res = {'protocol': 'ref inference_time_whole_model.py:297-347: per run {H2D inputs, forward, D2H main '
                       f'outputs}} inside one event pair, synchronised per run; {n_warm} warm-up + {n_runs} '
                       'timed runs on different random frames; fps = mean(1/t) +- std (x batch size)',
           'forward': 'whole-model hipGraph replay' if graphed is not None else 'eager forward',
           'as_reference': repeated('as_reference')}
    res['pinned_raw'] = repeated('pinned_raw')
    res['pinned_compact'] = repeated('pinned_compact')
    return res


def self_launch(args):
    """`python bench.py --gpus N` without a launcher around it: become the launcher -- one process
    per GPU through torch.distributed.run on 127.0.0.1 (the same command line the driver uses),
    stdout (rank 0's JSON line) passed through."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and os.environ.get('EMSA_DIST_BACKEND', 'nccl') == 'nccl':
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible (RCCL needs one "
                         "device per rank; EMSA_DIST_BACKEND=gloo rehearses the flow on one GPU)")
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.cpu_replica_worker >= 0:
        return cpu_replica_worker(args)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args))
    # stdout carries exactly ONE JSON line (rank 0): everything libraries print to fd 1 while the
    # benchmark runs (RCCL prints its version banner there) is routed to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    line = run(args)
    sys.stdout.flush()
    if line is not None:
        # fd 1 stays redirected (RCCL also prints at library teardown); the JSON line goes
        # straight to the original stdout
        os.write(real_stdout, (line + '\n').encode())


def run(args):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one process per GPU")
    assert torch.cuda.is_available(), "bench.py needs an AMD GPU"
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    if world > 1 or args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        # RCCL over xGMI; EMSA_DIST_BACKEND=gloo only exists to rehearse the multi-rank flow on a
        # one-GPU box (RCCL refuses two ranks on one device)
        dist.init_process_group(os.environ.get('EMSA_DIST_BACKEND', 'nccl'), rank=rank,
                                world_size=world)

    from emsanet_amd import _lib, full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from emsanet_amd.parallel import GradientBuckets, broadcast_parameters

    L = _lib.lib()
    a = full_args(input_height=args.height, input_width=args.width,
                  rgb_encoder_backbone=args.backbone, depth_encoder_backbone=args.backbone,
                  rgb_encoder_backbone_resnet_block=args.resnet_block,
                  depth_encoder_backbone_resnet_block=args.resnet_block,
                  compute_dtype={'f32': 'float32', 'bf16': 'bfloat16', 'f16': 'float16'}[args.dtype])
    torch.manual_seed(0)
    model = EMSANet(a, nyuv2_config())
    deterministic_init_(model)
    model.to(dev)
    broadcast_parameters(model)
    bs = args.batch_size
    batch = synthetic_batch_device(bs, args.height, args.width, 1234 + rank, dev)

    if args.eval:
        model.eval()
    else:
        model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    # FusedSGD folds the 1/world averaging into its update kernel; torch's optimizer needs it done
    if args.grad_dtype is None:
        args.grad_dtype = 'bf16' if args.dtype == 'bf16' else 'f32'
    comm_dtype = {'f32': None, 'bf16': torch.bfloat16}[args.grad_dtype]
    dist_on = dist.is_initialized() and (world > 1 or args.force_dist)
    # several ranks: the default is the segmented-hipGraph step (one graph per backward segment,
    # all-reduce issued eagerly in between) -- the hook-driven eager step costs 16 % in bf16 before a
    # byte crosses xGMI; --eager / --torch-optimizer / --h2d keep the eager path
    # one rank (round 5): the 16-bit step as ONE hipGraph by default as well -- bf16 955 vs 793-806
    # images/s, its eager step is host-bound; a refused capture falls back to the eager step below
    # (graph_auto).  Round 6 (last change of the round): the fp32 step too -- GPU-bound either way, but
    # the replay is consistently ahead: 325.8 / 325.9 / 325.6 vs 324.0 / 323.5 / 323.9 images/s
    # alternating on one box (profiles/r06_s_f32_{graph,eager}_{1,2,3}.json; round 5: 331.5 vs 329.9,
    # 319.0 vs 317.3 on two other boxes).  What the graph costs: the per-launch HIP events INSIDE the
    # timed region (`in_timed_region`); the roofline's own-rate pass runs behind the region either way.
    graph_auto = False
    if not args.eval and not args.eager and not args.torch_optimizer and not args.h2d and not args.graph:
        args.graph = True
        graph_auto = world == 1 and not args.force_dist
    segmented = bool(args.graph and not args.eval and dist_on)
    if segmented:
        # multi-rank step under hipGraphs: one graph per backward segment, the buckets of a segment
        # all-reduced eagerly while the next segment's graph runs (SegmentedGraphedTrainStep)
        if args.torch_optimizer:
            raise SystemExit("--graph with several ranks uses the fused optimizer")
        from emsanet_amd.graph import segment_parameter_groups
        # encoder cuts behind layer3, layer2 and layer1 (round 6: was (2, 1)) -- six backward segments;
        # layer4 + layer3 hold 45 % of the parameters and left as four buckets at one instant
        cuts = tuple(int(c) for c in args.cut_stages.split(',') if c != '')
        buckets = GradientBuckets(params, groups=segment_parameter_groups(model, cuts, decoder_cut=True),
                                  manual=True, force_collectives=args.force_dist, average=False,
                                  comm_dtype=comm_dtype, tail_bytes=4 << 20)
    elif dist_on and not args.eval:
        # bucket boundaries from the MEASURED gradient-arrival order of one backward pass (not
        # the registration order), the last bucket -- the first layers' gradients, which cannot
        # overlap anything -- at most 4 MiB
        from emsanet_amd.parallel import record_arrival_order

        def dry_backward():
            flat = flatten_outputs(model(batch))
            torch.autograd.backward(flat, [torch.zeros_like(t) for t in flat])
        order = record_arrival_order(params, dry_backward)
        for p in params:
            p.grad = None
        model.dropout_step = 0
        buckets = GradientBuckets(params, force_collectives=args.force_dist,
                                  average=args.torch_optimizer, comm_dtype=comm_dtype, order=order,
                                  tail_bytes=4 << 20)
    else:
        buckets = GradientBuckets(params, force_collectives=args.force_dist,
                                  average=args.torch_optimizer, comm_dtype=comm_dtype)
    # LR rule of the reference: 0.01 * batch/8 (args.py:1338-1344); tiny here so that the random
    # net stays finite over the benchmark steps
    if args.torch_optimizer:
        opt = torch.optim.SGD(params, lr=1e-5, momentum=0.9, weight_decay=1e-4, nesterov=True)
    else:
        # one fused kernel per flat bucket (emsa_sgd_nesterov), same update rule
        from emsanet_amd.optim import FusedSGD
        opt = FusedSGD(buckets, lr=1e-5, momentum=0.9, weight_decay=1e-4)
    cots = None
    crit, targets = None, None
    if args.losses and not args.eval:
        crit, targets = training_losses_and_targets(a, bs, args.height, args.width, dev)

    graphed = None
    if args.eval and args.graph:
        from emsanet_amd.graph import GraphedInference
        graphed = GraphedInference(model, batch)
    train_graph = None
    graph_fallback = None

    def step(fresh=None):
        # `fresh`: a batch that was just staged from host memory (--h2d); default: the HBM-resident one
        nonlocal cots
        b = batch if fresh is None else fresh
        if args.eval:
            if graphed is not None:
                graphed(b)
                return
            with torch.no_grad():
                model(b)
            return
        if train_graph is not None:
            train_graph.replay(fresh)
            return
        buckets.reset()
        if crit is not None:
            # --losses: the complete training step (all task losses on device, weighted like the
            # reference's training command) instead of fixed output cotangents
            total, _ = crit(model(b), targets)
            total.backward()
            buckets.finish()
            opt.step()
            return
        flat = flatten_outputs(model(b))
        if cots is None:
            g = torch.Generator(device='cpu').manual_seed(4321)
            cots = [(torch.randn(t.shape, generator=g) * 1e-3).to(dev).contiguous(
                memory_format=torch.channels_last if t.dim() == 4 else torch.contiguous_format)
                for t in flat]
        torch.autograd.backward(flat, cots)
        buckets.finish()
        opt.step()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if args.graph and not args.eval:
        # the whole training step (forward, backward, SGD) as ONE hipGraph replay; kernel timing by
        # HIP events is not available inside a graph
        from emsanet_amd.graph import GraphedTrainStep, SegmentedGraphedTrainStep
        flat0 = flatten_outputs(model(batch))
        g = torch.Generator(device='cpu').manual_seed(4321)
        cots = [(torch.randn(t.shape, generator=g) * 1e-3).to(dev).contiguous(
            memory_format=torch.channels_last if t.dim() == 4 else torch.contiguous_format)
            for t in flat0]
        del flat0
        kernel_timing_off = args.no_kernel_timing
        args.no_kernel_timing = True
        cls = SegmentedGraphedTrainStep if segmented else GraphedTrainStep
        # N > 1 (the default there is the segmented graph, which no multi-GPU node has run yet): a
        # capture that RCCL / the runtime refuses must not cost the scaling run -- the object then
        # replays its eager twin, the same complete step with the same collectives
        kw = {'eager_fallback': True} if (segmented and world > 1 and not args.force_dist) else {}
        if segmented:
            kw['decoder_cut'] = True       # the decoder segment's buckets leave in two steps (nn.CutPlan)
            kw['cut_stages'] = tuple(int(c) for c in args.cut_stages.split(',') if c != '')
        try:
            if crit is not None:
                train_graph = cls(model, batch, buckets, opt, loss_fn=lambda out: crit(out, targets)[0], **kw)
            else:
                train_graph = cls(model, batch, buckets, opt, cotangents=cots, **kw)
        except Exception as e:                  # noqa: BLE001
            if not graph_auto:
                raise
            # the graph was this script's own choice, not the caller's: run the eager step instead
            print(f'[bench] hipGraph capture of the training step failed ({type(e).__name__}: {e}); '
                  'eager step', file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            train_graph, args.graph, args.no_kernel_timing = None, False, kernel_timing_off
            graph_fallback = f'{type(e).__name__}: {e}'
        if train_graph is not None:
            graph_fallback = getattr(train_graph, 'capture_error', None)
        if graph_fallback:
            print(f'[bench] rank {rank}: hipGraph capture failed ({graph_fallback}); eager segmented step',
                  file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        step()
    buckets.reset_stats()
    timing = not args.no_kernel_timing
    L.emsa_prof_reset()
    # two streams: the roofline comes from the single-stream pass behind the timed region; inside
    # it the launches are only sampled sparsely (every 17th: coprime to the 4 launches of a block)
    # for the `in_timed_region` figures -- event pairs cost host time, and the eager 16-bit step
    # is host-bound
    from emsanet_amd import nn as enn_
    region_every = args.timing_every
    if timing and not args.graph and enn_._dual_stream(batch['rgb']):
        region_every = max(args.timing_every, 17)
    L.emsa_prof_enable(region_every if timing else 0)
    from emsanet_amd import functional as Fn
    Fn.PROF_REAL_FLOPS = timing      # padded / merged convs report their real FLOPs (stem 7x7x3 ...)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    L.emsa_prof_enable(0)
    Fn.PROF_REAL_FLOPS = False
    staged = None
    if args.h2d:
        # the same K steps once more with every batch coming from pinned HOST memory as raw uint8 /
        # uint16 frames: copied on a separate stream while the previous step runs, normalised on
        # the device (emsanet_amd/staging.py).  Reported beside `value`, never as `value`.
        import itertools
        from emsanet_amd.staging import BatchStager, pinned_raw_batch
        raw = [pinned_raw_batch(bs, args.height, args.width, seed=77 + i) for i in range(3)]
        stager = BatchStager(itertools.cycle(raw), dev, depth_stats=(2841.94, 1417.26))
        feed = iter(stager)
        for _ in range(2):
            step(next(feed))
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(next(feed))
        barrier()
        dt_h = time.perf_counter() - t1
        per = sum(v.numel() * v.element_size() for v in raw[0].values())
        staged = {'value': round(bs * world * args.steps / dt_h, 2), 'unit': 'images/s',
                  'ms_per_step': round(1e3 * dt_h / args.steps, 2),
                  'host_bytes_per_step': per,
                  'input': 'raw uint8 RGB + uint16 depth frames in pinned host memory, H2D on a copy '
                           'stream overlapped with the previous step, NormalizeRGB/NormalizeDepth + '
                           'HWC->CHW as device kernels'}
    ref_protocol = None
    if args.eval and args.protocol == 'reference' and world == 1:
        ref_protocol = reference_protocol(args, model, graphed, bs, dev)
    comm = None
    if dist.is_initialized():
        # evidence that the ranks really ran and exchanged: every rank reports (rank, device index,
        # its own wall time, the time its compute stream waited for un-hidden collectives)
        exposed = buckets.exposed_comm_ms()
        mine = torch.tensor([float(rank), float(local_rank), dt, -1.0 if exposed is None else exposed],
                            device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows = [r.tolist() for r in allr]
        st = buckets.stats
        steps_seen = max(1, st['steps'])
        comm = {'backend': dist.get_backend(), 'ranks_seen': sorted(int(r[0]) for r in rows),
                'devices': [int(r[1]) for r in rows],
                'per_rank_ms_per_step': [round(1e3 * r[2] / args.steps, 2) for r in rows],
                'exposed_comm_ms_per_step': [None if r[3] < 0 else round(r[3], 3) for r in rows],
                'allreduce_bytes_per_step': st['bytes'] // steps_seen,
                'collectives_per_step': st['collectives'] // steps_seen,
                'bucket_dtype': args.grad_dtype,
                'rccl_env': __import__('emsanet_amd.parallel', fromlist=['rccl_env']).rccl_env(),
                'conv_rs_cu_budget': getattr(buckets, 'rs_cu_budget', None),
                'grads_written_in_place': st['direct_tensors'] // steps_seen,
                'grads_gathered_by_copy': st['gathered_tensors'] // steps_seen,
                'bucket_bytes': [f.numel() * f.element_size() for f, _, _ in buckets.buckets],
                'path': ('segmented-graph' if not graph_fallback else
                         f'segmented-eager (graph capture failed: {graph_fallback})') if segmented else 'eager',
                'bucket_order': 'backward segments (graph per segment; the decoder segment cut in two)' if segmented
                else 'measured gradient-arrival order, last bucket <= 4 MiB'}
        if segmented and train_graph is not None:
            # per bucket: issued this long before the device finished the backward pass
            comm['bucket_launch_ms_before_backward_end'] = \
                train_graph.bucket_launch_ms_before_backward_end()
            comm['graphs'] = [i['nodes'] for i in train_graph.graph_info]
        dt = max(r[2] for r in rows)

    images = bs * world * args.steps
    value = images / dt
    # ---- roofline of the dominant kernel --------------------------------------------------
    def read_kernels():
        kernels = []
        for cls in range(32):
            if not L.emsa_prof_name(cls):
                break
            ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int32()
            _lib.check(L.emsa_prof_read(cls, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)),
                       'emsa_prof_read')
            if n.value:
                nbytes = ctypes.c_double()
                _lib.check(L.emsa_prof_read_bytes(cls, ctypes.byref(nbytes)), 'emsa_prof_read_bytes')
                seen = L.emsa_prof_seen(cls)
                scale = seen / n.value            # sampled -> all launches of the class
                kernels.append({'kernel': L.emsa_prof_name(cls).decode(), 'launches': seen,
                                'timed_launches': n.value,
                                'total_ms': round(ms.value * scale, 3),
                                'avg_us': round(1e3 * ms.value / n.value, 2),
                                'algo_gflop_per_launch': round(fl.value / n.value / 1e9, 4),
                                'tflops': round(fl.value / ms.value / 1e9, 2)})
                if nbytes.value > 0:
                    kernels[-1]['algo_mb_per_launch'] = round(nbytes.value / n.value / 1e6, 2)
                    kernels[-1]['algo_gbps'] = round(nbytes.value / ms.value / 1e6, 1)
                if 'wino' in kernels[-1]['kernel']:
                    # Winograd F(2,3): 4 MFMA multiplies per 6 direct-convolution multiplies;
                    # 'tflops' is algorithmic (direct-conv FLOPs / time), this is what the
                    # matrix pipe executed
                    kernels[-1]['mfma_executed_tflops'] = round(kernels[-1]['tflops'] * 4 / 6, 2)
        kernels.sort(key=lambda k: -k['total_ms'])
        return kernels

    kernels = read_kernels() if timing else []
    # whole-step roofline position: the algorithmic bytes of ONE step -- every
    # tensor operand of every launch counted once per launch (emsanet_amd.functional._p), summed over
    # an eager pass of the same step (a graph replay issues the same launches) -- against the step
    # time of the timed region and the 8 TB/s HBM roofline, next to the step's FLOPs against the
    # MFMA peak of the arithmetic
    whole_step = None
    if world == 1 and not args.h2d:
        cnt = [0]
        Fn.BYTES = cnt
        try:
            if train_graph is not None:
                (train_graph.eager_step(None) if hasattr(train_graph, 'eager_step') else train_graph._step())
            elif args.eval:
                with torch.no_grad():
                    model(batch)
            else:
                step()
        finally:
            Fn.BYTES = None
        torch.cuda.synchronize()
        step_s = dt / args.steps
        whole_step = {'algo_gb_per_step': round(cnt[0] / 1e9, 3),
                      'hbm_frac': round(cnt[0] / step_s / 8e12, 4),
                      'note': 'sum over all launches of one step of the bytes of every tensor operand, '
                              'each once per launch (weights, workspaces and statistics rows included; '
                              'the table-driven weight-pack launch excluded) / ms_per_step / 8 TB/s'}
    # The timed region runs the two independent halves of the network on TWO HIP streams (rgb |
    # depth encoder stage, semantic | instance decoder): launches overlap there, and the duration of
    # an overlapped launch is not its own -- two kernels share the CUs.  The kernel's own rate is
    # therefore measured live in `roofline_steps` more steps of the same workload with the second
    # stream switched off (same process, same batch, same sampling), right behind the timed
    # region; what the events saw INSIDE the timed region is reported next to it.
    in_region = None
    from emsanet_amd import nn as enn
    overlapped = timing and not args.graph and enn._dual_stream(batch['rgb'])
    single_pass = overlapped          # (every rank takes part: the steps contain the collectives)
    # a graph-replayed training step cannot be bracketed by HIP events per launch: its kernels are
    # timed in `roofline_steps` EAGER steps of the same step object (same kernels, same buffers, the
    # collectives included) behind the timed region, on one stream
    graph_roof = train_graph is not None and not args.eval and args.roofline_steps > 0

    def eager_twin():
        if hasattr(train_graph, 'eager_step'):
            train_graph.eager_step(None)
        else:
            train_graph._step()
    if overlapped and kernels:
        k0 = kernels[0]
        in_region = {'kernel': k0['kernel'], 'avg_us': k0['avg_us'], 'achieved': k0['tflops'],
                     'launches': k0['launches'],
                     'sum_of_launch_durations_over_step_time': round(
                         sum(k['total_ms'] for k in kernels) / (dt * 1e3), 4),
                     'note': 'launches of the two streams overlap: durations include CU sharing'}
        if args.dtype != 'f32' and 'algo_gbps' in k0:
            in_region['achieved'] = k0['algo_gbps']
    if (single_pass and kernels) or graph_roof:
        roof_step = eager_twin if graph_roof else step
        enn.DUAL_STREAM = False
        try:
            for _ in range(2):
                roof_step()
            L.emsa_prof_reset()
            L.emsa_prof_enable(args.timing_every)
            Fn.PROF_REAL_FLOPS = True
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.roofline_steps):
                roof_step()
            barrier()
            dt_r = time.perf_counter() - t1
            L.emsa_prof_enable(0)
            Fn.PROF_REAL_FLOPS = False
        finally:
            enn.DUAL_STREAM = None
        kernels = read_kernels()
        dt_roof, steps_roof = dt_r, args.roofline_steps
    else:
        dt_roof, steps_roof = dt, args.steps
    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return None
    roofline = None
    traffic, traffic_src = None, None
    pmc, pmc_file = None, None
    pmc_names = ('r06_pmc_traffic.json', 'r05_pmc_traffic.json') \
        if args.dtype == 'f32' else (f'r06_pmc_traffic_{args.dtype}.json', f'r05_pmc_traffic_{args.dtype}.json')
    pmc_stale = None
    for name in pmc_names:                                             # newest measurement first
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', name)))
            pmc_file = 'profiles/' + name
        except Exception:
            continue
        # a traffic file describes the kernels it was measured on: refuse it when the kernel
        # sources changed since (content hash of emsanet_amd/csrc -- the GPU box has no .git)
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        from pmc_traffic_json import csrc_sha16
        have = csrc_sha16(ROOT)
        if pmc.get('csrc_sha16') != have:
            pmc_stale = (f"{pmc_file} was measured on other kernel sources (csrc_sha16 "
                         f"{pmc.get('csrc_sha16', 'absent: a round-3 file')} != {have}): traffic withheld; "
                         "re-run tools/pmc_traffic2.sh")
            pmc = None
            continue
        break
    if kernels:
        k = kernels[0]
        if pmc and (args.height, args.width, bs) == (480, 640, 32) and not args.eval:
            name = k['kernel'].replace(' ', '').split('(')[0]
            ents = [v for kk, v in pmc['kernels'].items() if kk == name or kk.startswith(name + '<')]
            ent = None
            if ents:                   # all template instances of the class, weighted by launches
                n_l = sum(v['launches_profiled'] for v in ents)
                ent = {'hbm_bytes_per_launch': int(sum(v['hbm_bytes_per_launch'] * v['launches_profiled']
                                                       for v in ents) / n_l)}
            if ent:
                traffic = ent['hbm_bytes_per_launch']
                cal = pmc.get('calibration', {})
                traffic_src = f"{pmc_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this " \
                              f"command at commit {pmc.get('commit')}, kernel sources {pmc.get('csrc_sha16')}" \
                              f" = these; FETCH x {cal.get('fetch_factor')}, WRITE x {cal.get('write_factor')} " \
                              "from the 1 GiB read / write / copy calibration kernels of the same passes; " \
                              "the PMC passes cannot run inside the timed bench)"
        if traffic is None and pmc_stale:
            traffic_src = pmc_stale
        roofline = {'bound': 'mfma', 'kernel': k['kernel'], 'achieved': k['tflops'],
                    'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(k['tflops'] / MFMA_F32_PEAK_TFLOPS, 4), 'traffic': traffic,
                    'traffic_unit': 'HBM bytes per launch (PMC)', 'traffic_source': traffic_src,
                    'launches': k['launches'], 'avg_us': k['avg_us'],
                    'algo_gflop_per_launch': k['algo_gflop_per_launch'],
                    'share_of_step': round(k['total_ms'] / (dt_roof * 1e3), 4)}
        if args.dtype != 'f32' and 'algo_gbps' in k:
            # 16-bit operands: at 2.5 PFLOP/s the convolutions are bounded by bytes, not by the
            # matrix pipe: achieved = algorithmic bytes per launch / average duration
            roofline.update({'bound': 'hbm', 'achieved': k['algo_gbps'], 'peak': HBM_PEAK_GBPS,
                             'unit': 'GB/s', 'frac': round(k['algo_gbps'] / HBM_PEAK_GBPS, 4),
                             'algo_mb_per_launch': k['algo_mb_per_launch'],
                             'mfma_tflops': k['tflops'],
                             'mfma_frac_of_bf16_peak': round(k['tflops'] / MFMA_BF16_PEAK_TFLOPS, 4)})
        if in_region is not None and not single_pass:
            roofline['measured_over'] = ('the timed region, where launches of two streams overlap '
                                         '(durations include CU sharing; the single-stream pass '
                                         'needs a single process)')
        if in_region is not None and single_pass:
            roofline['measured_over'] = (
                f'{steps_roof} single-stream steps of the same workload behind the timed region '
                f'({round(1e3 * dt_roof / steps_roof, 2)} ms/step; every {args.timing_every}th launch '
                'bracketed by HIP events on its stream): the kernel\'s own rate')
            in_region['frac'] = round(in_region['achieved'] / roofline['peak'], 4)
            roofline['in_timed_region'] = in_region
        if graph_roof:
            roofline['measured_over'] = (
                f'{steps_roof} EAGER single-stream steps of the graphed step object behind the timed '
                f'region ({round(1e3 * dt_roof / steps_roof, 2)} ms/step; every {args.timing_every}th '
                'launch bracketed by HIP events): a graph replay cannot be bracketed per launch')
        if 'mfma_executed_tflops' in k:
            roofline['mfma_executed_tflops'] = k['mfma_executed_tflops']
            roofline['frac_mfma_executed'] = round(k['mfma_executed_tflops'] / MFMA_F32_PEAK_TFLOPS, 4)
            roofline['note'] = ('Winograd F(2,3) kernel: achieved = direct-convolution FLOPs / '
                                'time (SURVEY 8d algorithmic figure); the MFMA pipe executes 2/3 '
                                'of them (frac_mfma_executed)')
    # the 1-D NBt1D kernel and the dense 3x3 kernel reported apart (same kernels,
    # separate launch classes), forward / data gradient and weight gradient, each with the
    # direct-convolution fraction and the fraction the matrix pipe really executed
    by_class = None
    if kernels and args.dtype == 'f32':
        labels = {'conv1d_wino_kernel (1-D': 'nbt1d_1d_fwd_dgrad',
                  'conv1d_wino_kernel (dense': 'dense3x3_fwd_dgrad',
                  'conv_wgrad1d_wino_kernel<64,64> (1-D': 'nbt1d_1d_wgrad',
                  'conv_wgrad1d_wino_kernel<64,64> (dense': 'dense3x3_wgrad'}
        by_class = {}
        for kk in kernels:
            for frag, label in labels.items():
                if frag in kk['kernel']:
                    ex = kk.get('mfma_executed_tflops', kk['tflops'])
                    by_class[label] = {
                        'kernel': kk['kernel'], 'launches': kk['launches'], 'avg_us': kk['avg_us'],
                        'share_of_step': round(kk['total_ms'] / (dt_roof * 1e3), 4),
                        'achieved': kk['tflops'], 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                        'frac': round(kk['tflops'] / MFMA_F32_PEAK_TFLOPS, 4),
                        'mfma_executed_tflops': ex,
                        'frac_mfma_executed': round(ex / MFMA_F32_PEAK_TFLOPS, 4)}
    conv_ms = sum(k['total_ms'] for k in kernels)
    conv_fl = sum(k['total_ms'] * k['tflops'] for k in kernels)     # ms * TFLOP/s = GFLOP
    step_gflop = (1 if args.eval else 3) * FWD_GFLOP_PER_IMAGE * bs \
        if (args.height, args.width, args.backbone, args.resnet_block) == (480, 640, 'resnet34', 'nonbottleneck1d') \
        else None

    blk = {'nonbottleneck1d': 'NBt1D', 'basicblock': 'basic', 'bottleneck': 'bottleneck'}[args.resnet_block]
    head = (f'full EMSANet RGB-D ({args.backbone}-{blk} x2, SE-add fusion, PPM, '
            f'semantic+instance+orientation+scene heads), {args.width}x{args.height}, bs={bs}/GPU, '
            f'{args.dtype}')
    # which BASELINE.json config this run is (or that it is none of them)
    at_640 = (args.height, args.width) == (480, 640)
    if args.resnet_block != 'nonbottleneck1d':
        cfg = 'not a BASELINE.json config (other encoder block)'
    elif args.backbone == 'resnet101':
        cfg = ('BASELINE.json configs[3] (ResNet-101-NBt1D dual encoder, 960x720 rounded up to the '
               'next multiple of 32 rows, bs=16/GPU)' if (args.width, bs) == (960, 16) and args.height in (720, 736)
               else 'ResNet-101-NBt1D variant (not a BASELINE.json config)')
    elif args.eval and bs == 1 and at_640:
        cfg = 'BASELINE.json configs[4] (inference, bs=1' + (', fp16' if args.dtype == 'f16' else
                                                             f'; {args.dtype} instead of fp16') + ')'
    elif at_640 and bs == 32 and args.backbone == 'resnet34':
        cfg = f'BASELINE.json configs[{1 if args.dtype == "f32" else 2}]' + (' shape, inference' if args.eval else '')
    else:
        cfg = 'not a BASELINE.json config (other size / batch / backbone)'
    if args.eval:
        workload = (f'{cfg}: {head}, eval mode (BatchNorm folded into the convolutions), '
                    'step = one forward pass' + (' replayed from a hipGraph' if args.graph else ''))
    else:
        workload = (f'{cfg}: {head}, train mode '
                    '(BN batch stats, Dropout2d), step = fwd + bwd ('
                    + ('all task losses on device' if args.losses else 'fixed output cotangents')
                    + ') + grad all-reduce + SGD-nesterov update'
                    + (' replayed from a hipGraph' if args.graph else ''))
    if roofline is None and whole_step is not None and (args.eval or args.graph):
        # no per-launch brackets (graph replay) or no dominant MFMA class (inference): the roofline
        # position of the WHOLE step -- algorithmic bytes of all its launches / step time / 8 TB/s
        gbps = whole_step['algo_gb_per_step'] / (dt / args.steps)
        roofline = {'bound': 'hbm', 'kernel': 'whole step (every launch of one '
                    + ('forward pass' if args.eval else 'training step') + ')',
                    'achieved': round(gbps, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                    'frac': round(gbps / HBM_PEAK_GBPS, 4), 'traffic': None,
                    'algo_mb_per_step': round(whole_step['algo_gb_per_step'] * 1e3, 2),
                    'launches': (graphed.graph_info['nodes'] if graphed is not None else None),
                    'measured_over': 'the timed region (wall clock between two synchronisations)'}

    out = {
        'metric': f'images/sec ({args.width}x{args.height} RGB-D, bs={bs}/GPU) '
                  + ('forward (inference)' if args.eval else 'fwd+bwd'),
        'value': round(value, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(1e3 * dt / args.steps, 2),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        # EMSA_BF16_MFMA=1 is an opt-in mixed-precision mode (BASELINE config 3), never the headline
        'dtype': {'f32': 'bf16-mfma/f32-accumulate+storage'
                  if os.environ.get('EMSA_BF16_MFMA') == '1' else 'f32',
                  'bf16': 'bf16 (activations + MFMA operands; fp32 accumulate, master weights, '
                          'BatchNorm statistics, outputs)',
                  'f16': 'f16 (activations + MFMA operands; fp32 accumulate, inference only)'}[args.dtype],
        'data': 'synthetic',
        'config': {'workload': workload,
                   'global_batch': bs * world, 'parallelism': f'dp{world}',
                   'weights': 'random init (deterministic)',
                   'mode': ('eval-fwd-hipgraph' if args.graph else 'eval-fwd') if args.eval
                   else ('train+losses' if args.losses else 'train')
                   + ('-hipgraph' if args.graph and not graph_fallback else '')
                   + ('-eager-after-failed-capture' if graph_fallback else '')},
        'roofline': roofline,
        'whole_step': whole_step,
        'roofline_by_class': by_class,
        'conv_kernels': kernels,
        'conv_mfma_time_share': round(conv_ms / (dt_roof * 1e3), 4) if kernels else None,
        'conv_mfma_tflops_overall': round(conv_fl / conv_ms, 2) if conv_ms else None,
        'model_tflops_effective': round(step_gflop * world * args.steps / dt / 1e3, 2)
        if step_gflop else None,
        'peak_hbm_gib': round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
        'comm': comm,
        # nodes of the captured hipGraph(s) replayed per step (memset nodes replaced by fill kernels)
        'hipgraph': ({'graphs': 1, 'nodes': [graphed.graph_info['nodes']],
                      'memset_nodes_replaced_by_kernels': graphed.graph_info['replaced']} if graphed is not None
                     else {'graphs': len(getattr(train_graph, 'graph_info', None) or [1]),
                           'nodes': [i['nodes'] for i in train_graph.graph_info]
                           if isinstance(train_graph.graph_info, list)
                           else [train_graph.graph_info['nodes']]} if train_graph is not None else None),
        'input': 'resident in HBM before the timed region',
        'h2d_staged': staged,
        'reference_protocol': ref_protocol,
    }
    if not args.no_cpu_baseline and world == 1:
        out['cpu_baseline'] = cpu_baseline(args)
    else:
        out['cpu_baseline'] = None
    if dist.is_initialized():
        dist.destroy_process_group()
    return json.dumps(out)


if __name__ == '__main__':
    main()
