"""16-bit engine (BASELINE configs[2]: bf16 mixed-precision training; configs[4]: 16-bit whole-model
hipGraph inference) against the fp32 / fp64 CPU oracle on identical inputs and weights.

Stated tolerances (activations are STORED with an 8-bit (bf16) / 11-bit (fp16) mantissa between
~100 layers; all accumulation, BatchNorm statistics, parameters and outputs are fp32):
  * eval outputs: relative L2 error of every raw output vs the fp32 oracle  <= 3e-2 (bf16),
    <= 4e-3 (fp16); semantic / scene class maps: arg-max identical except where the oracle's top-2
    margin is below 4 x the measured max error (no disagreement at a non-tie), and on >= 97.5 %
    (bf16) / 99.7 % (fp16) of the pixels of a RANDOM-WEIGHT network, whose class margins are tiny
    (measured 98.3 % / 99.8 %);
  * tight gates, per module type, against the fp64 oracle in STORAGE-EMULATION mode (it rounds
    where the engine rounds: oracle Spec.STORAGE) on the engine's ReLU branch: NBt1D block and
    decoder module forward <= 1e-2, input / parameter gradients <= 3e-2 relative L2 with gradient
    norms within 1 % (measured: forward 3e-4 .. 5e-3, gradients 2e-3 .. 1.4e-2, norm ratios
    0.999 .. 1.003); eval forward of the whole model <= 2e-2 (bf16) / 3e-3 (fp16);
  * whole-model TRAIN step (bf16, BatchNorm batch statistics): rounding is chaotic at the ulp
    level -- sub-ulp differences become 1-ulp differences at the next stored tensor -- and every
    batch-statistics BatchNorm renormalises signal and storage noise alike, so the outputs of this
    ~100-layer random-weight network differ by 5-25 % relative L2 from exact arithmetic whichever
    16-bit implementation computes them (tools/stagewise_dtype.py / stagewise_emul.py: +0.3-0.5 %
    per block, nothing sudden).  Gate: outputs <= 0.5, direction of the gradient: whole-gradient
    cosine >= 0.95 (measured 0.9945), per-tensor median >= 0.95 (measured 0.995, min 0.93).
The fp32 engine keeps north_star's 1e-3 (tests/test_model_gpu.py).
"""
import pytest
import torch

from util import DEV, rnd

pytestmark = pytest.mark.gpu

OUT_TOL = {torch.bfloat16: 3e-2, torch.float16: 4e-3}
AGREE = {torch.bfloat16: 0.975, torch.float16: 0.997}
# against the storage-emulating oracle (rounding points reproduced): provisional, see the measured values
EMU_TOL = {torch.bfloat16: 2e-2, torch.float16: 3e-3}
# train-mode model level (BatchNorm batch statistics renormalise the storage noise at every layer):
# provisional gates, see the measured values printed by the test
TRAIN_OUT_TOL, TRAIN_COS = 0.5, 0.95
# frozen-BatchNorm (eval) gradients at 640x480, measured (three runs, two boxes): outputs rel-L2 1.0e-2 ..
# 6.2e-2; norm ratio median 0.9992 (encoder tensors 0.9981), p1 .. p99 0.986 .. 1.015; cosine median 0.9999,
# p1 0.9985, min 0.9856.  Ten of the 742 tensors leave the +-5 % band: SE fc.0 of the first two fusions
# (0.949 / 1.038), and the instance head's centre / offset task convs and up-sampling weights (0.95 .. 1.12;
# the one-element centre bias 1.54 at cosine +1) -- the gradient through sigmoid / tanh scales like
# exp(-|x|), so the 4-6 % forward deviation of these outputs becomes 10 % in their derivative; the
# orientation conv beside them (no saturating activation) sits at 0.9975 / 1.0000.
EVAL_GRAD_OUT_TOL, EVAL_GRAD_COS_MEDIAN, EVAL_GRAD_COS_MIN = 0.1, 0.999, 0.95
# per-tensor norm-ratio bands of that test for the tensors whose scale the train-mode test cannot gate.
# Measured (r06i, gpurun_out/grad_ratio_bf16_evalbn_480x640.txt): squeeze-excite linears 0.982 .. 1.037,
# tensors of < 1024 elements 0.967 .. 1.053, the one-element centre bias 1.124 (round 5: 1.54 -- it sums
# the derivative of a partly saturated sigmoid over all pixels); all 742 tensors inside 0.967 .. 1.124.
SE_RATIO, SMALL_RATIO, SMALL_RATIO_CENTRE = (0.93, 1.07), (0.9, 1.12), (0.6, 1.7)


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


@pytest.mark.parametrize('variant', ['rgbd', 'basicblock', 'bottleneck', 'normal', 'up_nearest', 'up_bilinear'])
def test_option_space_bf16(variant):
    """the model options beyond BASELINE's configs in bf16 storage: eval forward vs the fp32 oracle
    at the eval tolerance, and a train step (forward + backward) with finite gradients everywhere"""
    from emsanet_amd import full_args
    from oracle.emsanet_oracle import synthetic_batch
    kw = dict(input_height=96, input_width=128)
    if variant == 'rgbd':
        kw.update(input_modalities=('rgbd',), semantic_encoder_decoder_fusion='add-rgbd',
                  instance_encoder_decoder_fusion='add-rgbd')
    elif variant == 'basicblock':
        kw.update(rgb_encoder_backbone='resnet18', depth_encoder_backbone='resnet18',
                  rgb_encoder_backbone_resnet_block='basicblock',
                  depth_encoder_backbone_resnet_block='basicblock')
    elif variant == 'bottleneck':
        kw.update(rgb_encoder_backbone='resnet50', depth_encoder_backbone='resnet50',
                  rgb_encoder_backbone_resnet_block='bottleneck',
                  depth_encoder_backbone_resnet_block='bottleneck')
    elif variant.startswith('up_'):
        # weight-free decoder / prediction up-sampling (ref args.py:280-298,363-372,439-448)
        mode = variant[3:]
        kw.update(semantic_decoder_upsampling=mode, instance_decoder_upsampling=mode,
                  upsampling_prediction=mode)
    else:
        kw.update(tasks=('semantic', 'instance', 'orientation', 'scene', 'normal'))
    model, oracle = _pair(full_args(**kw))
    model.set_compute_dtype(torch.bfloat16)
    batch = synthetic_batch(4, 96, 128)
    dev_batch = {k: v.to(DEV) for k, v in batch.items()}
    model.eval(), oracle.eval()
    with torch.no_grad():
        ref = _flatten(oracle(batch))
        out = _flatten(model(dev_batch))
    for i, (a, b) in enumerate(zip(out, ref)):
        e = _rel_l2(a, b)
        # (ResNet-50 bottleneck: 50 instead of 34 rounded layers per encoder, measured 3.1e-2)
        tol = OUT_TOL[torch.bfloat16] * (2 if variant == 'bottleneck' else 1)
        assert a.dtype == torch.float32 and e <= tol, f"{variant} output {i}: {e:.3e}"
    model.train()
    outs = _flatten(model(dev_batch))
    sum((t * t).mean() for t in outs).backward()
    for k, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k


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


@pytest.mark.parametrize('shape', [(8, 256, 320), (2, 480, 640), (8, 480, 640)])
def test_train_bf16_pinned_gradients(shape, monkeypatch):
    """configs[2] arithmetic on one rank: bf16 train step (BatchNorm batch statistics, Dropout2d),
    fwd + bwd at 256x320 with bs 8 and AT THE BASELINE RESOLUTION 640x480 with bs 2 and bs 8 (the kernels'
    tile / K-step / persistent-grid choices depend on the shape: conv_rs takes the /4../32 stages
    there like in the bs 32 bench), against the fp64 oracle that (1) replays the engine's
    ReLU decisions and (2) rounds activations, their gradients and the conv weights to bf16 where
    the engine stores them.  (Against the PLAIN fp64 oracle the train-mode outputs of this
    random-weight network differ by 0.2-0.4 relative L2: every BatchNorm with batch statistics
    renormalises signal AND bf16 storage noise, ~0.5 % per block over ~100 layers -- measured with
    tools/stagewise_dtype.py; that is a property of 8-bit mantissas, not of the kernels.)"""
    import torch.nn.functional as F
    from emsanet_amd import full_args, ops
    from oracle import emsanet_oracle as O
    from test_model_gpu import _PinnedRelu
    bs, hh, ww = shape
    args = full_args(input_height=hh, input_width=ww)
    model, oracle = _pair(args)
    oracle = oracle.double()
    model.set_compute_dtype(torch.bfloat16)
    for m in (model, oracle):
        m.train()
        m.dropout_seed = 321
    batch = O.synthetic_batch(bs, hh, ww)
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
    cos = torch.tensor([torch.nn.functional.cosine_similarity(
        dict(model.named_parameters())[k].grad.detach().cpu().double().flatten(),
        pr[k].grad.flatten(), dim=0).item() for k in names])
    g_all = torch.cat([dict(model.named_parameters())[k].grad.detach().cpu().double().flatten()
                       for k in names])
    r_all = torch.cat([pr[k].grad.flatten() for k in names])
    cos_all = torch.nn.functional.cosine_similarity(g_all, r_all, dim=0).item()
    mp = dict(model.named_parameters())
    ratio = torch.tensor([mp[k].grad.norm().item() / pr[k].grad.norm().item() for k in names])
    import os
    if os.path.isdir('gpurun_out'):
        with open(f'gpurun_out/grad_ratio_bf16_{hh}x{ww}.txt', 'w') as f:
            for k, r_, c_ in zip(names, ratio.tolist(), cos.tolist()):
                f.write(f"{r_:.4f} {c_:.4f} {k}\n")
    print("gradient norm ratio engine/oracle: median %.4f p5 %.4f p95 %.4f; first layers %s; last %s" % (
        ratio.median(), ratio.quantile(0.05), ratio.quantile(0.95),
        ' '.join(f'{names[i].split(".")[-3][:8]}.{names[i].split(".")[-2][:9]}={ratio[i]:.3f}' for i in range(0, 12, 2)),
        ' '.join(f'{ratio[i]:.3f}' for i in range(len(names) - 6, len(names)))))
    print(f"bf16 train: {len(errs)} gradients vs emulating oracle: rel-L2 median {e.median():.2e} p95 "
          f"{e.quantile(0.95):.2e}; cosine per tensor median {cos.median():.4f} p5 "
          f"{cos.quantile(0.05):.4f} min {cos.min():.4f}; whole-gradient cosine {cos_all:.4f}; "
          f"{pinned.flips} of {pinned.total} ReLU decisions differ from the oracle's own")
    assert max(eo) <= TRAIN_OUT_TOL, eo
    if bs < 8:
        # 640x480 with TWO images: the /32 BatchNorms normalise over 600 samples per channel and
        # renormalise every bf16 rounding-boundary flip in front of them (2.6 % of the ReLU decisions
        # differ from the oracle's own here, 0.3 % at bs 8) -- the noise floor of this configuration,
        # not of a kernel: measured whole-gradient cosine 0.874 with conv_rs.hip, 0.759 with every
        # conv on the implicit GEMM (EMSA_CONV_RS=0), same box, same inputs (profiles/
        # r04_c_bf16_pinned_640x480_bs2.txt).  What this case pins is the launch set of the BASELINE
        # resolution (conv_rs plans per stage, 3x1 patch tiling at 120x160 ... 15x20): a wrong
        # kernel class shows as cosine << 0.5 on its tensors.
        # (four builds of round 4 on four boxes: whole-gradient cosine 0.874, 0.790, 0.790, and 0.759
        #  with conv_rs switched off; median gradient-norm ratio 1.11 ... 1.58 -- the draw moves with
        #  every change of a reduction's partition, e.g. the statistics rows.  With two images the
        #  train-mode network is a chaotic map of its bf16 rounding noise; the gates below only catch
        #  structural errors -- wrong kernel, wrong operand, sign -- which give cosines near 0 or -1)
        assert cos_all >= 0.5 and cos.median().item() >= 0.6, (cos_all, cos.median().item())
        assert (cos >= 0.3).float().mean().item() >= 0.9
        return
    assert cos_all >= TRAIN_COS and cos.median().item() >= TRAIN_COS
    # PER-TENSOR gates (the whole-gradient cosine alone would pass a sign
    # error in one small tensor).  Measured over the 666 gradient tensors of this configuration:
    # cosine min 0.934 (p1 0.949), norm ratio p1..p99 0.96..1.12, extremes 0.68 / 1.27 on eight
    # tiny tensors (side-head biases, SE fc.0 of the first fusion, the 9-tap upsampling weights).
    # A sign error gives cosine -1 (every tensor clears the cosine bound); a dropped term or factor of
    # two gives a ratio of 0.5 / 2, which the ratio band below catches on the tensors of >= 1024
    # elements outside the SE MLPs (97 % of all tensors inside the tight band) -- for the loose class see
    # the note at the ratio gate.
    # (tensors of < 1024 elements and the squeeze-excite linears form the loose class, as for the norm
    #  ratio below: 0.862 on decoders.instance_decoder.side_output_heads.2.task_convs.1.bias -- two
    #  elements, each a nearly cancelling sum over all pixels -- on one build of round 5; >= 0.7 there
    #  still separates a sign error, cosine -1)
    big = torch.tensor([mp[k].numel() >= 1024 and '.se_' not in k for k in names])
    for sel, cmin in ((big, 0.85), (~big, 0.7)):          # (big: measured minima 0.934 ... 0.944 over five builds)
        c_ = cos[sel]
        nm = [k for k, b_ in zip(names, sel.tolist()) if b_]
        assert c_.min().item() >= cmin, (nm[int(c_.argmin())], c_.min().item())
    # The norm ratio is NOT centred on 1: it climbs from 1.00 at the heads through every train-mode
    # BatchNorm of the decoders to ~1.09 on every encoder tensor.  That is a
    # property of the comparison, not of the kernels -- the oracle runs on the ENGINE's ReLU branch,
    # where its pinned "ReLU" x * m is no rectifier of its own pre-activation (smaller mean, larger
    # variance), and every batch-statistics BatchNorm returns that variance excess as a smaller
    # gradient (profiles/r05_actgrad_*, r05_bn_stats_*).  The yardstick is the
    # SAME experiment without any engine (tools/pin_artifact.py): an independent bf16-storage
    # implementation (the emulating oracle in float32, on its own branch) against the fp64
    # emulating oracle pinned to ITS decisions -- measured 1.1016 on the encoder tensors where the
    # engine shows 1.0954, and 1.032 / 1.052 / 1.065 / 1.097 / 1.118 along the decoder where the
    # engine shows 1.033 / 1.056 / 1.062 / 1.093 / 1.111.  Gate: the engine's gain follows the
    # control's, at the encoder plateau and at the check points along the backward path.
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from pin_artifact import CHECKPOINTS, run_pair
    ctrl = {k: r for k, r, _ in run_pair('emul32', 'emul64', hh, ww, bs, seed=321)[0]}
    eng = dict(zip(names, ratio.tolist()))
    enc = [k for k in names if k.startswith('encoder') and k in ctrl]
    med_e = torch.tensor([eng[k] for k in enc]).median().item()
    med_c = torch.tensor([ctrl[k] for k in enc]).median().item()
    print("gradient-norm gain on %d encoder tensors: engine / pinned oracle median %.4f; two-oracle "
          "control (no engine) %.4f; check points engine | control: %s" % (
              len(enc), med_e, med_c,
              ' '.join('%.3f|%.3f' % (eng[k], ctrl[k]) for k in CHECKPOINTS if k in eng and k in ctrl)))
    assert abs(med_e - med_c) <= 0.04, (med_e, med_c)
    for k in CHECKPOINTS:
        if k in eng and k in ctrl:
            assert abs(eng[k] - ctrl[k]) <= 0.04, (k, eng[k], ctrl[k])
    # extremes: [0.6, 1.4] for every tensor of >= 1024 elements outside the squeeze-excite MLPs.  The
    # loose class (tensors of < 1024 elements -- SE fc biases of 4-32 elements, one-element head biases:
    # sums over 8 samples / all pixels that nearly cancel -- and the SE linears whatever their size) has
    # NO ratio gate in this train-mode test any more: its ratios are one draw of a chaotic system that
    # moves with every last-bit change of any kernel upstream (measured on encoder.fusion_modules.2.
    # se_depth.fc.0.bias over the builds of rounds 4-6: 1.51, 1.47, 0.40, 0.24), so a band wide enough
    # to hold them ([0.25, 4] in round 5) no longer separated a factor of two.  What pins the
    # SCALE of those tensors instead: the frozen-BatchNorm test below (tight per-tensor bands for the SE
    # linears and the small tensors), tests/test_timed_size_gpu.py (every gradient of the bs-32 step
    # against the sum over two bs-16 steps at 1e-4) and the operator tests of the SE MLP backward /
    # head biases at 2e-4; here they keep the cosine gate above (>= 0.7: a sign error gives -1).
    r_ = ratio[big]
    nm = [k for k, b_ in zip(names, big.tolist()) if b_]
    lo, hi = int(r_.argmin()), int(r_.argmax())
    assert r_.min().item() >= 0.6 and r_.max().item() <= 1.4, \
        (nm[lo], r_.min().item(), nm[hi], r_.max().item())
    assert ((ratio - 1.0).abs() <= 0.15).float().mean().item() >= 0.97


def test_eval_bn_bf16_pinned_gradients_baseline_resolution(monkeypatch):
    """bf16 engine with FROZEN BatchNorm statistics (model.eval(), gradients on: the `eval_grad` path)
    at the BASELINE resolution 640x480 (bs 2: the conv_rs plans, patch tilings and launch set of
    configs[2]) against the storage-emulating fp64 oracle on the engine's ReLU branch.  Without
    batch statistics nothing renormalises the divergence of two bf16 roundings, so -- unlike the
    train-mode case above, whose norm ratios carry the pinning artefact -- per-tensor gates can be
    tight here: gradient-norm ratio centred on 1, cosine near 1."""
    import torch.nn.functional as F
    from emsanet_amd import full_args, ops
    from oracle import emsanet_oracle as O
    from test_model_gpu import _PinnedRelu
    bs, hh, ww = 2, 480, 640
    args = full_args(input_height=hh, input_width=ww)
    model, oracle = _pair(args)
    batch = O.synthetic_batch(bs, hh, ww)
    # frozen statistics OF THIS NETWORK: the deterministic state dict draws running_mean / running_var
    # at random, and an eval-mode forward through statistics that do not belong to the weights is not
    # normalised -- its head logits reach |x| > 15, where the fp32 sigmoid / tanh derivatives
    # y (1 - y), 1 - y^2 (the reference's own arithmetic, torch's formula) are quantisation noise
    # against the fp64 oracle (first run of this test: centre / offset task convs at cosine 0.90 /
    # 0.99 with the orientation conv beside them at 1.0000).  One train-mode pass of the fp32 oracle
    # with momentum 1 puts the batch statistics into the buffers of both sides (BatchNorm
    # recalibration), then both are frozen.
    with torch.no_grad():
        moms = {}
        for m in oracle.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                moms[m] = m.momentum
                m.momentum = 1.0
        oracle.train()
        oracle.dropout_seed = 321
        oracle(batch)
        for m, mom in moms.items():
            m.momentum = mom
    model.load_state_dict(oracle.state_dict())
    oracle = oracle.double()
    model.set_compute_dtype(torch.bfloat16)
    model.eval()
    oracle.eval()
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
    cots = [rnd(*t.shape, seed=100 + i, scale=1e-1) for i, t in enumerate(ref)]
    torch.autograd.backward(out, [c.to(DEV) for c in cots])
    torch.autograd.backward(ref, [c.double() for c in cots])
    monkeypatch.undo()
    pr, mp = dict(oracle.named_parameters()), dict(model.named_parameters())
    gmax = max(p.grad.abs().max().item() for p in pr.values() if p.grad is not None)
    names, ratio, cos = [], [], []
    for k, p in mp.items():
        r = pr[k].grad
        if p.grad is None or r is None or r.abs().max().item() < 1e-9 * gmax:
            continue
        assert torch.isfinite(p.grad).all(), k
        g = p.grad.detach().cpu().double().flatten()
        names.append(k)
        ratio.append(g.norm().item() / r.norm().item())
        cos.append(torch.dot(g, r.flatten()).item() / (g.norm().item() * r.norm().item()))
    ratio, cos = torch.tensor(ratio), torch.tensor(cos)
    enc = torch.tensor([k.startswith('encoder') for k in names])
    print("bf16 eval-BN 640x480: outputs rel-L2 vs emulating oracle %s; %d gradients: norm ratio median "
          "%.4f (encoder %.4f) p1 %.4f p99 %.4f min %.4f max %.4f; cosine median %.4f p1 %.4f min %.4f; "
          "%d of %d ReLU decisions differ from the oracle's own" % (
              ' '.join(f'{e:.1e}' for e in eo), len(names), ratio.median(), ratio[enc].median(),
              ratio.quantile(0.01), ratio.quantile(0.99), ratio.min(), ratio.max(), cos.median(),
              cos.quantile(0.01), cos.min(), pinned.flips, pinned.total))
    import os
    if os.path.isdir('gpurun_out'):
        with open('gpurun_out/grad_ratio_bf16_evalbn_480x640.txt', 'w') as f:
            for k, r_, c_ in zip(names, ratio.tolist(), cos.tolist()):
                f.write(f"{r_:.4f} {c_:.4f} {k}\n")
    assert max(eo) <= EVAL_GRAD_OUT_TOL, eo
    assert abs(ratio.median().item() - 1.0) <= 0.02 and abs(ratio[enc].median().item() - 1.0) <= 0.02
    assert cos.median().item() >= EVAL_GRAD_COS_MEDIAN
    lo, hi, wc = int(ratio.argmin()), int(ratio.argmax()), int(cos.argmin())
    assert ((ratio - 1.0).abs() <= 0.05).float().mean().item() >= 0.97, \
        (names[lo], ratio.min().item(), names[hi], ratio.max().item())
    assert cos.min().item() >= EVAL_GRAD_COS_MIN, (names[wc], cos.min().item())
    # the tensors the train-mode test above cannot gate on scale: every squeeze-excite
    # linear and every tensor of < 1024 elements, here without batch statistics -> no chaos
    for k, r_, c_ in zip(names, ratio.tolist(), cos.tolist()):
        small = mp[k].numel() < 1024
        if '.se_' in k:
            assert SE_RATIO[0] <= r_ <= SE_RATIO[1], (k, r_)
        elif small and k.endswith('task_convs.0.bias'):
            # one element, the derivative of a (partly saturated) sigmoid summed over all pixels
            assert SMALL_RATIO_CENTRE[0] <= r_ <= SMALL_RATIO_CENTRE[1] and c_ > 0.0, (k, r_, c_)
        elif small:
            assert SMALL_RATIO[0] <= r_ <= SMALL_RATIO[1], (k, r_)


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


@pytest.mark.parametrize('cin,cout,stride,p', [(64, 64, 1, 0.0), (64, 64, 1, 0.2), (64, 128, 2, 0.1)])
@pytest.mark.parametrize('mode', ['train', 'train_folded', 'train_fused_bnb', 'eval_grad', 'eval_fast'])
def test_nbt1d_block_bf16_vs_emulating_oracle(cin, cout, stride, p, mode, monkeypatch):
    """one NonBottleneck1D block in bf16 against the fp64 block that rounds where the engine rounds;
    'train_folded' (round 6): bn1 + ReLU formed in the loader of conv3x1_2 (emsa_conv1d_rs_inbn_t) and of
    its weight gradient (emsa_conv_wgrad_multi_inbn_t), bn1's backward passes recompute the ReLU
    decisions from y2 (emsa_bn_bwd_*_aff_t) -- 'train' is the same block with the separate normalise
    pass; 'train_fused_bnb': bn1's backward reduction inside the data gradient of conv3x1_2
    (emsa_conv1d_rs_bnb_t / emsa_conv_igemm_bnb_t -- opt-in since round 5, EMSA_BN_FUSE=1)"""
    import torch.nn.functional as F
    from emsanet_amd import functional as Fn, ops
    monkeypatch.setattr(Fn, 'BN1_FOLD16', mode == 'train_folded')
    if mode == 'train_fused_bnb':
        monkeypatch.setattr(Fn, '_BN_FUSE_ENV', '1')
    if mode in ('train_fused_bnb', 'train_folded'):
        mode = 'train'
    from emsanet_amd.nn import NonBottleneck1D
    from oracle import emsanet_oracle as O
    from test_model_gpu import _PinnedRelu
    dtype = torch.bfloat16
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
    ref = ref.double()
    x = rnd(4, cin, 24, 32, seed=1)
    xq = x.to(dtype).double()
    ref.train(mode == 'train'), blk.train(mode == 'train')
    monkeypatch.setattr(O.Spec, 'STORAGE', dtype)
    xg = x.to(dtype).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    if mode == 'eval_fast':
        with torch.no_grad():
            e = _rel_l2(blk(xg), ref(xq))
        print(f"block eval_fast rel-L2 {e:.2e}")
        assert e <= 3e-3
        return
    xr = xq.clone().requires_grad_(True)
    xg.requires_grad_(True)
    ops.MASK_TRACE = []
    try:
        yg = blk(xg)
        trace = ops.MASK_TRACE
    finally:
        ops.MASK_TRACE = None
    pinned = _PinnedRelu(trace)
    monkeypatch.setattr(F, 'relu', pinned)
    yr = ref(xr)
    e_out = _rel_l2(yg, yr)
    dy = rnd(*yr.shape, seed=2)
    yr.backward(dy.to(dtype).double())
    yg.backward(dy.to(dtype).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2))
    monkeypatch.undo()
    e_dx = _rel_l2(xg.grad, xr.grad)
    rp = dict(ref.named_parameters())
    eg = {k: _rel_l2(pg.grad, rp[k].grad) for k, pg in blk.named_parameters()
          if rp[k].grad.abs().max() > 1e-9
          and not (mode == 'train' and k.endswith(('conv1x3_1.bias', 'conv1x3_2.bias')))}
    worst = max(eg, key=eg.get)
    print(f"block {mode}: out {e_out:.2e} dx {e_dx:.2e} worst grad {worst} {eg[worst]:.2e}; "
          f"{pinned.flips}/{pinned.total} sign flips")
    assert e_out <= 3e-3 and e_dx <= 1e-2 and eg[worst] <= 2e-2


def _nhwc(t, dtype):
    return t.to(dtype).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


@pytest.mark.parametrize('mode', ['train', 'eval_grad'])
def test_decoder_module_bf16_vs_emulating_oracle(mode, monkeypatch):
    """conv3x3+BN+ReLU -> 3 x NBt1D -> side head -> learned x2 up-sampling + 1x1-fused rgb skip, in
    bf16, forward and backward (input, skip and every parameter gradient), against the fp64 module
    that rounds where the engine rounds"""
    import torch.nn.functional as F
    from emsanet_amd import decoder as D, ops
    from oracle import emsanet_oracle as O
    from test_model_gpu import _PinnedRelu
    dtype = torch.bfloat16
    cin, c, skip_c = 128, 64, 32
    torch.manual_seed(0)
    ref = O.DecoderModule(cin, c, 3, 0.2, skip_c)
    ref_side = O.SemanticSideHead(c, 40)
    lid = 0
    for m in ref.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
        if isinstance(m, O.HashDropout2d):
            m.layer_id, m.seed_fn = lid, (lambda: 7)
            lid += 1
    eng = D.DecoderModule(cin, c, 3, 0.2, skip_c)
    eng_side = D.SemanticSideHead(c, 40)
    eng.load_state_dict(ref.state_dict())
    eng_side.load_state_dict(ref_side.state_dict())
    lid = 0
    for m in eng.modules():
        if type(m).__name__ == 'Dropout2dHash':
            m.layer_id, m.seed_fn = lid, (lambda: 7)
            lid += 1
    eng.to(DEV), eng_side.to(DEV)
    ref, ref_side = ref.double(), ref_side.double()
    for m in (ref, ref_side, eng, eng_side):
        m.train(mode == 'train')
    x, skip = rnd(4, cin, 12, 16, seed=1), rnd(4, skip_c, 24, 32, seed=2)
    monkeypatch.setattr(O.Spec, 'STORAGE', dtype)
    xg, sg = _nhwc(x, dtype).requires_grad_(True), _nhwc(skip, dtype).requires_grad_(True)
    xr = x.to(dtype).double().requires_grad_(True)
    sr = skip.to(dtype).double().requires_grad_(True)
    ops.MASK_TRACE = []
    try:
        yg, side_g = eng(xg, sg, eng_side)
        trace = ops.MASK_TRACE
    finally:
        ops.MASK_TRACE = None
    # (the engine evaluates the side head in training mode only; run the oracle's the same way)
    pinned = _PinnedRelu(trace)
    monkeypatch.setattr(F, 'relu', pinned)
    yr, side_r = ref(xr, sr, ref_side)
    e_out = _rel_l2(yg, yr)
    dy = rnd(*yr.shape, seed=3)
    outs_g, outs_r, cots_g, cots_r = [yg], [yr], [_nhwc(dy, dtype)], [dy.to(dtype).double()]
    if mode == 'train':
        ds = rnd(*side_r.shape, seed=4)
        outs_g.append(ops.to_float(side_g)[:, :40]); outs_r.append(side_r)
        cots_g.append(ds.to(DEV)); cots_r.append(ds.double())
    torch.autograd.backward(outs_g, cots_g)
    torch.autograd.backward(outs_r, cots_r)
    monkeypatch.undo()
    rp = dict(ref.named_parameters())
    res = {'dx': (xg.grad, xr.grad), 'dskip': (sg.grad, sr.grad)}
    for k, pg in eng.named_parameters():
        if rp[k].grad is not None and rp[k].grad.abs().max() > 1e-9 and not (
                mode == 'train' and k.endswith(('conv1x3_1.bias', 'conv1x3_2.bias'))):
            res[k] = (pg.grad, rp[k].grad)
    worst_k = max(res, key=lambda k: _rel_l2(*res[k]))
    ratios = {k: a.detach().float().norm().item() / b.norm().item() for k, (a, b) in res.items()}
    print(f"decoder module {mode}: out {e_out:.2e}; dx {_rel_l2(*res['dx']):.2e} (norm ratio "
          f"{ratios['dx']:.4f}); dskip {_rel_l2(*res['dskip']):.2e} ({ratios['dskip']:.4f}); worst "
          f"{worst_k} {_rel_l2(*res[worst_k]):.2e}; norm ratios {min(ratios.values()):.4f}.."
          f"{max(ratios.values()):.4f}")
    assert e_out <= 1e-2
    assert all(_rel_l2(a, b) <= 3e-2 for a, b in res.values())
    assert all(abs(r - 1) <= 1e-2 for r in ratios.values())


def test_full_size_bf16_batch_consistency_and_determinism():
    """BASELINE configs[2]'s shape (bs=32, 640x480 RGB-D, all heads) in bf16 storage through
    size-independent properties (the bf16 twin of test_model_gpu.py::
    test_full_size_batch_consistency_and_determinism): (1) eval outputs of a sample do not depend on
    the batch it sits in -- at bs 32 every stride-1 3-tap conv runs on the persistent conv_rs kernel,
    at bs 1 the /16 and /32 stages fall back to the implicit GEMM, BatchNorm is folded either way --
    up to bf16 rounding-boundary flips of intermediates (same yardstick as the emulating-oracle
    test); (2) the train-mode forward (batch statistics, hash dropout) is bit-reproducible."""
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd import functional as Fn
    from emsanet_amd.model import EMSANet
    from util import deterministic_state_dict
    model = EMSANet(full_args(compute_dtype='bfloat16'), nyuv2_config())
    model.load_state_dict(deterministic_state_dict(model))
    model.to(DEV).eval()
    g = torch.Generator().manual_seed(7)
    rgb = torch.randn(32, 3, 480, 640, generator=g).to(DEV)
    depth = torch.randn(32, 1, 480, 640, generator=g).to(DEV)
    # the persistent kernel really is what runs at this size
    spec = Fn.ConvSpec(256, 256, (3, 1), 1, (1, 0))
    assert Fn.rs_supported(1, spec.geom_fwd(32, 30, 40, 256, 256))
    with torch.no_grad():
        big = _flatten(model({'rgb': rgb, 'depth': depth}))
        for i in (0, 17, 31):
            one = _flatten(model({'rgb': rgb[i:i + 1].contiguous(), 'depth': depth[i:i + 1].contiguous()}))
            errs = [_rel_l2(a[i:i + 1], b) for a, b in zip(big, one)]
            assert max(errs) <= EMU_TOL[torch.bfloat16], (i, errs)
    del big
    model.train()
    outs = []
    for _ in range(2):
        model.dropout_step = 3
        with torch.no_grad():
            outs.append([t.clone() for t in _flatten(model({'rgb': rgb, 'depth': depth}))])
    for a, b in zip(*outs):
        assert torch.equal(a, b), 'bf16 train-mode forward is not bit-reproducible'


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('twin', [True, False])
def test_half_blocks_match_separate_launches(dtype, twin, monkeypatch):
    """BASELINE configs[4] (640x480, batch 1, 16-bit): the NBt1D half-blocks at C = 64 / 128 -- encoder
    layer1 / layer2, decoder module 2 -- run as ONE launch each (emsa_nbt_half_block_t, csrc/
    conv_hb.hip: conv3x1 + ReLU -> conv1x3 + folded BatchNorm (+ residual) + ReLU, the intermediate row
    in LDS; ref emsanet/model.py:47-58).  Every model output is bit-identical to the forward with one
    launch per conv, on the twin path (19 fused launches: 6 + 7 in the encoders, 6 in the decoders)
    and without twin launches (2 x 19).  The kernel is opt-in (not faster than the launches it
    replaces); this test keeps it correct."""
    from emsanet_amd import _lib, full_args, functional as Fn, nn as enn, nyuv2_config
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import synthetic_batch
    model = EMSANet(full_args(compute_dtype='bfloat16' if dtype == torch.bfloat16 else 'float16'),
                    nyuv2_config()).to(DEV).eval()
    b = {k: v.to(DEV) for k, v in synthetic_batch(1, 480, 640, seed=3).items()}
    monkeypatch.setattr(enn, 'TWIN', twin)
    calls = {'hb': 0}
    L = _lib.lib()
    hb_fn = L.emsa_nbt_half_block_t

    class Counting:
        def __getattr__(self, name):
            if name == 'emsa_nbt_half_block_t':
                def f(*a):
                    calls['hb'] += 1
                    return hb_fn(*a)
                return f
            return getattr(L, name)
    monkeypatch.setattr(_lib, 'lib', lambda: Counting())
    with torch.no_grad():
        monkeypatch.setattr(Fn, 'HALF_BLOCK', False)
        ref = [t.clone() for t in _flatten(model(b))]
        assert calls['hb'] == 0
        monkeypatch.setattr(Fn, 'HALF_BLOCK', True)           # (opt-in: EMSA_HALF_BLOCK=1)
        got = [t.clone() for t in _flatten(model(b))]
    torch.cuda.synchronize()
    assert calls['hb'] == (19 if twin else 38), calls
    assert len(got) == len(ref)
    for a, r in zip(got, ref):
        assert torch.equal(a, r)
    # batch 16 is beyond HALF_BLOCK_MAX_PIXELS at both resolutions: the persistent conv_rs launches stay
    calls['hb'] = 0
    b32 = {k: v.to(DEV) for k, v in synthetic_batch(16, 480, 640, seed=4).items()}
    with torch.no_grad():
        model(b32)
    assert calls['hb'] == 0


@pytest.mark.parametrize('dtype', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('shape', [(1, 480, 640), (3, 96, 128)])
def test_twin_launches_match_separate_launches(dtype, shape, monkeypatch):
    """16-bit eval fast path: the rgb | depth encoder blocks and the semantic | instance decoder
    blocks run in lockstep with one twin launch per conv pair (emsa_conv1d_rs_pair_t;
    nn.FusedEncoder._forward_twin_eval, decoder.twin_bodies).  Every output is bit-identical to the
    forward with one launch per conv (EMSA_TWIN=0), and the twin path really is the one that ran"""
    from emsanet_amd import _lib, full_args, functional as Fn, nn as enn, nyuv2_config
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import synthetic_batch
    monkeypatch.setattr(Fn, 'HALF_BLOCK', False)     # (the opt-in fused half-blocks: their own test above)
    model = EMSANet(full_args(compute_dtype='bfloat16' if dtype == torch.bfloat16 else 'float16'),
                    nyuv2_config()).to(DEV).eval()
    b = {k: v.to(DEV) for k, v in synthetic_batch(*shape, seed=3).items()}
    calls = {'pair': 0, 'one': 0, 'igemm_pair': 0}
    L = _lib.lib()
    pair_fn, one_fn, ig_fn = L.emsa_conv1d_rs_pair_t, L.emsa_conv1d_rs_t, L.emsa_conv_igemm_pair_t

    class Counting:
        def __getattr__(self, name):
            if name == 'emsa_conv1d_rs_pair_t':
                def f(*a):
                    calls['pair'] += 1
                    return pair_fn(*a)
                return f
            if name == 'emsa_conv_igemm_pair_t':
                def f(*a):
                    calls['igemm_pair'] += 1
                    return ig_fn(*a)
                return f
            if name == 'emsa_conv1d_rs_t':
                def f(*a):
                    calls['one'] += 1
                    return one_fn(*a)
                return f
            return getattr(L, name)
    monkeypatch.setattr(_lib, 'lib', lambda: Counting())
    with torch.no_grad():
        monkeypatch.setattr(enn, 'TWIN', False)
        ref = [t.clone() for t in _flatten(model(b))]
        n_one = calls['one']
        assert calls['pair'] == 0 and n_one > 0
        calls['one'] = 0
        monkeypatch.setattr(enn, 'TWIN', True)
        got = [t.clone() for t in _flatten(model(b))]
    torch.cuda.synchronize()
    # ResNet-34: 16 blocks x 4 convs per encoder minus the 2 strided convs of 3 blocks = 58 pairs;
    # decoders: 3 modules x 3 blocks x 4 convs = 36 pairs
    assert calls['pair'] == 58 + 36, calls
    # + the implicit-GEMM pairs: 3 strided blocks x (conv3x1 s2, conv1x3 s2, 1x1 s2) of the encoders,
    # the 3x3 conv and the 1x1 skip-fusion conv of the 3 decoder modules
    assert calls['igemm_pair'] >= 9 + 3, calls
    assert calls['one'] == n_one - 2 * (58 + 36), (calls, n_one)
    assert len(got) == len(ref)
    for a, r in zip(got, ref):
        assert torch.equal(a, r)
    # merged post-processing dict with the PanopticHelper around the two decoders
    args = full_args(compute_dtype='bfloat16' if dtype == torch.bfloat16 else 'float16',
                     enable_panoptic=True)
    pm = EMSANet(args, nyuv2_config()).to(DEV).eval()
    with torch.no_grad():
        monkeypatch.setattr(enn, 'TWIN', False)
        r0 = pm(b, do_postprocessing=True)
        monkeypatch.setattr(enn, 'TWIN', True)
        calls['pair'] = 0
        r1 = pm(b, do_postprocessing=True)
    assert calls['pair'] == 58 + 36
    # every tensor of the merged dict -- since round 5 also the centre lists behind the NMS and the
    # instance / panoptic ids numbered after them: the top-k is exact over all NMS survivors and its
    # order total, however many pixels a saturated 16-bit sigmoid lets through
    keys = [k for k, v in r0.items() if torch.is_tensor(v)]
    assert 'semantic_output' in keys and 'instance_centers' in keys and 'scene_output' in keys
    assert 'instance_predicted_centers' in keys and 'instance_segmentation_idx' in keys
    for k in keys:
        assert torch.equal(r0[k], r1[k]), k


def _bf16_train_setup(lr=0.0):
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from emsanet_amd.optim import FusedSGD
    from emsanet_amd.parallel import GradientBuckets
    args = full_args(input_height=96, input_width=128, compute_dtype='bfloat16')
    torch.manual_seed(0)
    m = EMSANet(args, nyuv2_config()).to(DEV).train()
    m.dropout_seed = 7
    b = GradientBuckets([p for p in m.parameters() if p.requires_grad])
    o = FusedSGD(b, lr=lr, momentum=0.9, weight_decay=0.0)
    return m, b, o


@pytest.mark.parametrize('warmup', [1, 3])
def test_pack_table_is_frozen_once_a_train_graph_captured_it(warmup):
    """the captured step holds raw pointers to PackPlan's job table and arena.
    warmup=1 captures the FULL table; the first eager forward afterwards used to find a quiet step
    behind it and rebuild to LEAN -- freeing the table the next replay reads.  warmup=3 captures a
    lean table; an eager reader of a plain operand used to rebuild to full.  Now: the table, the
    arena and the lean flag do not change after a capture, and replay / eager eval / replay gives
    the losses of replay / replay."""
    from emsanet_amd.graph import GraphedTrainStep
    from oracle.emsanet_oracle import synthetic_batch
    batches = [{k: v.to(DEV) for k, v in synthetic_batch(2, 96, 128, seed=s).items()}
               for s in (1, 2, 3)]

    def loss_of(out):
        return sum((t * t).mean() for t in _flatten(out))

    def run(disturb):
        m, b, o = _bf16_train_setup()
        g = GraphedTrainStep(m, batches[0], b, o, loss_fn=loss_of, warmup=warmup)
        plan = m._pack_plan
        assert plan._captured
        ident = (plan._jobs.data_ptr(), plan._arena.data_ptr(), plan._lean)
        losses = [float(g.replay(batches[1])[0])]
        if disturb:
            m.eval()
            with torch.no_grad():
                m(batches[2])                      # quiet step behind it -> would have gone lean
                for rt in plan.rts:                # a plain-operand reader -> would have gone full
                    if rt.rs:
                        rt.packed(torch.bfloat16)
                        break
                m(batches[2])
            m.train()
            assert (plan._jobs.data_ptr(), plan._arena.data_ptr(), plan._lean) == ident
        losses.append(float(g.replay(batches[2])[0]))
        torch.cuda.synchronize()
        return losses, ident[2]
    la, lean_a = run(False)
    lb, lean_b = run(True)
    assert lean_a == lean_b == (warmup >= 2)
    assert la == lb, (la, lb)
    assert all(x == x for x in la)


def test_lean_pack_table_round_trip_matches_full(monkeypatch):
    """the lean 16-bit pack table (conv_rs convs keep only their fragment-ordered
    operands) against EMSA_PACK_LEAN=0 -- eager steps go full -> lean after a quiet step, a reader of
    a plain operand sends the table back to full, a quiet step back to lean; the forward outputs are
    bit-identical to the always-full plan's throughout."""
    from emsanet_amd import ops
    from oracle.emsanet_oracle import synthetic_batch
    batch = {k: v.to(DEV) for k, v in synthetic_batch(2, 96, 128, seed=5).items()}

    def outs(lean):
        monkeypatch.setattr(ops, 'PACK_LEAN', lean)
        m, _, _ = _bf16_train_setup()
        m.eval()
        plan = m._pack_plan
        res, states = [], []
        with torch.no_grad():
            for i in range(6):
                if i == 3:
                    rt = next(rt for rt in plan.rts if rt.rs)
                    rt.packed(torch.bfloat16)
                # a new weight version makes refresh() run its lean / full decision
                torch.autograd.graph.increment_version(plan.rts[0].conv.weight)
                res.append([t.clone() for t in _flatten_eval(m(batch))])
                states.append(plan._lean)
        return res, states
    full, s_full = outs(False)
    lean, s_lean = outs(True)
    assert not any(s_full)
    # full first, lean after a quiet step, a reader before pass 3 -> full, quiet again -> lean
    assert s_lean == [False, True, True, False, True, True], s_lean
    for a, b in zip(full, lean):
        for x, y in zip(a, b):
            assert torch.equal(x, y)


def _flatten_eval(outs):
    flat = []
    for o, sides in outs:
        flat += list(o) if isinstance(o, tuple) else [o]
    return flat


@pytest.mark.parametrize('multi', [True, False])
@pytest.mark.parametrize('c,h,w', [(64, 24, 32), (128, 13, 21), (256, 15, 20), (512, 15, 20), (64, 23, 30),
                                   (128, 61, 79), (256, 7, 9), (512, 3, 5)])
def test_bn1_fold16_is_the_unfolded_block_bit_for_bit(c, h, w, multi, monkeypatch):
    """16-bit bn1 fold (round 6) against the same block with the separate normalise pass: the loader
    forms bf16(relu(fma(y2, scale, shift))) -- the very value the normalise pass would have stored -- so
    the block output and the input gradient are BIT-identical and the parameter gradients differ by
    the fp32 summation order of the split-K partials only (odd sizes: padding rows / line ends must
    stay zero after the fold, not relu(shift))."""
    from emsanet_amd import functional as Fn, ops
    from emsanet_amd.nn import NonBottleneck1D
    # (multi = False: the block's weight gradients as single launches -- emsa_conv_wgrad_inbn_t instead of
    #  the per-job in_scale of emsa_conv_wgrad_multi_inbn_t)
    monkeypatch.setattr(Fn, 'WGRAD_MULTI', multi)
    res = []
    for fold in (False, True):
        monkeypatch.setattr(Fn, 'BN1_FOLD16', fold)
        torch.manual_seed(3)
        blk = NonBottleneck1D(c, c, 1, 0.1)
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0, 0.3)
        blk.dropout.layer_id = 5
        blk.dropout.seed_fn = lambda: 7
        blk.to(DEV).train()
        x = _nhwc(rnd(3, c, h, w, seed=1), torch.bfloat16).requires_grad_(True)
        launches = []
        orig = Fn.bn_act

        def counted(*a, **k):
            launches.append(1)
            return orig(*a, **k)
        monkeypatch.setattr(Fn, 'bn_act', counted)
        y = blk(x)
        monkeypatch.setattr(Fn, 'bn_act', orig)
        y.backward(_nhwc(rnd(3, c, h, w, seed=2), torch.bfloat16))
        torch.cuda.synchronize()
        res.append((y.detach().clone(), x.grad.clone(),
                    {k: q.grad.clone() for k, q in blk.named_parameters()}, len(launches)))
    (y0, dx0, g0, n0), (y1, dx1, g1, n1) = res
    assert n1 == n0 - 1, (n0, n1)                    # bn1's normalise pass is gone
    assert torch.equal(y0, y1), 'forward differs'
    assert torch.equal(dx0, dx1), 'input gradient differs'
    for k in g0:
        d = float((g0[k] - g1[k]).abs().max())
        assert d <= 2e-5 * max(1.0, float(g0[k].abs().max())), (k, d)


def test_bn1_fold16_whole_model_train_step_is_bit_identical(monkeypatch):
    """the whole bf16 TRAIN step (all heads, Dropout2d, side outputs) with bn1 of every NBt1D block
    folded into its consumers against the same step with the separate normalise passes: every output
    bit-identical, every parameter gradient within the fp32 summation order of the weight-gradient
    splits (the fold changes WHERE relu(bn1(y2)) is formed, not its bits; default rule: tensors
    >= 24 MiB, forced on here at a small size)"""
    from emsanet_amd import full_args, functional as Fn, nyuv2_config
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import synthetic_batch
    from util import deterministic_state_dict
    args = full_args(input_height=128, input_width=160, compute_dtype='bfloat16')
    batch = {k: v.to(DEV) for k, v in synthetic_batch(3, 128, 160, seed=9).items()}
    res = []
    for fold in (False, True):
        monkeypatch.setattr(Fn, 'BN1_FOLD16', fold)
        model = EMSANet(args, nyuv2_config())
        model.load_state_dict(deterministic_state_dict(model))
        model.to(DEV).train()
        model.dropout_seed, model.dropout_step = 11, 0
        out = _flatten(model(batch))
        g = torch.Generator().manual_seed(5)
        cots = [(torch.randn(t.shape, generator=g) * 1e-1).to(DEV) for t in out]
        torch.autograd.backward(out, cots)
        torch.cuda.synchronize()
        res.append(([t.detach().clone() for t in out],
                    {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}))
    (o0, g0), (o1, g1) = res
    for i, (a, b) in enumerate(zip(o0, o1)):
        assert torch.equal(a, b), f"output {i} differs with the fold"
    assert set(g0) == set(g1)
    gmax = max(float(v.abs().max()) for v in g0.values())
    for k in g0:
        d = float((g0[k] - g1[k]).abs().max())
        assert d <= 2e-5 * max(float(g0[k].abs().max()), 1e-3 * gmax), (k, d)


@pytest.mark.parametrize('n_sem', [37, 19])
def test_eval_bf16_other_class_counts(n_sem, monkeypatch):
    """bf16 eval forward with 37 / 19 semantic classes (channel-padded head + up-samplings) against the
    storage-emulating fp64 oracle, at the tolerance of the 40-class case"""
    from emsanet_amd import full_args
    from emsanet_amd.data import DatasetConfig
    from emsanet_amd.model import EMSANet
    from oracle import emsanet_oracle as O
    args = full_args(input_height=96, input_width=128)
    cfg = DatasetConfig(n_sem, 7)
    oracle = O.EMSANetOracle(args, cfg)
    sd = O.deterministic_state_dict(oracle, 0)
    oracle.load_state_dict(sd)
    model = EMSANet(args, cfg)
    model.load_state_dict(sd)
    model.to(DEV)
    oracle = oracle.double()
    model.set_compute_dtype(torch.bfloat16)
    model.eval(), oracle.eval()
    monkeypatch.setattr(O.Spec, 'STORAGE', torch.bfloat16)
    batch = O.synthetic_batch(3, 96, 128)
    with torch.no_grad():
        ref = _flatten(oracle({k: v.double() for k, v in batch.items()}))
        out = _flatten(model({k: v.to(DEV) for k, v in batch.items()}))
    assert out[0].shape[1] == n_sem and out[-1].shape[1] == 7
    errs = [_rel_l2(a, b) for a, b in zip(out, ref)]
    # the class count only touches the semantic head (outputs 0) and the scene head (last); the instance
    # outputs belong to a different random network per class count (the deterministic weights are
    # drawn in state-dict order) and their bf16 error moves with the draw: 1.1e-2 .. 4.3e-2 measured
    assert errs[0] <= EMU_TOL[torch.bfloat16] and errs[-1] <= EMU_TOL[torch.bfloat16], errs
    assert max(errs) <= 3 * EMU_TOL[torch.bfloat16], errs
