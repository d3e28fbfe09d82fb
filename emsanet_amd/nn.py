# -*- coding: utf-8 -*-
"""
nn.Module building blocks of the EMSANet engine.

torch's nn.Conv2d / nn.BatchNorm2d / nn.Linear objects are used ONLY as parameter and buffer
containers (so that `state_dict()` has the reference's OIHW shapes and BatchNorm buffers and
`load_state_dict(strict=True)` of a reference checkpoint layout works, /root/reference/emsanet/
weights.py:162); their `forward` is never called -- every module's forward dispatches to the
fused HIP operators in `emsanet_amd/ops.py`.

The modules stand in for `nicr_mt_scene_analysis.model.*` v0.3.1 as composed by
/root/reference/emsanet/model.py:47-160 and /root/reference/emsanet/decoder.py:61-199 (the
library itself is an un-vendored submodule; structure per SURVEY.md App. A, micro-details tagged
[U] there are the constants of `Spec`).
"""
import os

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .functional import ACT_NONE, ACT_RELU


class Spec:
    """[U] constants of SURVEY.md App. A -- the SAME table as the oracle's `Spec`
    (oracle/emsanet_oracle.py): every micro-detail of the un-vendored upstream library that could
    not be read off /root/reference is a switch here, read when a module is CONSTRUCTED, so a
    correction costs one line on each side (tests/test_model_gpu.py flips each of them)."""
    BLOCK_BN_EPS = 1e-3
    DEFAULT_BN_EPS = 1e-5
    BN_MOMENTUM = 0.1
    SE_REDUCTION = 16
    PPM_BINS = (1, 5)
    STEM_BIAS = False                  # bias on the 7x7 stem convolution
    DW_UPSAMPLE_BIAS = True            # bias on the depth-wise 3x3 of 'learned-3x3-zeropad'
    SIDE_OUTPUT_KERNEL = 1             # kernel size of the side-output heads (1 or 3)
    SKIP_FUSION_1X1 = True             # True: 1x1 conv+BN+ReLU on the rgb skip when the channel
    #                                    counts differ; 'always': also when they match; False: never
    ORIENTATION_L2_NORMALIZE = False   # L2-normalise the 2-channel orientation output
    RESNET_LAYERS = {'resnet18': (2, 2, 2, 2), 'resnet34': (3, 4, 6, 3), 'resnet50': (3, 4, 6, 3),
                     'resnet101': (3, 4, 23, 3)}


def _fast_eval(module):
    """eval-mode, no autograd graph wanted -> folded-BatchNorm fast path."""
    return (not module.training) and (not torch.is_grad_enabled())


class ConvNormAct(nn.Module):
    """conv (no bias) + BatchNorm + optional ReLU; children named 'conv' / 'norm'."""

    def __init__(self, cin, cout, kernel_size, stride=1, act=True, eps=Spec.DEFAULT_BN_EPS):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride=stride, padding=kernel_size // 2,
                              bias=False)
        self.norm = nn.BatchNorm2d(cout, eps=eps, momentum=Spec.BN_MOMENTUM)
        self.act = ACT_RELU if act else ACT_NONE
        self._crt, self._brt = ops.ConvRT(self.conv), ops.BNRT(self.norm)

    def forward(self, x):
        if _fast_eval(self):
            return ops.conv_bn_act_eval(Fn.as_act(x), self._crt, self._brt, self.act)
        return ops.ConvBNActFunction.apply(x, self._crt, self._brt, self.act, self.conv.weight,
                                           self.norm.weight, self.norm.bias)


class Dropout2dHash(nn.Module):
    """Dropout2d with the counter-based channel mask shared with the oracle."""

    def __init__(self, p):
        super().__init__()
        self.p = float(p)
        self.layer_id = -1
        self.seed_fn = lambda: 0

    def mask(self, n, c, device):
        if not self.training or self.p == 0.0:
            return None
        pre = getattr(self, '_premade', None)
        if pre is not None and pre.shape == (n, c) and pre.device == device:
            # a view into the step's mask buffer (EMSANet._prepare_dropout_masks: all layers in
            # ONE launch); consumed once
            self._premade = None
            return pre
        return Fn.dropout2d_mask(n, c, self.p, self.seed_fn(), self.layer_id, device)


class NonBottleneck1D(nn.Module):
    def __init__(self, cin, cout, stride=1, dropout_p=0.0):
        super().__init__()
        self.conv3x1_1 = nn.Conv2d(cin, cout, (3, 1), stride=(stride, 1), padding=(1, 0))
        self.conv1x3_1 = nn.Conv2d(cout, cout, (1, 3), stride=(1, stride), padding=(0, 1))
        self.bn1 = nn.BatchNorm2d(cout, eps=Spec.BLOCK_BN_EPS, momentum=Spec.BN_MOMENTUM)
        self.conv3x1_2 = nn.Conv2d(cout, cout, (3, 1), padding=(1, 0))
        self.conv1x3_2 = nn.Conv2d(cout, cout, (1, 3), padding=(0, 1))
        self.bn2 = nn.BatchNorm2d(cout, eps=Spec.BLOCK_BN_EPS, momentum=Spec.BN_MOMENTUM)
        self.dropout = Dropout2dHash(dropout_p)
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                nn.BatchNorm2d(cout, eps=Spec.DEFAULT_BN_EPS, momentum=Spec.BN_MOMENTUM))
        else:
            self.downsample = None
        self._rt = ops.NBt1DRT(self)

    def forward(self, x):
        if _fast_eval(self):
            return ops.nbt1d_eval(Fn.as_act(x, dense=True), self._rt)
        drop = self.dropout.mask(x.shape[0], self.conv1x3_2.out_channels, x.device)
        return ops.NBt1DFunction.apply(x, self._rt, drop, *self._rt.params())


class _ResidualBlock(nn.Module):
    """conv -> BN -> ReLU chains closed by conv -> BN -> + identity -> ReLU, built from the
    conv + BatchNorm functions of the decoders (no block-level fusion: an ablation option of the
    reference, `--*-encoder-backbone-resnet-block`, args.py:159-166)."""

    def _add(self, name, cin, cout, k, stride=1):
        conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False)
        bn = nn.BatchNorm2d(cout, eps=Spec.DEFAULT_BN_EPS, momentum=Spec.BN_MOMENTUM)
        setattr(self, f'conv{name}', conv)
        setattr(self, f'bn{name}', bn)
        rt = (ops.ConvRT(conv), ops.BNRT(bn), conv, bn)
        self._crts.append(rt[0])
        return rt

    def _finish(self, cin, cout, stride):
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                nn.BatchNorm2d(cout, eps=Spec.DEFAULT_BN_EPS, momentum=Spec.BN_MOMENTUM))
            self._ds = (ops.ConvRT(self.downsample[0]), ops.BNRT(self.downsample[1]),
                        self.downsample[0], self.downsample[1])
            self._crts.append(self._ds[0])
        else:
            self.downsample, self._ds = None, None

    def forward(self, x):
        fast = _fast_eval(self)
        x = Fn.as_act(x, dense=True) if fast else x
        y = x
        for crt, brt, conv, bn in self._chain[:-1]:
            y = ops.conv_bn_act_eval(y, crt, brt, ACT_RELU) if fast else \
                ops.ConvBNActFunction.apply(y, crt, brt, ACT_RELU, conv.weight, bn.weight, bn.bias)
        idn = x
        if self._ds is not None:
            crt, brt, conv, bn = self._ds
            idn = ops.conv_bn_act_eval(x, crt, brt, ACT_NONE) if fast else \
                ops.ConvBNActFunction.apply(x, crt, brt, ACT_NONE, conv.weight, bn.weight, bn.bias)
        crt, brt, conv, bn = self._chain[-1]
        if fast:
            return ops.conv_bn_act_eval(y, crt, brt, ACT_RELU, residual=idn)
        return ops.ConvBNAddActFunction.apply(y, idn, crt, brt, ACT_RELU, conv.weight, bn.weight,
                                              bn.bias)


class BasicBlock(_ResidualBlock):
    """'basicblock': conv3x3(stride) -> BN -> ReLU -> conv3x3 -> BN -> + identity -> ReLU; children
    conv1 / bn1 / conv2 / bn2 / downsample (torchvision's names [U])."""
    expansion = 1

    def __init__(self, cin, c, stride=1, dropout_p=0.0):
        super().__init__()
        self._crts = []
        self._chain = [self._add('1', cin, c, 3, stride), self._add('2', c, c, 3)]
        self._finish(cin, c, stride)


class Bottleneck(_ResidualBlock):
    """'bottleneck': conv1x1 -> BN -> ReLU -> conv3x3(stride) -> BN -> ReLU -> conv1x1 (x4 channels) -> BN
    -> + identity -> ReLU; children conv1..3 / bn1..3 / downsample, the stride on the 3x3 convolution
    (torchvision's ResNet v1.5 layout and names [U]) -- the block of `--*-encoder-backbone resnet50`
    (/root/reference/inference_time.bash:8,13; emsanet/tests/test_interface_model.py:133)"""
    expansion = 4

    def __init__(self, cin, c, stride=1, dropout_p=0.0):
        super().__init__()
        self._crts = []
        self._chain = [self._add('1', cin, c, 1), self._add('2', c, c, 3, stride),
                       self._add('3', c, 4 * c, 1)]
        self._finish(cin, 4 * c, stride)


RESNET_BLOCKS = {'nonbottleneck1d': None, 'basicblock': BasicBlock, 'bottleneck': Bottleneck}


class ResNetNBt1D(nn.Module):
    """ResNet-18/34/101 layout; NonBottleneck1D blocks (the published models) or, `block=`, basic
    blocks (both expansion 1)."""

    def __init__(self, name, n_input_channels, dropout_p, block='nonbottleneck1d'):
        super().__init__()
        if name not in Spec.RESNET_LAYERS:
            raise NotImplementedError(f"backbone '{name}' (ResNet-18/34/101 layouts)")
        if block not in RESNET_BLOCKS:
            raise NotImplementedError(f"resnet block '{block}'")
        layers = Spec.RESNET_LAYERS[name]
        self.conv1 = nn.Conv2d(n_input_channels, 64, 7, stride=2, padding=3, bias=Spec.STEM_BIAS)
        self.bn1 = nn.BatchNorm2d(64, eps=Spec.DEFAULT_BN_EPS, momentum=Spec.BN_MOMENTUM)
        cin = 64
        cls = RESNET_BLOCKS[block] or NonBottleneck1D
        exp = getattr(cls, 'expansion', 1)
        for i, (c, n) in enumerate(zip((64, 128, 256, 512), layers)):
            blocks = []
            for j in range(n):
                blocks.append(cls(cin, c, stride=2 if (i > 0 and j == 0) else 1,
                                  dropout_p=dropout_p))
                cin = c * exp
            setattr(self, f'layer{i + 1}', nn.Sequential(*blocks))
        self.stage_channels = (64, 64 * exp, 128 * exp, 256 * exp, 512 * exp)
        self.stage_downsamplings = (2, 4, 8, 16, 32)
        self._stem = ops.StemRT(self.conv1, self.bn1)
        # storage type of the activations this backbone produces (set by EMSANet.set_compute_dtype):
        # the fp32 network input is converted when the stem packs it
        self.compute_dtype = torch.float32

    def forward_stage(self, i, x):
        if i == 0:
            if _fast_eval(self):
                return ops.stem_eval(x, self._stem, self.compute_dtype)
            return ops.StemFunction.apply(x, self._stem, self.conv1.weight, self.bn1.weight,
                                          self.bn1.bias, self.conv1.bias, self.compute_dtype)
        if i == 1:
            x = ops.MaxPoolFunction.apply(x)
        return getattr(self, f'layer{i}')(x)


class SqueezeAndExcitation(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.fc = nn.Sequential(nn.Conv2d(c, c // Spec.SE_REDUCTION, 1), nn.ReLU(),
                                nn.Conv2d(c // Spec.SE_REDUCTION, c, 1), nn.Sigmoid())


class SEAddUniRGB(nn.Module):
    """'se-add-uni-rgb' (/root/reference/emsanet/args.py:143-147)."""

    def __init__(self, c):
        super().__init__()
        self.se_rgb = SqueezeAndExcitation(c)
        self.se_depth = SqueezeAndExcitation(c)

    def forward(self, rgb, depth):
        ps = ops._se_params(self.se_rgb) + ops._se_params(self.se_depth)
        # (the depth stream continues on the Function's pass-through output: its gradient from the
        #  next stage is added inside the SE backward kernel instead of by autograd)
        return ops.SEAddFunction.apply(rgb, depth, *ps)


# Two HIP streams for the two independent halves of the network (rgb | depth encoder stage,
# semantic | instance decoder): EMSA_DUAL_STREAM=0 switches it off.  Measured on one box (bs=32
# 640x480 train step): fp32 297.9 -> 306.7 images/s (+3.0 %), bf16 658.7 -> 703.6 (+6.8 %) for the
# encoder alone -- the launches of the /16 and /32 stages are a few hundred workgroups for 256 CUs,
# each a chain of dependent K steps, and fill each other's idle CUs
_DUAL_ENV = os.environ.get('EMSA_DUAL_STREAM')
DUAL_STREAM = None          # bench.py / tests: True / False overrides (None: the rule below)


def _dual_stream(t):
    if t is None or not t.is_cuda or not torch.cuda.is_available():
        return False
    if DUAL_STREAM is not None:
        return DUAL_STREAM
    if _DUAL_ENV is not None:
        return _DUAL_ENV != '0'
    return True


_TWIN_ENV = os.environ.get('EMSA_TWIN')
TWIN = None                 # bench.py / tests: True / False overrides (None: on unless EMSA_TWIN=0)


# network-input pixels per batch up to which the twin launches are used by default (one box, fp16,
# whole-model graph, twin vs two streams): batch 1 (640x480) 1.24 vs 1.73 ms (158 vs 271 nodes),
# batch 2 1.57 vs 1.74, batch 4 2.25 vs 2.06, batch 8 3.29 vs 3.11; batch 32 eager bf16 3,497 vs
# 3,623 images/s -- from batch 4 on two independent launches on two streams are the faster form.
# Second box (tools/jobs/r04_th.sh): batch 3 1.73 vs 1.82 ms; ResNet-101 encoders at 960x736 batch 1
# (706,560 pixels) 2.27 vs 2.69 ms, batch 2 (1.41 M pixels) 2.96 vs 2.84
TWIN_MAX_PIXELS = 3 * 640 * 480


def twin_launches(t=None):
    """eval fast path in 16-bit storage: twin modules (rgb | depth encoder blocks, semantic | instance
    decoder blocks) run in lockstep with one launch per conv pair (ops.nbt1d_eval_pair).
    t: a feature map at 1 / ds of the input (`_twin_ds` set by the caller) or the input itself"""
    if TWIN is not None:
        return TWIN
    if _TWIN_ENV is not None:
        return _TWIN_ENV != '0'
    return t is None or t.shape[0] * t.shape[2] * t.shape[3] <= TWIN_MAX_PIXELS


class CutPlan:
    """Autograd-graph cuts for the SEGMENTED backward pass (emsanet_amd.graph.
    SegmentedGraphedTrainStep): at a cut the forward continues on a detached copy that is a leaf
    of its own; the backward pass then runs segment by segment -- outputs -> decoder inputs ->
    encoder cut(s) -> network input -- handing the boundary gradients on by hand, so that every
    segment can be its own hipGraph with the gradient all-reduce issued eagerly in between.
    `stages`: encoder stage indices (0 stem, 1..4 layer1..4) AFTER which the encoder is cut;
    the encoder / decoder boundary is always a cut."""

    DECODERS = 99          # `stage` of the encoder/decoder boundary group
    DECODER_MID = 98       # group of the cut INSIDE the dense decoders (decoder_cut)

    def __init__(self, stages=(2, 1), decoder_cut=False):
        """decoder_cut: also cut every dense decoder behind its FIRST module (the 512-channel one:
        three quarters of the decoders' parameters).  The decoder segment then runs as two backward
        segments -- heads + later modules first, the first modules + context module + scene head
        second -- so that its gradient buckets leave in two steps instead of all at the segment's
        end."""
        self.stages = tuple(sorted(set(int(s) for s in stages), reverse=True))
        if any(s < 0 or s > 3 for s in self.stages):
            raise ValueError("CutPlan: encoder cuts go after stage 0..3")
        self.decoder_cut = bool(decoder_cut)
        self.records = []      # (original, cut copy, producing stage, group)
        self.late_stages = set()   # producing stages of the boundary tensors the FIRST decoder modules
        #                            (and the context module) read: leaves of the second decoder segment

    def begin(self):
        self.records = []

    def cut(self, t, stage, group):
        c = t.detach().requires_grad_(True)
        self.records.append((t, c, stage, group))
        return c


class FusedEncoder(nn.Module):
    """dual ResNet-NBt1D with SE-add fusion after the stem and after every layer
    (`get_encoder(...)`, /root/reference/emsanet/model.py:95-106)."""

    def __init__(self, backbone_rgb, backbone_depth, fusion, skip_downsamplings,
                 backbone_rgbd=None):
        super().__init__()
        self.backbone_rgb = backbone_rgb
        self.backbone_depth = backbone_depth
        # ONE encoder over the concatenated RGB-D image (/root/reference/emsanet/model.py:76-92)
        self.backbone_rgbd = backbone_rgbd
        if backbone_rgbd is not None and (backbone_rgb is not None or backbone_depth is not None):
            raise ValueError("input modality 'rgbd' excludes 'rgb' and 'depth'")
        bb = backbone_rgb if backbone_rgb is not None else backbone_depth
        bb = bb if bb is not None else backbone_rgbd
        self.two = backbone_rgb is not None and backbone_depth is not None
        if self.two:
            if fusion != 'se-add-uni-rgb':
                raise NotImplementedError(f"encoder fusion '{fusion}'")
            self.fusion_modules = nn.ModuleList([SEAddUniRGB(c) for c in bb.stage_channels])
        self.skip_downsamplings = tuple(skip_downsamplings)
        self._side_stream = None
        self.downsampling = 32
        self.n_channels_out = bb.stage_channels[-1]
        ch = dict(zip(bb.stage_downsamplings, bb.stage_channels))
        self.skips_n_channels = tuple(ch[d] for d in self.skip_downsamplings)

    def forward(self, inputs, plan=None):
        """plan: a CutPlan (segmented backward); skips / outputs are then returned as
        (tensor, producing stage) pairs for the caller to cut at the decoder boundary"""
        if self.backbone_rgbd is not None:
            return self._forward_single('rgbd', self.backbone_rgbd, inputs['rgbd'], plan)
        rgb, depth = inputs.get('rgb'), inputs.get('depth')
        if self._twin_eval_ok(plan, rgb):
            return self._forward_twin_eval(rgb, depth)
        skips = {}
        bb = self.backbone_rgb if self.backbone_rgb is not None else self.backbone_depth
        dual = self.two and _dual_stream(rgb)
        if dual:
            cur = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream()
            side = self._side_stream
            from .parallel import register_stream
            register_stream(cur)
            register_stream(side)
        for i, ds in enumerate(bb.stage_downsamplings):
            if dual:
                # the two encoders' stages are independent until the fusion: the depth stage runs
                # on a second stream (autograd replays its backward there too), so the small
                # launches of the /16 and /32 stages -- a few hundred workgroups for 256 CUs, each
                # a chain of dependent K steps -- fill each other's idle CUs
                # (host order rgb, then depth -- as without the second stream: tests replay the
                #  recorded ReLU decisions in call order)
                side.wait_stream(cur)
                rgb = self.backbone_rgb.forward_stage(i, rgb)
                with torch.cuda.stream(side):
                    depth = self.backbone_depth.forward_stage(i, depth)
                cur.wait_stream(side)
                depth.record_stream(cur)
            else:
                if rgb is not None:
                    rgb = self.backbone_rgb.forward_stage(i, rgb)
                if depth is not None:
                    depth = self.backbone_depth.forward_stage(i, depth)
            if self.two:
                rgb, depth = self.fusion_modules[i](rgb, depth)
            if ds in self.skip_downsamplings:
                skips[str(ds)] = {k: (v if plan is None else (v, i))
                                  for k, v in (('rgb', rgb), ('depth', depth)) if v is not None}
            if plan is not None and i in plan.stages:
                # the next stage continues on leaves of its own; the originals are the roots of
                # the backward segment that ends here
                if rgb is not None:
                    rgb = plan.cut(rgb, i, i)
                if depth is not None:
                    depth = plan.cut(depth, i, i)
        last = len(bb.stage_downsamplings) - 1
        outs = {k: (v if plan is None else (v, last))
                for k, v in (('rgb', rgb), ('depth', depth)) if v is not None}
        return outs, skips

    def _twin_eval_ok(self, plan, x=None):
        if not (self.two and plan is None and _fast_eval(self) and twin_launches(x)):
            return False
        r, d = self.backbone_rgb, self.backbone_depth
        if r.compute_dtype == torch.float32 or r.compute_dtype != d.compute_dtype or not Fn.CONV_RS:
            return False
        for i in range(1, 5):
            lr, ld = getattr(r, f'layer{i}'), getattr(d, f'layer{i}')
            if len(lr) != len(ld) or not all(isinstance(m, NonBottleneck1D) for m in list(lr) + list(ld)):
                return False
        return True

    def _forward_twin_eval(self, rgb, depth):
        """16-bit inference: the two encoders walk their stages block by block in lockstep on ONE
        stream, each conv pair as one twin launch (ops.nbt1d_eval_pair).  At batch 1 every launch
        costs its fixed ~5 us whatever it computes (profiles/r04_tl_*): 58 launches fewer per
        forward than two chains, and no cross-stream edges in the captured graph"""
        r, d = self.backbone_rgb, self.backbone_depth
        skips = {}
        for i, ds in enumerate(r.stage_downsamplings):
            if i == 0:
                rgb, depth = r.forward_stage(0, rgb), d.forward_stage(0, depth)
            else:
                if i == 1:
                    rgb, depth = ops.MaxPoolFunction.apply(rgb), ops.MaxPoolFunction.apply(depth)
                rgb, depth = Fn.as_act(rgb, dense=True), Fn.as_act(depth, dense=True)
                for br, bd in zip(getattr(r, f'layer{i}'), getattr(d, f'layer{i}')):
                    rgb, depth = ops.nbt1d_eval_pair(rgb, depth, br._rt, bd._rt)
            rgb, depth = self.fusion_modules[i](rgb, depth)
            if ds in self.skip_downsamplings:
                skips[str(ds)] = {'rgb': rgb, 'depth': depth}
        return {'rgb': rgb, 'depth': depth}, skips

    def _forward_single(self, key, bb, x, plan):
        """one encoder, no fusion modules: the stream is handed on under its modality's name"""
        skips = {}
        for i, ds in enumerate(bb.stage_downsamplings):
            x = bb.forward_stage(i, x)
            if ds in self.skip_downsamplings:
                skips[str(ds)] = {key: x if plan is None else (x, i)}
            if plan is not None and i in plan.stages:
                x = plan.cut(x, i, i)
        last = len(bb.stage_downsamplings) - 1
        return {key: x if plan is None else (x, last)}, skips

    def stage_parameters(self, i):
        """parameters of encoder stage i (both modalities + the fusion module behind it)"""
        ps = []
        for bb in (self.backbone_rgb, self.backbone_depth, self.backbone_rgbd):
            if bb is None:
                continue
            mods = [bb.conv1, bb.bn1] if i == 0 else [getattr(bb, f'layer{i}')]
            for m in mods:
                ps += list(m.parameters())
        if self.two:
            ps += list(self.fusion_modules[i].parameters())
        return ps


class PyramidPoolingModule(nn.Module):
    """'ppm' context module (/root/reference/emsanet/args.py:243-256)."""

    def __init__(self, cin, cout, input_size, upsampling='bilinear'):
        super().__init__()
        if upsampling not in ('bilinear', 'nearest'):       # (args.py:250-256: the two choices)
            raise NotImplementedError(f"upsampling_context_module='{upsampling}'")
        self.upsampling = upsampling
        bins = Spec.PPM_BINS
        self.bins = bins
        self.n_channels_reduction = cin // len(bins)
        # child layout mirrors nn.Sequential(AdaptiveAvgPool2d, ConvNormAct): index '1' is the conv
        self.features = nn.ModuleList([
            nn.Sequential(nn.Identity(), ConvNormAct(cin, self.n_channels_reduction, 1))
            for _ in bins])
        self.final_conv = ConvNormAct(cin + self.n_channels_reduction * len(bins), cout, 1)

    def forward(self, x):
        feats = []
        for b, f in zip(self.bins, self.features):
            feats.append(f[1](ops.AdaptiveAvgPoolFunction.apply(x, b)))
        cat = ops.PPMConcatFunction.apply(x, *feats, self.upsampling)
        return self.final_conv(cat), tuple(feats)


class LearnedUpsampling(nn.Module):
    """'learned-3x3-zeropad' (/root/reference/emsanet/args.py:290-298); `c_pad` >= c is the
    channel count of the (zero padded) tensor the kernel runs on."""

    def __init__(self, c, c_pad=None):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1, groups=c, bias=Spec.DW_UPSAMPLE_BIAS)
        w = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16.
        with torch.no_grad():
            self.conv.weight.copy_(w.expand(c, 1, 3, 3))
            if self.conv.bias is not None:
                self.conv.bias.zero_()
        self.c, self.c_pad = c, c_pad or c

    def _padded(self):
        w, b = self.conv.weight, self.conv.bias
        if self.c_pad != self.c:
            # without autograd (inference) the padded pair is cached per parameter state: four
            # zero-fill / concat launches per head less in the batch-1 graph
            key = None
            if not torch.is_grad_enabled():
                key = (w._version, w.data_ptr(), None if b is None else (b._version, b.data_ptr()))
                hit = getattr(self, '_pad_cache', None)
                if hit is not None and hit[0] == key:
                    return hit[1], hit[2]
            # parameter-sized glue (a few hundred floats); autograd routes the slice back
            w = torch.cat([w, w.new_zeros(self.c_pad - self.c, 1, 3, 3)], 0)
            if b is not None:
                b = torch.cat([b, b.new_zeros(self.c_pad - self.c)], 0)
            if key is not None:
                self._pad_cache = (key, w, b)
        return w, b

    def forward(self, x, skip=None, out_f32=False):
        w, b = self._padded()
        return ops.UpsampleDWFunction.apply(x, w, b, skip, out_f32)

    @staticmethod
    def eval_pair(ma, mb, xa, xb, ska, skb):
        """forward of two twin up-sampling modules (no autograd) as one twin launch, else one by one"""
        (wa, ba), (wb, bb) = ma._padded(), mb._padded()
        xa, xb = Fn.as_act(xa, dense=True), Fn.as_act(xb, dense=True)
        if ska is not None and skb is not None:
            ska, skb = Fn.as_act(ska, dense=True), Fn.as_act(skb, dense=True)
        det = lambda t: None if t is None else t.detach()      # noqa: E731
        r = Fn.up2x_dw_fwd_pair((xa, xb), (wa.detach().contiguous(), wb.detach().contiguous()),
                                (det(ba), det(bb)), (ska, skb)) if wa.shape == wb.shape else None
        return r if r is not None else (ma(xa, ska), mb(xb, skb))


class PlainUpsampling(nn.Module):
    """'nearest' / 'bilinear' x2 up-sampling (no parameters): the other two buildable values of
    `--*-decoder-upsampling` / `--upsampling-prediction` (/root/reference/emsanet/args.py:280-298);
    bilinear = align_corners False, the convention of the context module's interpolation [U]"""

    def __init__(self, mode):
        super().__init__()
        self.mode = mode

    def forward(self, x, skip=None, out_f32=False):
        return ops.PlainUpsampleFunction.apply(x, self.mode, skip, out_f32)


UPSAMPLING_MODES = ('learned-3x3-zeropad', 'nearest', 'bilinear')


def make_upsampling(mode, c, c_pad=None):
    """`get_upsampling_class(name)` of the reference's factories (emsanet/decoder.py:55-57,78,123,176)
    for the modes this engine builds; 'learned-3x3' (the variant without zero padding) is refused: how
    the un-vendored library pads it is [U]"""
    if mode == 'learned-3x3-zeropad':
        return LearnedUpsampling(c, c_pad)
    if mode in ('nearest', 'bilinear'):
        return PlainUpsampling(mode)
    raise NotImplementedError(f"upsampling '{mode}' (built: {', '.join(UPSAMPLING_MODES)})")


def make_plain_conv_rt(conv):
    """runtime for an nn.Conv2d (+bias) evaluated by the MFMA kernel with its output channels
    zero-padded to a multiple of 8 (16-byte accesses also for 16-bit storage); `plain_conv`
    returns the PADDED tensor (callers slice)."""
    k = conv.kernel_size[0]
    return ops.MultiConvRT([(conv, 0, 0)], Fn.pad8(conv.out_channels), conv.in_channels, k,
                           conv.padding[0])


def plain_conv(rt, x):
    return ops.MultiConvFunction.apply(x, rt, *rt.params())
