# -*- coding: utf-8 -*-
"""
ORACLE -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

Plain-PyTorch (CPU, fp32) restatement of the EMSANet forward path, i.e. of what
`/root/reference/emsanet/model.py:27-233` and `/root/reference/emsanet/decoder.py:32-201`
compose out of `nicr_mt_scene_analysis.model.*` (v0.3.1, README.md:665).

* Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
  this file.  The product (`emsanet_amd/`) must never import it.
* "PARITY UNPINNED": the arithmetic of the reference lives in an un-vendored git submodule
  (`.gitmodules:1-3`, directory empty) and the reference's own tests assert shapes/types only
  (`emsanet/tests/test_interface_model.py:96-101`, `test_interface_decoders.py:119-131`), so
  there is no golden vector to pin this restatement against.  It follows SURVEY.md App. A;
  every micro-detail that could not be read off `/root/reference` is a named constant in
  `class Spec` below ([U] = unverified against upstream).
* What IS pinned by the reference and honoured here:
    - constructor / forward contract              emsanet/model.py:27-31,192-233
    - decoder input layout + channel counts       emsanet/tests/test_interface_decoders.py:42-48,73-88
    - input shapes                                emsanet/tests/test_interface_model.py:53-59
    - default hyper-parameters                    emsanet/args.py (see emsanet_amd/args.py)
    - state-dict key fragments                    emsanet/weights.py:22-26,39-56,82-119
"""
from collections import ChainMap, OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# [U] constants.  Flip here when upstream source becomes available.
# --------------------------------------------------------------------------------------
class Spec:
    BLOCK_BN_EPS = 1e-3          # [U] ESANet's NonBottleneck1D hard-codes eps=1e-3
    DEFAULT_BN_EPS = 1e-5        # torch default everywhere else
    BN_MOMENTUM = 0.1            # torch default
    SE_REDUCTION = 16            # [U]
    PPM_BINS = (1, 5)            # [U] consistent with n_channels_reduction = 512 // 2
    STEM_BIAS = False            # [U] torchvision ResNet stem
    DW_UPSAMPLE_BIAS = True      # [U] ESANet 'learned-3x3-zeropad'
    SIDE_OUTPUT_KERNEL = 1       # [U] 1x1 conv side heads
    SKIP_FUSION_1X1 = True       # [U] 1x1 conv + BN + act on the rgb skip when channels differ
    #                              ('always': also when they match; False: never)
    ORIENTATION_L2_NORMALIZE = False   # [U] raw 2-ch biternion
    RESNET_LAYERS = {'resnet18': (2, 2, 2, 2), 'resnet34': (3, 4, 6, 3), 'resnet50': (3, 4, 6, 3),
                     'resnet101': (3, 4, 23, 3)}
    # Storage emulation (NOT part of the reference, which computes in fp32): None = plain
    # arithmetic in the module's dtype.  torch.bfloat16 / torch.float16: every tensor the 16-bit
    # engine keeps in HBM is rounded to that type where the engine rounds it (conv / BN+act / SE /
    # pooling / up-sampling outputs, conv weights; gradients of those tensors are rounded on the
    # way back), everything else -- accumulation, BatchNorm statistics (taken BEFORE the conv
    # output is rounded, like the engine's fused epilogue does), parameters, SE vectors, the model
    # outputs -- stays in the module's dtype.  tests/test_model16_gpu.py runs the fp64 oracle in
    # this mode so that the 16-bit engine can be checked at a tight tolerance: what is left is
    # accumulation order and rounding-boundary flips, not 100 layers of compounding storage noise.
    STORAGE = None


class _Store(torch.autograd.Function):
    """value AND gradient rounded to the storage type (a tensor the engine keeps in 16 bits)"""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.dtype = dtype
        return x.to(dtype).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dtype).to(g.dtype), None


class _RoundSTE(torch.autograd.Function):
    """16-bit copy of an fp32 master parameter: rounded value, gradient passed straight through
    (the engine computes the weight gradient in fp32 from the 16-bit activations)"""

    @staticmethod
    def forward(ctx, w, dtype):
        return w.to(dtype).to(w.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


def store(x):
    return x if Spec.STORAGE is None else _Store.apply(x, Spec.STORAGE)


def conv_q(conv, x):
    """conv with the engine's 16-bit operand copy of the weights (bias stays fp32)"""
    if Spec.STORAGE is None:
        return conv(x)
    return F.conv2d(x, _RoundSTE.apply(conv.weight, Spec.STORAGE), conv.bias, conv.stride,
                    conv.padding, conv.dilation, conv.groups)


def _fused_eval(module):
    """the engine's no-grad eval path folds BatchNorm into the conv epilogue: the conv output is
    never stored on its own"""
    return (not module.training) and (not torch.is_grad_enabled())


def bn_q(bn, y, fused=False):
    """BatchNorm of a conv output.  Storage emulation: the statistics come from the UNROUNDED
    accumulators (conv epilogue), the normalisation reads the STORED (rounded) conv output --
    unless the engine's fused eval path applies (no intermediate tensor)."""
    if Spec.STORAGE is None or fused:
        return bn(y)
    yq = store(y)
    if not (bn.training or bn.running_mean is None):
        return bn(yq)
    dims = (0, 2, 3)
    mean = y.mean(dims)
    var = y.var(dims, unbiased=False)
    if bn.training and bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            n = y.numel() / y.shape[1]
            mom = bn.momentum if bn.momentum is not None else 0.1
            bn.running_mean.mul_(1 - mom).add_(mom * mean)
            bn.running_var.mul_(1 - mom).add_(mom * var * (n / max(n - 1, 1)))
            bn.num_batches_tracked += 1
    inv = torch.rsqrt(var + bn.eps)
    sh = (1, -1, 1, 1)
    return (yq - mean.view(sh)) * (inv * bn.weight).view(sh) + bn.bias.view(sh)


# --------------------------------------------------------------------------------------
# Counter-based Dropout2d mask (SURVEY.md §7 hard part (vii)): the build defines it, oracle
# and HIP engine share the definition bit for bit.
# --------------------------------------------------------------------------------------
def _lowbias32(x):
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def dropout2d_scale_mask(seed, layer_id, n, c, p):
    """(n, c) float32 array: 0 where the channel is dropped else 1/(1-p)."""
    nn_, cc = np.meshgrid(np.arange(n, dtype=np.uint64), np.arange(c, dtype=np.uint64),
                          indexing='ij')
    key = _lowbias32((np.uint64(seed) + np.uint64(layer_id) * np.uint64(0x9E3779B1)) & 0xFFFFFFFF)
    h = _lowbias32((key + nn_ * np.uint64(0x85EBCA77) + cc * np.uint64(0xC2B2AE3D)) & 0xFFFFFFFF)
    u = (h >> 8).astype(np.float32) * np.float32(1.0 / 16777216.0)
    keep = u >= np.float32(p)
    return np.where(keep, np.float32(1.0 / (1.0 - p)), np.float32(0.0)).astype(np.float32)


class HashDropout2d(nn.Module):
    """Dropout2d whose channel mask is `dropout2d_scale_mask` (not torch's RNG)."""

    def __init__(self, p):
        super().__init__()
        self.p = float(p)
        self.layer_id = -1      # assigned by the model
        self.seed_fn = lambda: 0

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        m = dropout2d_scale_mask(self.seed_fn(), self.layer_id, x.shape[0], x.shape[1], self.p)
        return x * torch.from_numpy(m).to(x.device)[:, :, None, None]


# --------------------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------------------
class ConvNormAct(nn.Sequential):
    # [U] child names 'conv' / 'norm' (weights.py:39-47 shows '...shared_conv...norm...')
    def __init__(self, cin, cout, kernel_size, stride=1, act=True, eps=Spec.DEFAULT_BN_EPS):
        super().__init__()
        self.add_module('conv', nn.Conv2d(cin, cout, kernel_size, stride=stride,
                                          padding=kernel_size // 2, bias=False))
        self.add_module('norm', nn.BatchNorm2d(cout, eps=eps, momentum=Spec.BN_MOMENTUM))
        if act:
            self.add_module('act', nn.ReLU())

    def forward(self, x):
        y = bn_q(self.norm, conv_q(self.conv, x), _fused_eval(self))
        if hasattr(self, 'act'):
            y = self.act(y)
        return store(y)


class NonBottleneck1D(nn.Module):
    """conv3x1+b -> ReLU -> conv1x3+b -> BN -> ReLU -> conv3x1+b -> ReLU -> conv1x3+b -> BN
    -> Dropout2d -> + identity -> ReLU   (figure doc/EMSANet-model.png)."""

    def __init__(self, cin, cout, stride=1, dropout_p=0.0):
        super().__init__()
        self.conv3x1_1 = nn.Conv2d(cin, cout, (3, 1), stride=(stride, 1), padding=(1, 0), bias=True)
        self.conv1x3_1 = nn.Conv2d(cout, cout, (1, 3), stride=(1, stride), padding=(0, 1), bias=True)
        self.bn1 = nn.BatchNorm2d(cout, eps=Spec.BLOCK_BN_EPS, momentum=Spec.BN_MOMENTUM)
        self.conv3x1_2 = nn.Conv2d(cout, cout, (3, 1), padding=(1, 0), bias=True)
        self.conv1x3_2 = nn.Conv2d(cout, cout, (1, 3), padding=(0, 1), bias=True)
        self.bn2 = nn.BatchNorm2d(cout, eps=Spec.BLOCK_BN_EPS, momentum=Spec.BN_MOMENTUM)
        self.dropout = HashDropout2d(dropout_p)
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                nn.BatchNorm2d(cout, eps=Spec.DEFAULT_BN_EPS, momentum=Spec.BN_MOMENTUM))
        else:
            self.downsample = None

    def forward(self, x):
        fused = _fused_eval(self)
        out = store(F.relu(conv_q(self.conv3x1_1, x)))
        out = store(F.relu(bn_q(self.bn1, conv_q(self.conv1x3_1, out), fused)))
        out = store(F.relu(conv_q(self.conv3x1_2, out)))
        out = bn_q(self.bn2, conv_q(self.conv1x3_2, out), fused)
        out = self.dropout(out)
        identity = x if self.downsample is None else \
            store(bn_q(self.downsample[1], conv_q(self.downsample[0], x), fused))
        return store(F.relu(out + identity))


class _ResidualBlock(nn.Module):
    """conv -> BN -> ReLU ... conv -> BN -> + identity -> ReLU (`--*-encoder-backbone-resnet-block`
    basicblock, args.py:159-166; the classes live in the un-vendored library:
    torchvision's layout, names and stride placement assumed [U])"""

    def _make(self, specs, cin, cout_last, stride):
        self.n_convs = len(specs)
        for i, (ci, co, k, st) in enumerate(specs, 1):
            setattr(self, f'conv{i}', nn.Conv2d(ci, co, k, stride=st, padding=k // 2, bias=False))
            setattr(self, f'bn{i}', nn.BatchNorm2d(co, eps=Spec.DEFAULT_BN_EPS,
                                                   momentum=Spec.BN_MOMENTUM))
        if stride != 1 or cin != cout_last:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout_last, 1, stride=stride, bias=False),
                nn.BatchNorm2d(cout_last, eps=Spec.DEFAULT_BN_EPS, momentum=Spec.BN_MOMENTUM))
        else:
            self.downsample = None

    def forward(self, x):
        fused = _fused_eval(self)
        out = x
        for i in range(1, self.n_convs):
            conv, bn = getattr(self, f'conv{i}'), getattr(self, f'bn{i}')
            out = store(F.relu(bn_q(bn, conv_q(conv, out), fused)))
        conv, bn = getattr(self, f'conv{self.n_convs}'), getattr(self, f'bn{self.n_convs}')
        out = bn_q(bn, conv_q(conv, out), fused)
        identity = x if self.downsample is None else \
            store(bn_q(self.downsample[1], conv_q(self.downsample[0], x), fused))
        return store(F.relu(out + identity))


class BasicBlock(_ResidualBlock):
    expansion = 1

    def __init__(self, cin, c, stride=1, dropout_p=0.0):
        super().__init__()
        self._make([(cin, c, 3, stride), (c, c, 3, 1)], cin, c, stride)


class Bottleneck(_ResidualBlock):
    """1x1 -> 3x3 (stride) -> 1x1 (x4), torchvision v1.5 [U]; `--*-encoder-backbone-resnet-block
    bottleneck` (inference_time.bash:13)"""
    expansion = 4

    def __init__(self, cin, c, stride=1, dropout_p=0.0):
        super().__init__()
        self._make([(cin, c, 1, 1), (c, c, 3, stride), (c, 4 * c, 1, 1)], cin, 4 * c, stride)


RESNET_BLOCKS = {'nonbottleneck1d': NonBottleneck1D, 'basicblock': BasicBlock, 'bottleneck': Bottleneck}


class ResNetNBt1D(nn.Module):
    def __init__(self, name, n_input_channels, dropout_p, block='nonbottleneck1d'):
        super().__init__()
        layers = Spec.RESNET_LAYERS[name]
        cls = RESNET_BLOCKS[block]
        exp = getattr(cls, 'expansion', 1)
        self.conv1 = nn.Conv2d(n_input_channels, 64, 7, stride=2, padding=3, bias=Spec.STEM_BIAS)
        self.bn1 = nn.BatchNorm2d(64, eps=Spec.DEFAULT_BN_EPS, momentum=Spec.BN_MOMENTUM)
        cin = 64
        for i, (c, n) in enumerate(zip((64, 128, 256, 512), layers)):
            blocks = []
            for j in range(n):
                blocks.append(cls(cin, c, stride=2 if (i > 0 and j == 0) else 1,
                                  dropout_p=dropout_p))
                cin = c * exp
            setattr(self, f'layer{i + 1}', nn.Sequential(*blocks))
        self.stage_channels = (64, 64 * exp, 128 * exp, 256 * exp, 512 * exp)
        self.stage_downsamplings = (2, 4, 8, 16, 32)

    def forward_stage(self, i, x):
        if i == 0:
            # (the fp32 network input is converted to the storage type when the stem packs it)
            return store(F.relu(bn_q(self.bn1, conv_q(self.conv1, store(x)), _fused_eval(self))))
        if i == 1:
            x = F.max_pool2d(x, 3, stride=2, padding=1)
        return getattr(self, f'layer{i}')(x)


class SqueezeAndExcitation(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.fc = nn.Sequential(nn.Conv2d(c, c // Spec.SE_REDUCTION, 1), nn.ReLU(),
                                nn.Conv2d(c // Spec.SE_REDUCTION, c, 1), nn.Sigmoid())

    def forward(self, x):
        return x * self.fc(F.adaptive_avg_pool2d(x, 1))


class SEAddUniRGB(nn.Module):
    """'se-add-uni-rgb' (args.py:143-147): rgb <- SE(rgb) + SE(depth); depth unchanged."""

    def __init__(self, c):
        super().__init__()
        self.se_rgb = SqueezeAndExcitation(c)
        self.se_depth = SqueezeAndExcitation(c)

    def forward(self, rgb, depth):
        return store(self.se_rgb(rgb) + self.se_depth(depth)), depth


class FusedEncoder(nn.Module):
    def __init__(self, backbone_rgb, backbone_depth, fusion, skip_downsamplings,
                 backbone_rgbd=None):
        super().__init__()
        self.backbone_rgb = backbone_rgb
        self.backbone_depth = backbone_depth
        self.backbone_rgbd = backbone_rgbd          # model.py:76-92: one encoder over cat(rgb, depth)
        assert backbone_rgbd is None or (backbone_rgb is None and backbone_depth is None)
        bb = backbone_rgb if backbone_rgb is not None else backbone_depth
        bb = bb if bb is not None else backbone_rgbd
        self.two = backbone_rgb is not None and backbone_depth is not None
        if self.two:
            assert fusion == 'se-add-uni-rgb', fusion
            self.fusion_modules = nn.ModuleList([SEAddUniRGB(c) for c in bb.stage_channels])
        self.skip_downsamplings = tuple(skip_downsamplings)
        self.downsampling = 32
        self.n_channels_out = bb.stage_channels[-1]
        ch = dict(zip(bb.stage_downsamplings, bb.stage_channels))
        self.skips_n_channels = tuple(ch[d] for d in self.skip_downsamplings)

    def forward(self, inputs):
        if self.backbone_rgbd is not None:
            x, skips = inputs['rgbd'], {}
            for i, ds in enumerate(self.backbone_rgbd.stage_downsamplings):
                x = self.backbone_rgbd.forward_stage(i, x)
                if ds in self.skip_downsamplings:
                    skips[str(ds)] = {'rgbd': x}
            return {'rgbd': x}, skips
        rgb, depth = inputs.get('rgb'), inputs.get('depth')
        skips = {}
        bb = self.backbone_rgb if self.backbone_rgb is not None else self.backbone_depth
        for i, ds in enumerate(bb.stage_downsamplings):
            if rgb is not None:
                rgb = self.backbone_rgb.forward_stage(i, rgb)
            if depth is not None:
                depth = self.backbone_depth.forward_stage(i, depth)
            if self.two:
                rgb, depth = self.fusion_modules[i](rgb, depth)
            if ds in self.skip_downsamplings:
                skips[str(ds)] = {k: v for k, v in (('rgb', rgb), ('depth', depth))
                                  if v is not None}
        outs = {k: v for k, v in (('rgb', rgb), ('depth', depth)) if v is not None}
        return outs, skips


class PyramidPoolingModule(nn.Module):
    """'ppm' (args.py:243-256): bins (1,5), 1x1 conv+BN+ReLU per bin, bilinear up, concat, 1x1."""

    def __init__(self, cin, cout, input_size, upsampling='bilinear'):
        super().__init__()
        assert upsampling in ('bilinear', 'nearest')       # --upsampling-context-module, args.py:250-256
        self.upsampling = upsampling
        bins = Spec.PPM_BINS
        self.n_channels_reduction = cin // len(bins)
        self.features = nn.ModuleList([
            nn.Sequential(nn.AdaptiveAvgPool2d(b), ConvNormAct(cin, self.n_channels_reduction, 1))
            for b in bins])
        self.final_conv = ConvNormAct(cin + self.n_channels_reduction * len(bins), cout, 1)

    def forward(self, x):
        h, w = x.shape[2:]
        outs, feats = [x], []
        for f in self.features:
            y = f[1](store(f[0](x)))
            feats.append(y)
            if self.upsampling == 'nearest':
                outs.append(store(F.interpolate(y, (h, w), mode='nearest')))
            else:
                outs.append(store(F.interpolate(y, (h, w), mode='bilinear', align_corners=False)))
        return self.final_conv(torch.cat(outs, 1)), tuple(feats)


class LearnedUpsampling(nn.Module):
    """'learned-3x3-zeropad': nearest x2 then depth-wise 3x3 (zero pad) initialised to the
    bilinear kernel [[1,2,1],[2,4,2],[1,2,1]]/16 (args.py:290-298)."""

    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1, groups=c, bias=Spec.DW_UPSAMPLE_BIAS)
        w = torch.tensor([[1., 2., 1.], [2., 4., 2.], [1., 2., 1.]]) / 16.
        with torch.no_grad():
            self.conv.weight.copy_(w.expand(c, 1, 3, 3))
            if self.conv.bias is not None:
                self.conv.bias.zero_()

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2, mode='nearest'))


class PlainUpsampling(nn.Module):
    """'nearest' / 'bilinear' (align_corners=False [U]) x2: the weight-free values of
    `--*-decoder-upsampling` / `--upsampling-prediction` (args.py:280-298,363-372)"""

    def __init__(self, mode):
        super().__init__()
        assert mode in ('nearest', 'bilinear')
        self.mode = mode

    def forward(self, x):
        if self.mode == 'nearest':
            return F.interpolate(x, scale_factor=2, mode='nearest')
        return F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)


def make_upsampling(mode, c):
    return LearnedUpsampling(c) if mode == 'learned-3x3-zeropad' else PlainUpsampling(mode)


class SemanticSideHead(nn.Module):
    # key fragments ('semantic_decoder', 'head', 'conv') with shape[0] = n_classes: the only
    # semantic keys the reference's checkpoint surgery resizes (weights.py:95-119,147-160)
    def __init__(self, c, n_classes):
        super().__init__()
        self.conv = nn.Conv2d(c, n_classes, Spec.SIDE_OUTPUT_KERNEL,
                              padding=Spec.SIDE_OUTPUT_KERNEL // 2)

    def forward(self, x):
        return store(conv_q(self.conv, x))


class InstanceSideHead(nn.Module):
    # per-task convs so that "remove orientation" = delete '...head...task_convs.2' keys
    # (weights.py:28-52) without touching any other side-output shape
    def __init__(self, c, with_orientation):
        super().__init__()
        outs = (1, 2, 2) if with_orientation else (1, 2)
        k = Spec.SIDE_OUTPUT_KERNEL
        self.task_convs = nn.ModuleList([nn.Conv2d(c, o, k, padding=k // 2) for o in outs])

    def forward(self, x):
        return store(torch.cat([conv_q(conv, x) for conv in self.task_convs], dim=1))


class DecoderModule(nn.Module):
    def __init__(self, cin, c, n_blocks, dropout_p, skip_c, upsampling='learned-3x3-zeropad',
                 block='nonbottleneck1d'):
        super().__init__()
        self.conv3x3 = ConvNormAct(cin, c, 3)
        # `get_block_class(args.<task>_decoder_block, dropout_p=...)` (decoder.py:68-71,100-103,167-170)
        cls = {'nonbottleneck1d': NonBottleneck1D, 'basicblock': BasicBlock}[block]
        self.blocks = nn.Sequential(*[cls(c, c, dropout_p=dropout_p) for _ in range(n_blocks)])
        self.upsampling = make_upsampling(upsampling, c)
        if Spec.SKIP_FUSION_1X1 == 'always' or (Spec.SKIP_FUSION_1X1 and skip_c != c):
            self.skip_fusion = ConvNormAct(skip_c, c, 1)
        else:
            self.skip_fusion = None

    def forward(self, x, skip, side_head):
        x = self.blocks(self.conv3x3(x))
        side = side_head(x) if self.training else None
        x = self.upsampling(x)
        if isinstance(self.upsampling, PlainUpsampling):
            x = store(x)                    # (weight-free modes: interpolation and skip add are two kernels)
        if self.skip_fusion is not None:
            skip = self.skip_fusion(skip)
        return store(x + skip), side        # (learned up-sampling + skip add: one kernel, one rounding)


class DecoderBody(nn.Module):
    def __init__(self, n_channels_in, n_channels, n_blocks, dropout_p, fusion_n_channels,
                 fusion_downsamplings, side_head_factory, fusion='add-rgb',
                 upsampling='learned-3x3-zeropad', prediction_upsampling='learned-3x3-zeropad',
                 block='nonbottleneck1d'):
        super().__init__()
        self.fusion = fusion            # 'add-<modality>', or 'add' = the only stream there is
        mods, cin = [], n_channels_in
        for c, sc in zip(n_channels, fusion_n_channels):
            mods.append(DecoderModule(cin, c, n_blocks, dropout_p, sc, upsampling, block))
            cin = c
        self.decoder_modules = nn.ModuleList(mods)
        self.side_output_heads = nn.ModuleList([side_head_factory(c) for c in n_channels])
        self.fusion_downsamplings = tuple(fusion_downsamplings)

    def forward(self, x, skips):
        sides = []
        for m, h, ds in zip(self.decoder_modules, self.side_output_heads,
                            self.fusion_downsamplings):
            sk = skips[str(ds)]
            if self.fusion.startswith('add-'):
                skip = sk[self.fusion[4:]]
            else:
                assert len(sk) == 1, "fusion 'add' needs a single modality"
                skip = next(iter(sk.values()))
            x, s = m(x, skip, h)
            sides.append(s)
        return x, tuple(sides)


class SemanticHead(nn.Module):
    def __init__(self, c, n_classes, upsampling='learned-3x3-zeropad'):
        super().__init__()
        self.conv = nn.Conv2d(c, n_classes, 3, padding=1)
        self.upsampling = nn.Sequential(make_upsampling(upsampling, n_classes),
                                        make_upsampling(upsampling, n_classes))

    def forward(self, x):
        # head conv and the first up-sampling are stored; the last one writes the fp32 logits
        return self.upsampling[1](store(self.upsampling[0](store(conv_q(self.conv, x)))))


class SemanticDecoder(DecoderBody):
    def __init__(self, n_classes, **kw):
        super().__init__(side_head_factory=lambda c: SemanticSideHead(c, n_classes), **kw)
        self.head = SemanticHead(kw['n_channels'][-1], n_classes,
                                 kw.get('prediction_upsampling', 'learned-3x3-zeropad'))
        self.side_output_downscales = (32, 16, 8)     # module order (taken before each x2)

    def forward(self, x, skips, batch=None, do_postprocessing=False):
        x, sides = DecoderBody.forward(self, x[0], skips)
        out = self.head(x)
        sides = sides if self.training else ()
        if not do_postprocessing:
            return out, sides
        r = {'semantic_output': out, 'semantic_side_outputs': sides}
        if not self.training:
            score, idx = F.softmax(out, dim=1).max(dim=1)
            r['semantic_segmentation_score'], r['semantic_segmentation_idx'] = score, idx
        return r


class NormalDecoder(SemanticDecoder):
    """decoder.py:160-175: the class lives in the un-vendored library [U]; restated as the dense
    decoder with `normal_n_channels_out` raw output maps (no activation, no normalisation)"""

    def forward(self, x, skips, batch=None, do_postprocessing=False):
        x, sides = DecoderBody.forward(self, x[0], skips)
        out = self.head(x)
        sides = sides if self.training else ()
        if not do_postprocessing:
            return out, sides
        return {'normal_output': out, 'normal_side_outputs': sides}


class InstanceHead(nn.Module):
    """shared_conv 3x3 C->32*T (+norm+act), task_convs.{0,1,2} 3x3 32->1/2/2, shared DW
    upsampling x2 x2 over the concatenated 5 channels (weights.py:39-56)."""

    def __init__(self, c, with_orientation, n_per_task=32, upsampling='learned-3x3-zeropad'):
        super().__init__()
        outs = (1, 2, 2) if with_orientation else (1, 2)
        self.n_per_task = n_per_task
        self.shared_conv = ConvNormAct(c, n_per_task * len(outs), 3)
        self.task_convs = nn.ModuleList([nn.Conv2d(n_per_task, o, 3, padding=1) for o in outs])
        self.upsampling = nn.Sequential(make_upsampling(upsampling, sum(outs)),
                                        make_upsampling(upsampling, sum(outs)))

    def forward(self, x):
        x = self.shared_conv(x)
        parts = torch.split(x, self.n_per_task, dim=1)
        x = store(torch.cat([conv_q(conv, p) for conv, p in zip(self.task_convs, parts)], dim=1))
        # both up-samplings are stored; the head activations then produce the fp32 outputs
        return store(self.upsampling[1](store(self.upsampling[0](x))))


class InstanceDecoder(DecoderBody):
    def __init__(self, with_orientation, sigmoid_for_center, tanh_for_offset, **kw):
        self.with_orientation = with_orientation
        super().__init__(side_head_factory=lambda c: InstanceSideHead(c, with_orientation), **kw)
        self.head = InstanceHead(kw['n_channels'][-1], with_orientation,
                                 upsampling=kw.get('prediction_upsampling', 'learned-3x3-zeropad'))
        self.sigmoid_for_center = sigmoid_for_center
        self.tanh_for_offset = tanh_for_offset
        self.side_output_downscales = (32, 16, 8)

    def _split(self, y):
        center, offset = y[:, 0:1], y[:, 1:3]
        if self.sigmoid_for_center:
            center = torch.sigmoid(center)
        if self.tanh_for_offset:
            offset = torch.tanh(offset)
        if not self.with_orientation:
            return center, offset
        orientation = y[:, 3:5]
        if Spec.ORIENTATION_L2_NORMALIZE:
            orientation = F.normalize(orientation, dim=1)
        return center, offset, orientation

    def forward(self, x, skips, batch=None, do_postprocessing=False):
        x, sides = DecoderBody.forward(self, x[0], skips)
        out = self._split(self.head(x))
        sides = tuple(self._split(s) for s in sides) if self.training else ()
        if not do_postprocessing:
            return out, sides
        r = {'instance_output': out, 'instance_side_outputs': sides,
             'instance_centers': out[0], 'instance_offsets': out[1]}
        if self.with_orientation:
            r['instance_orientation'] = out[2]
        return r


class SceneClassificationDecoder(nn.Module):
    def __init__(self, cin, n_classes):
        super().__init__()
        self.head = nn.Linear(cin, n_classes)
        self.side_output_downscales = ()

    def forward(self, x, skips, batch=None, do_postprocessing=False):
        feat = x[1][0]                      # GAP branch of the context module (B,256,1,1)
        if Spec.STORAGE is None:
            out = self.head(torch.flatten(feat, 1))
        else:
            out = store(F.linear(torch.flatten(feat, 1),
                                 _RoundSTE.apply(self.head.weight, Spec.STORAGE), self.head.bias))
        if not do_postprocessing:
            return out, ()
        r = {'scene_output': out}
        if not self.training:
            score, idx = F.softmax(out, dim=1).max(dim=1)
            r['scene_class_score'], r['scene_class_idx'] = score, idx
        return r


# --------------------------------------------------------------------------------------
# the model (mirrors emsanet/model.py:27-233)
# --------------------------------------------------------------------------------------
class EMSANetOracle(nn.Module):
    def __init__(self, args, dataset_config):
        super().__init__()
        self.args = args
        self.dataset_config = dataset_config
        n_sem = len(dataset_config.semantic_label_list_without_void)
        n_scene = len(dataset_config.scene_label_list_without_void)
        mods = tuple(args.input_modalities)

        def bb(name, block, cin):
            return ResNetNBt1D(name, cin, args.dropout_p, block=block)

        b_rgb = bb(args.rgb_encoder_backbone, args.rgb_encoder_backbone_resnet_block, 3) \
            if 'rgb' in mods else None
        b_d = bb(args.depth_encoder_backbone, args.depth_encoder_backbone_resnet_block, 1) \
            if 'depth' in mods else None
        b_rgbd = bb(args.rgbd_encoder_backbone, args.rgbd_encoder_backbone_resnet_block, 3 + 1) \
            if 'rgbd' in mods else None
        self.encoder = FusedEncoder(b_rgb, b_d, args.encoder_fusion,
                                    args.encoder_decoder_skip_downsamplings, backbone_rgbd=b_rgbd)
        assert args.context_module == 'ppm'
        c_enc = self.encoder.n_channels_out
        self.context_module = PyramidPoolingModule(
            c_enc, c_enc, (args.input_height // 32, args.input_width // 32),
            upsampling=getattr(args, 'upsampling_context_module', 'bilinear'))      # (model.py:109-119)

        fus_c = self.encoder.skips_n_channels[::-1]
        fus_d = tuple(args.encoder_decoder_skip_downsamplings)[::-1]
        dec = OrderedDict()
        if 'semantic' in args.tasks:
            dec['semantic_decoder'] = SemanticDecoder(
                n_classes=n_sem, n_channels_in=c_enc,
                n_channels=tuple(args.semantic_decoder_n_channels),
                n_blocks=args.semantic_decoder_n_blocks,
                dropout_p=args.semantic_decoder_block_dropout_p,
                fusion_n_channels=fus_c, fusion_downsamplings=fus_d,
                fusion=args.semantic_encoder_decoder_fusion,
                upsampling=getattr(args, 'semantic_decoder_upsampling', 'learned-3x3-zeropad'),
                prediction_upsampling=getattr(args, 'upsampling_prediction', 'learned-3x3-zeropad'),
                block=getattr(args, 'semantic_decoder_block', 'nonbottleneck1d'))
        if 'instance' in args.tasks:
            if args.instance_offset_encoding not in ('tanh', 'relative', 'deeplab'):
                raise NotImplementedError
            dec['instance_decoder'] = InstanceDecoder(
                with_orientation='orientation' in args.tasks,
                sigmoid_for_center=args.instance_center_encoding == 'sigmoid',
                tanh_for_offset=args.instance_offset_encoding == 'tanh',
                n_channels_in=c_enc,
                n_channels=tuple(args.instance_decoder_n_channels),
                n_blocks=args.instance_decoder_n_blocks,
                dropout_p=args.instance_decoder_block_dropout_p,
                fusion_n_channels=fus_c, fusion_downsamplings=fus_d,
                fusion=args.instance_encoder_decoder_fusion,
                upsampling=getattr(args, 'instance_decoder_upsampling', 'learned-3x3-zeropad'),
                prediction_upsampling=getattr(args, 'upsampling_prediction', 'learned-3x3-zeropad'),
                block=getattr(args, 'instance_decoder_block', 'nonbottleneck1d'))
        if 'normal' in args.tasks:
            dec['normal_decoder'] = NormalDecoder(
                n_classes=3, n_channels_in=c_enc,
                n_channels=tuple(getattr(args, 'normal_decoder_n_channels', (512, 256, 128))),
                n_blocks=getattr(args, 'normal_decoder_n_blocks', 3),
                dropout_p=getattr(args, 'normal_decoder_block_dropout_p', 0.2),
                fusion_n_channels=fus_c, fusion_downsamplings=fus_d,
                fusion=getattr(args, 'normal_encoder_decoder_fusion', 'add-rgb'),
                upsampling=getattr(args, 'normal_decoder_upsampling', 'learned-3x3-zeropad'),
                prediction_upsampling=getattr(args, 'upsampling_prediction', 'learned-3x3-zeropad'),
                block=getattr(args, 'normal_decoder_block', 'nonbottleneck1d'))
        if 'scene' in args.tasks:
            dec['scene_decoder'] = SceneClassificationDecoder(
                self.context_module.n_channels_reduction, n_scene)
        self.decoders = nn.ModuleDict(dec)

        # reference init (model.py:162-190): He for encoder fusion, zero last BN gamma in decoders
        if 'encoder-fusion' in args.he_init and self.encoder.two:
            for m in self.encoder.fusion_modules.modules():
                if isinstance(m, nn.Conv2d):
                    # weights only: biases keep PyTorch's default (args.py:633-637)
                    nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        if not args.no_zero_init_decoder_residuals:
            for m in self.decoders.modules():
                if isinstance(m, (NonBottleneck1D, BasicBlock)):
                    nn.init.zeros_(m.bn2.weight)

        # dropout bookkeeping (shared definition with the HIP engine)
        self.dropout_seed = 0
        self.dropout_step = 0
        lid = 0
        for m in self.modules():
            if isinstance(m, HashDropout2d):
                m.layer_id = lid
                m.seed_fn = self._dropout_seed
                lid += 1

    def _dropout_seed(self):
        return (self.dropout_seed + 0x632BE5AB * self.dropout_step) & 0xFFFFFFFF

    def forward(self, batch, do_postprocessing=False):
        if 'rgbd' in self.args.input_modalities:        # model.py:195-199
            enc_inputs = {'rgbd': torch.cat([batch['rgb'], batch['depth']], dim=1)}
        else:
            enc_inputs = {k: batch[k] for k in ('rgb', 'depth') if k in self.args.input_modalities}
        enc_outputs, skips = self.encoder(enc_inputs)
        con_in = enc_outputs['rgb'] if len(enc_inputs) == 2 else list(enc_outputs.values())[0]
        con_out, con_ctx = self.context_module(con_in)
        outputs = [d((con_out, con_ctx), skips, batch, do_postprocessing=do_postprocessing)
                   for d in self.decoders.values()]
        if self.training:
            self.dropout_step += 1
        if do_postprocessing:
            outputs = dict(ChainMap(*outputs))
        return outputs


# --------------------------------------------------------------------------------------
# deterministic parameters (so container and GPU box regenerate identical weights)
# --------------------------------------------------------------------------------------
def deterministic_state_dict(model, seed=0):
    """Every parameter/buffer from numpy default_rng(seed) in state_dict order.  Non-trivial
    BN statistics and gammas (also the zero-initialised decoder gammas) so that parity tests
    exercise every term."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for k, v in model.state_dict().items():
        shp = tuple(v.shape)
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith('running_mean'):
            sd[k] = torch.from_numpy(rng.normal(0, 0.1, shp).astype(np.float32))
        elif k.endswith('running_var'):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif v.dim() == 1 and k.endswith('bn2.weight'):
            # residual branches enter with a small gain so that 16+9 stacked blocks keep
            # activations O(1) (well-conditioned parity tests), yet never the zero init
            sd[k] = torch.from_numpy(rng.uniform(0.2, 0.4, shp).astype(np.float32))
        elif v.dim() == 1 and (('bn' in k or 'norm' in k or k.split('.')[-2] == '1')
                               and k.endswith('weight')):
            sd[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif v.dim() == 1:
            sd[k] = torch.from_numpy(rng.normal(0, 0.05, shp).astype(np.float32))
        else:
            fan_in = int(np.prod(shp[1:]))
            std = np.sqrt(2.0 / fan_in)
            if 'upsampling' in k:            # keep DW kernels near the bilinear init
                base = np.array([[1, 2, 1], [2, 4, 2], [1, 2, 1]], np.float32) / 16
                sd[k] = torch.from_numpy(
                    (base + rng.normal(0, 0.02, shp)).astype(np.float32))
            else:
                sd[k] = torch.from_numpy(rng.normal(0, std, shp).astype(np.float32))
    return sd


def synthetic_batch(batch_size, height, width, seed=1234, modalities=('rgb', 'depth')):
    """The reference's own synthetic generator (inference_time_whole_model.py:519-545)."""
    rng = np.random.default_rng(seed)
    batch = {}
    rgb = rng.integers(0, 255, (batch_size, height, width, 3), dtype=np.uint8)
    depth = rng.integers(0, 40000, (batch_size, height, width), dtype=np.uint16)
    if 'rgb' in modalities:
        batch['rgb'] = torch.from_numpy(
            (rgb.astype(np.float32) / 255).transpose(0, 3, 1, 2).copy())
    if 'depth' in modalities:
        batch['depth'] = torch.from_numpy((depth.astype(np.float32) / 20000)[:, None].copy())
    return batch
