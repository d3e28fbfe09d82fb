# -*- coding: utf-8 -*-
"""
Decoders of the EMSANet engine -- mirror of /root/reference/emsanet/decoder.py
(`get_decoders(args, ...) -> nn.ModuleDict`, `KNOWN_DECODERS`; reference lines 26-48, 201) with
the decoder classes the reference imports from `nicr_mt_scene_analysis.model.decoder`
(reference lines 10-23) rebuilt on the HIP operators.

Supported: the 'emsanet' decoder type for the semantic and instance(+orientation) tasks and the
scene-classification head, the `PanopticHelper` wrapper with the eval-time panoptic merge --
the full multi-task configuration of BASELINE.json.  The 'segformermlp' decoders are outside the hot path and raise
NotImplementedError.
"""
from collections import OrderedDict
from typing import Tuple, Union

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .nn import (BasicBlock, ConvNormAct, LearnedUpsampling, NonBottleneck1D, Spec, make_plain_conv_rt, make_upsampling,
                 plain_conv)
from .postprocessing import (InstancePostprocessing, PanopticPostprocessing, gt_instance_orientations,
                             softmax_argmax)

_FUSIONS = ('add-rgb', 'add-depth', 'add-rgbd', 'add')     # encoder -> decoder skip fusions built here

KNOWN_DECODERS = (
    'emsanet',         # decoder used in EMSANet publication
    'segformermlp',    # MLP decoder used in SegFormer publication (not on the hot path)
)


class SemanticSideHead(nn.Module):
    """1x1 side-output head; key fragments ('semantic_decoder', 'head', 'conv') -- the ones the
    reference's checkpoint surgery resizes (/root/reference/emsanet/weights.py:95-119)."""

    def __init__(self, c, n_classes):
        super().__init__()
        k = Spec.SIDE_OUTPUT_KERNEL
        self.conv = nn.Conv2d(c, n_classes, k, padding=k // 2)
        self._rt = make_plain_conv_rt(self.conv)

    def forward(self, x):
        return plain_conv(self._rt, x)                 # channels padded to a multiple of 4


class InstanceSideHead(nn.Module):
    """per-task 1x1 side heads ('...head...task_convs.{0,1,2}', weights.py:28-52) evaluated as
    ONE C -> 8 GEMM (output channel offsets 0, 1, 3)."""

    def __init__(self, c, with_orientation):
        super().__init__()
        outs = (1, 2, 2) if with_orientation else (1, 2)
        k = Spec.SIDE_OUTPUT_KERNEL
        self.task_convs = nn.ModuleList([nn.Conv2d(c, o, k, padding=k // 2) for o in outs])
        placements, co = [], 0
        for conv, o in zip(self.task_convs, outs):
            placements.append((conv, co, 0))
            co += o
        self._rt = ops.MultiConvRT(placements, Fn.pad8(sum(outs)), c, k, k // 2)

    def forward(self, x):
        return ops.MultiConvFunction.apply(x, self._rt, *self._rt.params())


DECODER_BLOCKS = {'nonbottleneck1d': NonBottleneck1D, 'basicblock': BasicBlock}


def decoder_block_class(name):
    """`get_block_class(args.<task>_decoder_block, dropout_p=...)` of /root/reference/emsanet/decoder.py:
    68-71,100-103,167-170 for the blocks that keep the channel count (the decoder modules chain
    `n_blocks` of them at one width): the NBt1D block (default) and the basic block -- whose library
    form takes no dropout [U]; 'bottleneck' (x4 channels) is refused"""
    if name not in DECODER_BLOCKS:
        raise NotImplementedError(f"decoder block '{name}' (built: {', '.join(DECODER_BLOCKS)})")
    return DECODER_BLOCKS[name]


DEFAULT_UPSAMPLING = 'learned-3x3-zeropad'          # /root/reference/emsanet/args.py:280-298,363-372


class DecoderModule(nn.Module):
    """conv3x3+BN+ReLU -> n_blocks x NBt1D -> [train: 1x1 side head] -> nearest x2 + DW3x3
    -> + (1x1 conv+BN+ReLU of the rgb skip)          (figure doc/EMSANet-model.png)."""

    def __init__(self, cin, c, n_blocks, dropout_p, skip_c, upsampling=DEFAULT_UPSAMPLING,
                 block='nonbottleneck1d'):
        super().__init__()
        self.conv3x3 = ConvNormAct(cin, c, 3)
        cls = decoder_block_class(block)
        self.blocks = nn.Sequential(*[cls(c, c, dropout_p=dropout_p) for _ in range(n_blocks)])
        self.upsampling = make_upsampling(upsampling, c)
        fuse = Spec.SKIP_FUSION_1X1 == 'always' or (Spec.SKIP_FUSION_1X1 and skip_c != c)
        self.skip_fusion = ConvNormAct(skip_c, c, 1) if fuse else None

    def forward(self, x, skip, side_head):
        x = self.blocks(self.conv3x3(x))
        side = side_head(x) if self.training else None
        if self.skip_fusion is not None:
            skip = self.skip_fusion(skip)
        return self.upsampling(x, skip), side


class DecoderBody(nn.Module):
    def __init__(self, n_channels_in, n_channels, n_blocks, dropout_p, fusion_n_channels,
                 fusion_downsamplings, side_head_factory, fusion='add-rgb',
                 upsampling=DEFAULT_UPSAMPLING, prediction_upsampling=DEFAULT_UPSAMPLING,
                 block='nonbottleneck1d'):
        super().__init__()
        self.fusion = fusion
        # `upsampling`: between the decoder modules; `prediction_upsampling`: the two x2 steps of the
        # head (ref emsanet/decoder.py:55-57,78,123,176)
        mods, cin = [], n_channels_in
        for c, sc in zip(n_channels, fusion_n_channels):
            mods.append(DecoderModule(cin, c, n_blocks, dropout_p, sc, upsampling, block))
            cin = c
        self.decoder_modules = nn.ModuleList(mods)
        self.side_output_heads = nn.ModuleList([side_head_factory(c) for c in n_channels])
        self.fusion_downsamplings = tuple(fusion_downsamplings)
        self.side_output_downscales = (32, 16, 8)
        self.postprocessing = None
        self._pre = None             # (x, sides) of this forward pass when a twin launch made it

    def _skip(self, skips, ds):
        sk = skips[str(ds)]
        if self.fusion.startswith('add-'):
            return sk[self.fusion[4:]]              # 'add-rgb' (default), 'add-depth', 'add-rgbd'
        # 'add' (single-modality models, /root/reference/emsanet/tests/test_interface_model.py:32):
        # the only encoder stream there is
        if len(sk) != 1:
            raise NotImplementedError("encoder-decoder fusion 'add' with two modalities")
        return next(iter(sk.values()))

    def body(self, x, skips):
        if self._pre is not None:
            # computed in lockstep with a twin decoder (twin_bodies below) for THIS forward pass
            r, self._pre = self._pre, None
            return r
        sides = []
        plan = getattr(self, '_cut_plan', None)
        for i, (m, h, ds) in enumerate(zip(self.decoder_modules, self.side_output_heads,
                                           self.fusion_downsamplings)):
            x, s = m(x, self._skip(skips, ds), h)
            sides.append(s)
            if i == 0 and plan is not None and plan.decoder_cut and len(self.decoder_modules) > 1:
                # segmented backward (nn.CutPlan): the later modules and the head run on a leaf
                x = plan.cut(x, None, plan.DECODER_MID)
        return x, tuple(sides)


def twin_bodies_ok(da, db, x):
    """two dense decoders of one layout in the 16-bit eval fast path: their NBt1D blocks can run as
    twin launches"""
    from .nn import _fast_eval, twin_launches
    # (x = the context module's output at 1/32 of the input)
    if not (Fn.CONV_RS and _fast_eval(da) and _fast_eval(db)):
        return False
    from . import nn as enn
    if enn.TWIN is None and enn._TWIN_ENV is None:
        if x.shape[0] * x.shape[2] * x.shape[3] * 32 * 32 > enn.TWIN_MAX_PIXELS:
            return False
    elif not twin_launches():
        return False
    if x.dtype == torch.float32 or len(da.decoder_modules) != len(db.decoder_modules) or \
            da.fusion_downsamplings != db.fusion_downsamplings:
        return False
    for ma, mb in zip(da.decoder_modules, db.decoder_modules):
        if len(ma.blocks) != len(mb.blocks) or \
                ma.conv3x3.conv.out_channels != mb.conv3x3.conv.out_channels:
            return False
        if not (isinstance(ma.upsampling, LearnedUpsampling) and isinstance(mb.upsampling, LearnedUpsampling)):
            return False               # (the twin up-sampling launch is the learned kernel's)
        if not all(isinstance(b, NonBottleneck1D) for b in list(ma.blocks) + list(mb.blocks)):
            return False               # (... and the twin block launch the NBt1D block's)
    return True


def twin_bodies(da, db, x, skips):
    """DecoderBody.body of two decoders (semantic | instance, /root/reference/emsanet/decoder.py:63-139:
    same channels, blocks and resolutions) in lockstep on one stream: the 3x3 convs, the NBt1D blocks
    and the 1x1 skip-fusion convs of a module pair run as twin launches (ops.conv_bn_act_eval_pair,
    ops.nbt1d_eval_pair).  Results == the two bodies run one after the other."""
    xa = xb = x
    for ma, mb, ds in zip(da.decoder_modules, db.decoder_modules, da.fusion_downsamplings):
        ska, skb = da._skip(skips, ds), db._skip(skips, ds)
        xa, xb = ops.conv_bn_act_eval_pair(Fn.as_act(xa), Fn.as_act(xb), ma.conv3x3, mb.conv3x3)
        xa, xb = Fn.as_act(xa, dense=True), Fn.as_act(xb, dense=True)
        for ba, bb in zip(ma.blocks, mb.blocks):
            xa, xb = ops.nbt1d_eval_pair(xa, xb, ba._rt, bb._rt)
        if ma.skip_fusion is not None and mb.skip_fusion is not None:
            ska, skb = ops.conv_bn_act_eval_pair(Fn.as_act(ska), Fn.as_act(skb), ma.skip_fusion,
                                                 mb.skip_fusion)
        else:
            if ma.skip_fusion is not None:
                ska = ma.skip_fusion(ska)
            if mb.skip_fusion is not None:
                skb = mb.skip_fusion(skb)
        xa, xb = LearnedUpsampling.eval_pair(ma.upsampling, mb.upsampling, xa, xb, ska, skb)
    n = len(da.decoder_modules)
    return (xa, (None,) * n), (xb, (None,) * n)


class SemanticHead(nn.Module):
    def __init__(self, c, n_classes, upsampling=DEFAULT_UPSAMPLING):
        super().__init__()
        self.conv = nn.Conv2d(c, n_classes, 3, padding=1)
        cp = Fn.pad8(n_classes)
        self.upsampling = nn.Sequential(make_upsampling(upsampling, n_classes, cp),
                                        make_upsampling(upsampling, n_classes, cp))
        self._rt = make_plain_conv_rt(self.conv)

    def forward(self, x):
        y = plain_conv(self._rt, x)
        # the full-resolution logits leave the engine as fp32 whatever the features' storage type
        return self.upsampling[1](self.upsampling[0](y), out_f32=True)


class SemanticDecoder(DecoderBody):
    """`SemanticDecoder(...)` of /root/reference/emsanet/decoder.py:63-91."""

    def __init__(self, n_classes, **kw):
        super().__init__(side_head_factory=lambda c: SemanticSideHead(c, n_classes), **kw)
        self.n_classes = n_classes
        self.head = SemanticHead(kw['n_channels'][-1], n_classes,
                                 kw.get('prediction_upsampling', DEFAULT_UPSAMPLING))

    def forward(self, x, skips, batch=None, do_postprocessing=False):
        x, sides = self.body(x[0], skips)
        nc = self.n_classes
        out = self.head(x)[:, :nc]
        sides = tuple(ops.to_float(s)[:, :nc] for s in sides) if self.training else ()
        if not do_postprocessing:
            return out, sides
        r = {'semantic_output': out, 'semantic_side_outputs': sides}
        if not self.training:
            score, idx = softmax_argmax(out)             # one fused pass over the logits
            r['semantic_segmentation_score'], r['semantic_segmentation_idx'] = score, idx
        return r


class NormalDecoder(SemanticDecoder):
    """`NormalDecoder(...)` of /root/reference/emsanet/decoder.py:160-175 (the class itself lives in
    the un-vendored library): restated as the dense decoder with `normal_n_channels_out` output maps
    -- body, side heads and head exactly as the semantic decoder's, no activation on the maps.  The
    outputs are the raw maps; unit-length normalisation is left to the caller's post-processing
    (undetermined upstream)."""

    def forward(self, x, skips, batch=None, do_postprocessing=False):
        x, sides = self.body(x[0], skips)
        nc = self.n_classes
        out = self.head(x)[:, :nc]
        sides = tuple(ops.to_float(s)[:, :nc] for s in sides) if self.training else ()
        if not do_postprocessing:
            return out, sides
        return {'normal_output': out, 'normal_side_outputs': sides}


class InstanceHead(nn.Module):
    """shared_conv 3x3 C->32*T + norm + act; task_convs.{0,1,2} 3x3 32->1/2/2 evaluated as ONE
    block-diagonal 96->8 convolution; shared DW upsampling x2 x2
    (/root/reference/emsanet/weights.py:39-56)."""

    def __init__(self, c, with_orientation, n_per_task=32, upsampling=DEFAULT_UPSAMPLING):
        super().__init__()
        outs = (1, 2, 2) if with_orientation else (1, 2)
        self.outs = outs
        self.n_out = sum(outs)
        cp = Fn.pad8(self.n_out)
        self.shared_conv = ConvNormAct(c, n_per_task * len(outs), 3)
        self.task_convs = nn.ModuleList([nn.Conv2d(n_per_task, o, 3, padding=1) for o in outs])
        self.upsampling = nn.Sequential(make_upsampling(upsampling, self.n_out, cp),
                                        make_upsampling(upsampling, self.n_out, cp))
        placements, co = [], 0
        for i, (conv, o) in enumerate(zip(self.task_convs, outs)):
            placements.append((conv, co, i * n_per_task))
            co += o
        self._rt = ops.MultiConvRT(placements, cp, n_per_task * len(outs), 3, 1)

    def forward(self, x):
        x = self.shared_conv(x)
        y = ops.MultiConvFunction.apply(x, self._rt, *self._rt.params())
        return self.upsampling[1](self.upsampling[0](y))


class InstanceDecoder(DecoderBody):
    """`InstanceDecoder(...)` of /root/reference/emsanet/decoder.py:94-139."""

    def __init__(self, with_orientation, sigmoid_for_center, tanh_for_offset, **kw):
        self.with_orientation = with_orientation
        super().__init__(side_head_factory=lambda c: InstanceSideHead(c, with_orientation), **kw)
        self.head = InstanceHead(kw['n_channels'][-1], with_orientation,
                                 upsampling=kw.get('prediction_upsampling', DEFAULT_UPSAMPLING))
        self.sigmoid_for_center = sigmoid_for_center
        self.tanh_for_offset = tanh_for_offset
        self.normalize_orientation = bool(Spec.ORIENTATION_L2_NORMALIZE) and with_orientation

    def _split(self, y):
        n_sig = 1 if self.sigmoid_for_center else 0
        n_tanh = 2 if self.tanh_for_offset else 0
        n_norm = 2 if self.normalize_orientation else 0
        if n_sig or n_tanh or n_norm:
            if n_tanh and not n_sig:
                raise NotImplementedError("tanh offsets without sigmoid centres")
            # activation + split in one autograd node (its backward assembles the task gradients
            # with strided channel copies instead of autograd's zero-filled slice gradients)
            sizes = (1, 2, 2) if self.with_orientation else (1, 2)
            return ops.HeadActFunction.apply(y, n_sig, n_tanh, sizes, n_norm)
        y = ops.to_float(y)
        center, offset = y[:, 0:1], y[:, 1:3]
        if not self.with_orientation:
            return center, offset
        return center, offset, y[:, 3:5]

    def forward(self, x, skips, batch=None, do_postprocessing=False):
        x, sides = self.body(x[0], skips)
        out = self._split(self.head(x))
        sides = tuple(self._split(s) for s in sides) if self.training else ()
        if not do_postprocessing:
            return out, sides
        r = {'instance_output': out, 'instance_side_outputs': sides,
             'instance_centers': out[0], 'instance_offsets': out[1]}
        if self.with_orientation:
            r['instance_orientation'] = out[2]
        if not self.training and self.postprocessing is not None:
            # centre NMS / top-k and pixel grouping.  The pure instance task groups inside the GROUND-TRUTH
            # foreground: the dataset hands it over as batch['instance_foreground'] ((N,1,H,W) or (N,H,W)
            # bool: /root/reference/emsanet/tests/test_interface_model.py:60-65, visualization.py:643) and
            # the consumers read the result as 'instance_segmentation_gt_foreground'
            # (visualization.py:607-620).  (Until round 6 the mask was looked up under
            # the OUTPUT key, i.e. never found.)
            fg = None
            if batch is not None:
                fg = batch.get('instance_foreground', batch.get('instance_segmentation_gt_foreground'))
            if fg is not None and fg.dim() == 4:
                fg = fg[:, 0]
            res = self.postprocessing(out[0], out[1], fg, with_meta=True)
            r.update(res)
            if fg is not None:
                r['instance_segmentation_gt_foreground'] = res['instance_segmentation_idx']
                r['instance_segmentation_gt_meta'] = res['instance_segmentation_meta']
            if len(out) > 2:
                o = gt_instance_orientations(out[2], batch)
                if o is not None:
                    r['orientations_gt_instance_gt_orientation_foreground'] = o
        return r


class PanopticHelper(nn.Module):
    """`PanopticHelper(semantic_decoder, instance_decoder, postprocessing)` of
    /root/reference/emsanet/decoder.py:141-158: same two decoders (state-dict keys
    `decoders.panoptic_helper.{semantic,instance}_decoder.*`, weights.py:58-66), raw output
    ((semantic logits, instance outputs), (semantic side outputs, instance side outputs))
    (inference_time_whole_model.py:324-333), eval post-processing = Panoptic-DeepLab merge."""

    def __init__(self, semantic_decoder, instance_decoder, classes_is_thing, class_has_orientation=None):
        super().__init__()
        self.semantic_decoder = semantic_decoder
        self.instance_decoder = instance_decoder
        self.side_output_downscales = semantic_decoder.side_output_downscales
        # (`compute_scores=True`: /root/reference/emsanet/decoder.py:152)
        self.postprocessing = PanopticPostprocessing(
            instance_decoder.postprocessing, classes_is_thing,
            semantic_class_has_orientation=class_has_orientation, compute_scores=True)

    def forward(self, x, skips, batch=None, do_postprocessing=False):
        sem, sem_side = self.semantic_decoder(x, skips, batch, do_postprocessing=False)
        inst, inst_side = self.instance_decoder(x, skips, batch, do_postprocessing=False)
        if not do_postprocessing:
            return (sem, inst), (sem_side, inst_side)
        r = {'semantic_output': sem, 'semantic_side_outputs': sem_side, 'instance_output': inst,
             'instance_side_outputs': inst_side, 'instance_centers': inst[0],
             'instance_offsets': inst[1]}
        if len(inst) > 2:
            r['instance_orientation'] = inst[2]
        if not self.training:
            r.update(self.postprocessing(sem, inst[0], inst[1], inst[2] if len(inst) > 2 else None))
            if len(inst) > 2:
                o = gt_instance_orientations(inst[2], batch)
                if o is not None:
                    r['orientations_gt_instance_gt_orientation_foreground'] = o
        return r


class SceneClassificationDecoder(nn.Module):
    """`SceneClassificationDecoder(...)` of /root/reference/emsanet/decoder.py:191-199."""

    def __init__(self, cin, n_classes):
        super().__init__()
        self.head = nn.Linear(cin, n_classes)
        self.n_classes = n_classes
        self.side_output_downscales = ()
        self.postprocessing = None
        self._rt = ops.MultiConvRT([(self.head, 0, 0)], Fn.pad8(n_classes), cin, 1, 0)

    def forward(self, x, skips, batch=None, do_postprocessing=False):
        feat = x[1][0]                      # GAP branch of the context module (B, C, 1, 1)
        y = ops.to_float(ops.MultiConvFunction.apply(feat, self._rt, *self._rt.params()))
        out = y.flatten(1)[:, :self.n_classes]
        if not do_postprocessing:
            return out, ()
        r = {'scene_output': out}
        if not self.training:
            score, idx = softmax_argmax(out)
            r['scene_class_score'], r['scene_class_idx'] = score, idx
        return r


def get_decoders(
    args,
    n_channels_in: int,
    downsampling_in: int,
    semantic_n_classes: int = 40,
    instance_normalized_offset: bool = True,
    instance_offset_distance_threshold: Union[None, int] = None,
    instance_sigmoid_for_center: bool = True,
    instance_tanh_for_offset: bool = True,
    panoptic_semantic_classes_is_thing: Tuple[bool, ...] = (True, )*40,
    panoptic_has_orientation: Tuple[bool, ...] = (True, )*40,
    normal_n_channels_out: int = 3,
    scene_n_channels_in: int = 512//2,
    scene_n_classes: int = 10,
    fusion_n_channels: Tuple[int, ...] = (512, 256, 128),
    **kwargs
) -> nn.ModuleDict:
    """Same signature and defaults as /root/reference/emsanet/decoder.py:32-48."""
    fusion_downsamplings = tuple(args.encoder_decoder_skip_downsamplings)[::-1]
    if fusion_downsamplings != (16, 8, 4):
        # one encoder skip per decoder module at the module's output resolution; the library's handling of
        # fewer / other skips (`--encoder-decoder-skip-downsamplings`, args.py:261-268) is [U]: refused,
        # not silently built with fewer decoder modules
        raise NotImplementedError(f"encoder_decoder_skip_downsamplings={tuple(args.encoder_decoder_skip_downsamplings)}"
                                  " (only (4, 8, 16))")
    if getattr(args, 'decoder_normalization', 'batchnorm') not in ('batchnorm', 'bn'):
        raise NotImplementedError("only batchnorm decoders")
    from .nn import UPSAMPLING_MODES
    for a in ('upsampling_prediction', 'semantic_decoder_upsampling', 'instance_decoder_upsampling',
              'normal_decoder_upsampling'):
        # 'learned-3x3-zeropad' (default), 'nearest', 'bilinear'; the library's 'learned-3x3' is refused
        if getattr(args, a, DEFAULT_UPSAMPLING) not in UPSAMPLING_MODES:
            raise NotImplementedError(f"{a}={getattr(args, a)} (built: {', '.join(UPSAMPLING_MODES)})")
    up_pred = getattr(args, 'upsampling_prediction', DEFAULT_UPSAMPLING)

    # the decoder modules run at /16, /8, /4 (`*_decoder_downsamplings`, /root/reference/emsanet/decoder.py:
    # 67,99,166; default (16, 8, 4)): another schedule is refused, not silently replaced
    for task in ('semantic', 'instance', 'normal'):
        ds = getattr(args, f'{task}_decoder_downsamplings', (16, 8, 4))
        if task in args.tasks and tuple(ds) != (16, 8, 4):
            raise NotImplementedError(f"{task}_decoder_downsamplings={tuple(ds)} (only (16, 8, 4))")
        nch = getattr(args, f'{task}_decoder_n_channels', (512, 256, 128))
        if task in args.tasks and len(tuple(nch)) != 3:
            # ("Length of tuple determines the number of decoder modules", args.py:353-362: one per skip)
            raise NotImplementedError(f"{task}_decoder_n_channels={tuple(nch)} (three decoder modules)")

    decoders = OrderedDict()
    if 'semantic' in args.tasks:
        if args.semantic_decoder.lower() != 'emsanet':
            raise NotImplementedError(f"semantic decoder '{args.semantic_decoder}'")
        if args.semantic_encoder_decoder_fusion not in _FUSIONS:
            raise NotImplementedError(args.semantic_encoder_decoder_fusion)
        decoders['semantic_decoder'] = SemanticDecoder(
            n_classes=semantic_n_classes, n_channels_in=n_channels_in,
            n_channels=tuple(args.semantic_decoder_n_channels),
            n_blocks=args.semantic_decoder_n_blocks,
            dropout_p=args.semantic_decoder_block_dropout_p,
            fusion_n_channels=tuple(fusion_n_channels),
            fusion_downsamplings=fusion_downsamplings,
            fusion=args.semantic_encoder_decoder_fusion,
            upsampling=getattr(args, 'semantic_decoder_upsampling', DEFAULT_UPSAMPLING),
            prediction_upsampling=up_pred,
            block=getattr(args, 'semantic_decoder_block', 'nonbottleneck1d'))
    if 'instance' in args.tasks:
        if args.instance_decoder.lower() != 'emsanet':
            raise NotImplementedError(f"instance decoder '{args.instance_decoder}'")
        if args.instance_encoder_decoder_fusion not in _FUSIONS:
            raise NotImplementedError(args.instance_encoder_decoder_fusion)
        decoders['instance_decoder'] = InstanceDecoder(
            with_orientation='orientation' in args.tasks,
            sigmoid_for_center=instance_sigmoid_for_center,
            tanh_for_offset=instance_tanh_for_offset,
            n_channels_in=n_channels_in,
            n_channels=tuple(args.instance_decoder_n_channels),
            n_blocks=args.instance_decoder_n_blocks,
            dropout_p=args.instance_decoder_block_dropout_p,
            fusion_n_channels=tuple(fusion_n_channels),
            fusion_downsamplings=fusion_downsamplings,
            fusion=args.instance_encoder_decoder_fusion,
            upsampling=getattr(args, 'instance_decoder_upsampling', DEFAULT_UPSAMPLING),
            prediction_upsampling=up_pred,
            block=getattr(args, 'instance_decoder_block', 'nonbottleneck1d'))
        # post-processing parameters as in /root/reference/emsanet/decoder.py:95-104
        decoders['instance_decoder'].postprocessing = InstancePostprocessing(
            heatmap_threshold=args.instance_center_heatmap_threshold,
            heatmap_nms_kernel_size=args.instance_center_heatmap_nms_kernel_size,
            heatmap_apply_foreground_mask=args.instance_center_heatmap_apply_foreground_mask,
            top_k_instances=args.instance_center_heatmap_top_k,
            normalized_offset=instance_normalized_offset,
            offset_distance_threshold=instance_offset_distance_threshold)
    if getattr(args, 'enable_panoptic', False):
        # /root/reference/emsanet/decoder.py:141-158: both decoders move under one helper
        if 'semantic_decoder' not in decoders or 'instance_decoder' not in decoders:
            raise ValueError("enable_panoptic needs the semantic and the instance task")
        helper = PanopticHelper(decoders.pop('semantic_decoder'), decoders.pop('instance_decoder'),
                                panoptic_semantic_classes_is_thing, panoptic_has_orientation)
        decoders = OrderedDict([('panoptic_helper', helper)] + list(decoders.items()))
    if 'normal' in args.tasks:
        # /root/reference/emsanet/decoder.py:160-189
        if getattr(args, 'normal_decoder', 'emsanet').lower() != 'emsanet':
            raise NotImplementedError(f"normal decoder '{args.normal_decoder}'")
        fusion = getattr(args, 'normal_encoder_decoder_fusion', 'add-rgb')
        if fusion not in _FUSIONS:
            raise NotImplementedError(fusion)
        decoders['normal_decoder'] = NormalDecoder(
            n_classes=normal_n_channels_out, n_channels_in=n_channels_in,
            n_channels=tuple(getattr(args, 'normal_decoder_n_channels', (512, 256, 128))),
            n_blocks=getattr(args, 'normal_decoder_n_blocks', 3),
            dropout_p=getattr(args, 'normal_decoder_block_dropout_p', 0.2),
            fusion_n_channels=tuple(fusion_n_channels),
            fusion_downsamplings=fusion_downsamplings, fusion=fusion,
            upsampling=getattr(args, 'normal_decoder_upsampling', DEFAULT_UPSAMPLING),
            prediction_upsampling=up_pred,
            block=getattr(args, 'normal_decoder_block', 'nonbottleneck1d'))
    if 'scene' in args.tasks:
        decoders['scene_decoder'] = SceneClassificationDecoder(scene_n_channels_in,
                                                               scene_n_classes)
    return nn.ModuleDict(decoders)
