# -*- coding: utf-8 -*-
"""
`EMSANet(args, dataset_config)` -- the MI355X-native drop-in for the reference's
`emsanet.model.EMSANet` (/root/reference/emsanet/model.py:26-233): same constructor, same
`forward(batch, do_postprocessing=False)` contract (list of per-decoder `(outputs, side_outputs)`
tuples, or one merged dict), same attributes callers touch (`encoder`, `context_module`,
`decoders`, `state_dict()`), see SURVEY.md §8(b).  All arithmetic runs in libemsanet_hip.so.
"""
from typing import Any, Dict

import os

import torch
import torch.nn as nn

from . import _lib
from . import ops
from .decoder import get_decoders
from .nn import (Dropout2dHash, FusedEncoder, NonBottleneck1D, PyramidPoolingModule, ResNetNBt1D)


def _tensors_of(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, dict):
        return [t for v in o.values() for t in _tensors_of(v)]
    if isinstance(o, (list, tuple)):
        return [t for v in o for t in _tensors_of(v)]
    return []


class EMSANet(nn.Module):
    def __init__(self, args, dataset_config) -> None:
        super().__init__()
        _lib.lib()     # fail loudly at construction if the HIP extension is missing

        self.args, self.dataset_config = args, dataset_config          # (model.py:34-36)
        labels = dataset_config.semantic_label_list_without_void       # (model.py:39-43)
        n_sem, n_scene = len(labels), len(dataset_config.scene_label_list_without_void)

        # --- encoders (model.py:46-106): one NBt1D ResNet per modality, SE-add fusion ----------
        if getattr(args, 'activation', 'relu') != 'relu':
            raise NotImplementedError("only the default 'relu' activation has kernels")
        if getattr(args, 'encoder_normalization', 'batchnorm') not in ('batchnorm', 'bn'):
            # (model.py:47-74 hands it to the backbones: refused, not silently replaced)
            raise NotImplementedError(f"encoder_normalization='{args.encoder_normalization}' (only batchnorm)")
        nets = {}
        for modality, n_in in (('rgb', 3), ('depth', 1), ('rgbd', 3 + 1)):
            if modality not in args.input_modalities:
                nets[modality] = None
                continue
            block = getattr(args, f'{modality}_encoder_backbone_resnet_block')
            nets[modality] = ResNetNBt1D(getattr(args, f'{modality}_encoder_backbone'), n_in,
                                         args.dropout_p, block=block)
        # pretrained backbones (model.py:58-59,72-73,88-89: `pretrained=not args.no_pretrained_backbone,
        # pretrained_filepath=...`): the reference's library downloads ImageNet weights when no file is
        # named; there is no such source here -- a file is loaded, its absence is an error, never a
        # silent random initialisation
        if not getattr(args, 'no_pretrained_backbone', False):
            from .weights import load_backbone_weights
            for modality, net in nets.items():
                if net is None:
                    continue
                fp = getattr(args, f'{modality}_encoder_backbone_pretrained_weights_filepath', None)
                if not fp:
                    raise NotImplementedError(
                        f"pretrained '{modality}' backbone requested (no_pretrained_backbone is False) "
                        f"without --{modality}-encoder-backbone-pretrained-weights-filepath: the reference "
                        "downloads ImageNet weights through nicr_mt_scene_analysis, this engine loads a "
                        "file or starts from scratch with --no-pretrained-backbone")
                load_backbone_weights(net, fp, modality, verbose=bool(getattr(args, 'debug', False)))
        self.encoder = FusedEncoder(nets['rgb'], nets['depth'], args.encoder_fusion,
                                    args.encoder_decoder_skip_downsamplings,
                                    backbone_rgbd=nets['rgbd'])
        c_enc, ds_enc = self.encoder.n_channels_out, self.encoder.downsampling

        # --- context module (model.py:109-119) --------------------------------------------------
        if args.context_module != 'ppm':
            raise NotImplementedError(f"context module '{args.context_module}'")
        self.context_module = PyramidPoolingModule(
            c_enc, c_enc, (args.input_height // ds_enc, args.input_width // ds_enc),
            upsampling=getattr(args, 'upsampling_context_module', 'bilinear'))   # (model.py:109-119)

        # --- decoders (model.py:122-160) ----------------------------------------------------------
        # offset encoding -> (normalised by the image size?, tanh on the head?)
        encodings = {'tanh': (True, True), 'relative': (True, False), 'deeplab': (False, False)}
        if args.instance_offset_encoding not in encodings:
            raise NotImplementedError(args.instance_offset_encoding)
        offsets_normalised, offsets_tanh = encodings[args.instance_offset_encoding]
        head_options = dict(
            semantic_n_classes=n_sem, scene_n_classes=n_scene, normal_n_channels_out=3,
            scene_n_channels_in=self.context_module.n_channels_reduction,
            instance_normalized_offset=offsets_normalised, instance_tanh_for_offset=offsets_tanh,
            instance_sigmoid_for_center=args.instance_center_encoding == 'sigmoid',
            instance_offset_distance_threshold=args.instance_offset_distance_threshold,
            panoptic_semantic_classes_is_thing=labels.classes_is_thing,
            panoptic_has_orientation=tuple(labels.classes_use_orientations),
            fusion_n_channels=tuple(reversed(self.encoder.skips_n_channels)))
        self.decoders = get_decoders(args, n_channels_in=c_enc, downsampling_in=ds_enc,
                                     **head_options)

        # initialisation (model.py:162-190): He init of the selected parts -- convolution / linear
        # WEIGHTS only, biases keep PyTorch's default (args.py:633-637) -- then the last BatchNorm
        # gamma of every decoder block := 0
        from .decoder import DecoderModule
        from .nn import LearnedUpsampling, SEAddUniRGB

        def he_(root, skip=()):
            banned = {id(c) for b in root.modules() if isinstance(b, skip) for c in b.modules()} \
                if skip else set()
            for m in root.modules():
                if isinstance(m, (nn.Conv2d, nn.Linear)) and id(m) not in banned:
                    nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

        for part in args.he_init:
            if part == 'encoder-fusion':
                for m in self.encoder.modules():
                    if isinstance(m, SEAddUniRGB):
                        he_(m)
            elif part == 'encoder-decoder-fusion':
                for m in self.decoders.modules():
                    if isinstance(m, DecoderModule) and m.skip_fusion is not None:
                        he_(m.skip_fusion)
            elif part == 'context-module':
                he_(self.context_module)
            elif part == 'decoder':
                he_(self.decoders, skip=(LearnedUpsampling,))      # blacklist=(Upsampling,)
            else:
                raise ValueError(f"he_init part '{part}' (choices: encoder-fusion, "
                                 "encoder-decoder-fusion, context-module, decoder; args.py:626-638)")
        if not args.no_zero_init_decoder_residuals:
            from .nn import BasicBlock
            for m in self.decoders.modules():
                if isinstance(m, (NonBottleneck1D, BasicBlock)):     # (the last BatchNorm of either: bn2)
                    nn.init.zeros_(m.bn2.weight)

        # Dropout2d bookkeeping (counter-based masks, one id per dropout layer)
        self.dropout_seed = 0
        self.dropout_step = 0
        self._cut_plan = None        # nn.CutPlan: segmented backward (graph.SegmentedGraphedTrainStep)
        self._seed_dev = None        # device copy {seed, step} (see use_device_dropout_state)
        self._seed_dev_host = None   # the host values the device copy corresponds to
        lid = 0
        for m in self.modules():
            if isinstance(m, Dropout2dHash):
                m.layer_id = lid
                m.seed_fn = self._dropout_seed
                lid += 1
        self._side_stream = None     # second HIP stream of the decoders (see _run_decoders)
        self._drop_plan = None       # (key, device job table, [(layer, offset, c)], total, max_c)
        self._nbt_blocks = None

        # every convolution's weight transforms (packed / Winograd layouts) in one launch per step
        rts = []
        for m in self.modules():
            rt = getattr(m, '_rt', None)
            if isinstance(rt, ops.NBt1DRT):
                rts += [rt.c31_1, rt.c13_1, rt.c31_2, rt.c13_2] + ([rt.cds] if rt.cds else [])
            crt = getattr(m, '_crt', None)
            if isinstance(crt, ops.ConvRT):
                rts.append(crt)
            if isinstance(rt, ops.MultiConvRT):
                rts.append(rt)
            rts += list(getattr(m, '_crts', ()))        # basic / bottleneck blocks
        self._pack_plan = ops.PackPlan(rts)
        # storage type of the activations (NOT a reference option: the reference has no mixed
        # precision): 'float32' = the reference's arithmetic (default), 'bfloat16' =
        # BASELINE configs[2] mixed-precision training, 'bfloat16' / 'float16' = configs[4] inference
        self.compute_dtype = torch.float32
        self.set_compute_dtype(getattr(args, 'compute_dtype', 'float32'))
        # BatchNorm step counters are kept on the host (on each BatchNorm module) and written to
        # the buffers when a state_dict is taken -- of the model or of any sub-module -- by the
        # per-module hooks of ops.BNRT (no per-layer counter kernel in the training step)

    def set_compute_dtype(self, dtype):
        """activations (and the packed conv operands) are stored as `dtype`; parameters, BatchNorm
        statistics, SE vectors, gradients of parameters and the model's outputs stay fp32, all
        accumulation is fp32.  fp16 has no loss scaling here: inference only."""
        if isinstance(dtype, str):
            dtype = {'float32': torch.float32, 'fp32': torch.float32, 'bfloat16': torch.bfloat16,
                     'bf16': torch.bfloat16, 'float16': torch.float16, 'fp16': torch.float16}[dtype]
        if dtype not in (torch.float32, torch.bfloat16, torch.float16):
            raise ValueError(f"compute dtype {dtype}")
        self.compute_dtype = dtype
        for bb in (self.encoder.backbone_rgb, self.encoder.backbone_depth,
                   self.encoder.backbone_rgbd):
            if bb is not None:
                bb.compute_dtype = dtype
        return self

    def _dropout_seed(self):
        if self._seed_dev is not None:
            return self._seed_dev
        return (self.dropout_seed + 0x632BE5AB * self.dropout_step) & 0xFFFFFFFF

    def use_device_dropout_state(self, enable=True):
        """keep {dropout_seed, dropout_step} in device memory and let the mask kernels form the
        step's seed there (identical masks): a training step captured in a hipGraph then draws
        fresh masks at every replay.  The host attributes stay the source of truth between steps
        (`_sync_dropout_state` re-uploads them when they were changed by hand)."""
        if not enable:
            self._seed_dev = self._seed_dev_host = None
            return self
        dev = next(self.parameters()).device
        self._seed_dev = torch.zeros(2, dtype=torch.int32, device=dev)
        self._seed_dev_host = None
        self._sync_dropout_state()
        return self

    def _sync_dropout_state(self):
        if self._seed_dev is None:
            return
        now = (self.dropout_seed & 0xFFFFFFFF, self.dropout_step & 0xFFFFFFFF)
        if now != self._seed_dev_host:
            import numpy as np
            host = torch.from_numpy(np.array(now, dtype=np.uint32).view(np.int32).copy())
            self._seed_dev.copy_(host)
            self._seed_dev_host = now

    def _run_decoders(self, x, skips, batch, do_postprocessing):
        """the decoders in `self.decoders` order; with two dense decoders (semantic, instance) the
        second one runs on a second HIP stream: same inputs, independent work, and the launches of
        their /32 and /16 modules are too small to fill 256 CUs alone (nn._dual_stream)"""
        from .decoder import DecoderBody, PanopticHelper, twin_bodies, twin_bodies_ok
        from .nn import _dual_stream
        decs = list(self.decoders.values())
        dense = [d for d in decs if isinstance(d, DecoderBody)]
        # 16-bit inference: the first two dense decoders (also inside a PanopticHelper) walk their
        # modules in lockstep, block pairs as twin launches, everything on one stream
        bodies = []
        for d in decs:
            bodies += [d.semantic_decoder, d.instance_decoder] if isinstance(d, PanopticHelper) else \
                ([d] if isinstance(d, DecoderBody) else [])
        plan = self._cut_plan if self.training else None
        for bdy in bodies:
            bdy._cut_plan = plan
        twin = len(bodies) >= 2 and twin_bodies_ok(bodies[0], bodies[1], x[0])
        if twin:
            bodies[0]._pre, bodies[1]._pre = twin_bodies(bodies[0], bodies[1], x[0], skips)
        try:
            return self._run_decoder_heads(decs, dense, twin, x, skips, batch, do_postprocessing)
        finally:
            if twin:
                bodies[0]._pre = bodies[1]._pre = None

    def _run_decoder_heads(self, decs, dense, twin, x, skips, batch, do_postprocessing):
        from .nn import _dual_stream
        # (behind twin bodies only the two heads are left: they share the calling stream unless
        #  EMSA_TWIN_HEADS_DUAL=1 -- A/B switch)
        if len(dense) < 2 or not _dual_stream(x[0]) or \
                (twin and os.environ.get('EMSA_TWIN_HEADS_DUAL') != '1'):
            return [d(x, skips, batch, do_postprocessing=do_postprocessing) for d in decs]
        cur = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
        side = self._side_stream
        from .parallel import register_stream
        register_stream(cur)
        register_stream(side)
        side.wait_stream(cur)
        results = []
        for d in decs:
            if d is dense[1]:
                with torch.cuda.stream(side):
                    r = d(x, skips, batch, do_postprocessing=do_postprocessing)
                for t in _tensors_of(r):
                    t.record_stream(cur)
            else:
                r = d(x, skips, batch, do_postprocessing=do_postprocessing)
            results.append(r)
        cur.wait_stream(side)
        return results

    def _prepare_dropout_masks(self, n, device):
        """every Dropout2d mask of this training step in ONE launch (50 launches of ~5 us each
        before): the layers then pick their views up in `Dropout2dHash.mask`"""
        if self._nbt_blocks is None:
            self._nbt_blocks = [m for m in self.modules() if isinstance(m, NonBottleneck1D)]
        blocks = [m for m in self._nbt_blocks if m.training and m.dropout.p > 0.0]
        if not blocks or device.type != 'cuda':
            return
        key = (n, device, tuple((id(b), b.dropout.p, b.dropout.layer_id) for b in blocks))
        if self._drop_plan is None or self._drop_plan[0] != key:
            jobs = (_lib.EmsaDropoutJob * len(blocks))()
            layout, off = [], 0
            for j, b in enumerate(blocks):
                c = b.conv1x3_2.out_channels
                jobs[j] = _lib.EmsaDropoutJob(off, c, b.dropout.layer_id, b.dropout.p, 0)
                layout.append((b.dropout, off, c))
                off += (n * c + 3) // 4 * 4            # 16-byte aligned views
            table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(device)
            self._drop_plan = (key, table, layout, off, max(c for _, _, c in layout))
        _, table, layout, total, max_c = self._drop_plan
        from . import functional as Fn
        buf = Fn.dropout2d_mask_batch(table, len(layout), total, n, max_c, self._dropout_seed(),
                                      device)
        for layer, off, c in layout:
            layer._premade = buf[off:off + n * c].view(n, c)

    def _advance_dropout_step(self):
        self.dropout_step += 1
        if self._seed_dev is not None:
            _lib.check(_lib.lib().emsa_u32_add(self._seed_dev[1:].data_ptr(), 1,
                                               torch.cuda.current_stream().cuda_stream),
                       'emsa_u32_add')
            self._seed_dev_host = (self.dropout_seed & 0xFFFFFFFF, self.dropout_step & 0xFFFFFFFF)

    def _skip_streams_read(self):
        """names of the encoder skip streams the decoders' skip fusions read (a CutPlan
        must cut exactly these; 'add' with one modality reads the only stream there is)"""
        read = set()
        for m in self.decoders.modules():
            f = getattr(m, 'fusion', None)
            if isinstance(f, str) and f.startswith('add-'):
                read.add(f[4:])
        return read or {'rgb'}

    def forward(self, batch, do_postprocessing=False) -> Dict[str, Any]:
        """contract of /root/reference/emsanet/model.py:192-233: list of per-decoder
        (outputs, side outputs) in `self.decoders` order, or one merged dict when post-processing"""
        if 'rgbd' in self.args.input_modalities:
            # /root/reference/emsanet/model.py:195-199: the encoder's input is cat(rgb, depth)
            feeds = {'rgb': batch['rgb'], 'depth': batch['depth']}
        else:
            feeds = {m: batch[m] for m in ('rgb', 'depth') if m in self.args.input_modalities}
        for name, t in feeds.items():
            if t.dim() != 4 or t.shape[-2] % 32 or t.shape[-1] % 32:
                # five stride-2 encoder stages and x2 decoder upsampling with skip additions
                raise _lib.EmsaError(
                    f"batch['{name}'] has shape {tuple(t.shape)}: height and width must be "
                    "multiples of 32 (encoder downsampling 32, decoder skip connections)")
            if not t.is_cuda:
                raise _lib.EmsaError(
                    f"batch['{name}'] lives on {t.device}: the EMSANet engine only runs on an "
                    "AMD GPU (no CPU fallback)")
        if self.compute_dtype == torch.float16 and self.training:
            raise _lib.EmsaError("float16 storage is an inference mode (no loss scaling); train "
                                 "in bfloat16 or float32")
        if 'rgbd' in self.args.input_modalities:
            if feeds['rgb'].shape[0] != feeds['depth'].shape[0] or \
                    feeds['rgb'].shape[2:] != feeds['depth'].shape[2:]:
                raise _lib.EmsaError("batch['rgb'] and batch['depth'] differ in batch size or "
                                     "resolution")
            feeds = {'rgbd': torch.cat([feeds['rgb'], feeds['depth']], dim=1)}
        self._pack_plan.refresh(self.compute_dtype)
        if self.training and self._seed_dev is not None and \
                not torch.cuda.is_current_stream_capturing():
            self._sync_dropout_state()

        if self.training:
            first = next(iter(feeds.values()))
            self._prepare_dropout_masks(first.shape[0], first.device)
        plan = self._cut_plan if self.training else None
        if plan is not None:
            plan.begin()
        deep, skips = self.encoder(feeds, plan)
        if plan is not None:
            # encoder / decoder boundary: everything the context module and the decoders read is a
            # detached leaf; the originals become roots of the encoder's backward segments
            D = plan.DECODERS
            deep_raw = deep
            deep = {k: plan.cut(v, st, D) for k, (v, st) in deep.items()}
            # every skip stream a decoder READS ('add-rgb' by default; 'add-depth' / 'add-rgbd' per
            # decoder, emsanet/decoder.py:63-91) is cut into a leaf whose gradient flows back into
            # its encoder segment; a stream no decoder reads is just detached (its gradient comes
            # through the fusion modules only)
            read = self._skip_streams_read()
            # (which of these leaves belong to the SECOND decoder segment of a decoder_cut plan: the
            #  deep features and the skip the first decoder modules add, the one of the largest ds)
            first_ds = str(max(int(ds) for ds in skips)) if skips else None
            plan.late_stages = {st for _, (_, st) in deep_raw.items()} | \
                ({st for _, (_, st) in skips[first_ds].items()} if first_ds is not None else set())
            skips = {ds: {k: (plan.cut(v, st, D) if k in read or len(sk) == 1 else v.detach())
                          for k, (v, st) in sk.items()} for ds, sk in skips.items()}
        # the context module sees the fused rgb stream, or the only stream there is
        ctx_in = deep['rgb'] if len(feeds) == 2 else next(iter(deep.values()))
        ctx, ctx_branches = self.context_module(ctx_in)

        results = self._run_decoders((ctx, ctx_branches), skips, batch, do_postprocessing)
        if self.training:
            self._advance_dropout_step()
            if self._drop_plan is not None:
                for layer, _, _ in self._drop_plan[2]:
                    layer._premade = None        # (a mask view is good for ONE forward pass)
        if not do_postprocessing:
            return results
        merged = {}
        for r in reversed(results):          # earlier decoders win on duplicate keys
            merged.update(r)
        if not self.training:
            # inference with the un-resized frames in the batch: `<key>_fullres` predictions for the
            # reference's consumers (inference_samples.py:153-163, inference_dataset.py:223-520)
            from .postprocessing import add_fullres_predictions, fullres_shape
            hw = fullres_shape(batch)
            if hw is not None:
                add_fullres_predictions(merged, hw)
        return merged
