# -*- coding: utf-8 -*-
"""
`EMSANet(args, dataset_config)` -- the MI355X-native drop-in for the reference's
`emsanet.model.EMSANet` (/root/reference/emsanet/model.py:26-233): same constructor, same
`forward(batch, do_postprocessing=False)` contract (list of per-decoder `(outputs, side_outputs)`
tuples, or one merged dict), same attributes callers touch (`encoder`, `context_module`,
`decoders`, `state_dict()`), see SURVEY.md §8(b).  All arithmetic runs in libemsanet_hip.so.
"""
from collections import ChainMap
from typing import Any, Dict

import torch
import torch.nn as nn

from . import _lib
from . import ops
from .decoder import get_decoders
from .nn import (Dropout2dHash, FusedEncoder, NonBottleneck1D, PyramidPoolingModule, ResNetNBt1D)


class EMSANet(nn.Module):
    def __init__(self, args, dataset_config) -> None:
        super().__init__()
        _lib.lib()     # fail loudly at construction if the HIP extension is missing

        # store args and dataset parameters (model.py:34-36)
        self.args = args
        self.dataset_config = dataset_config

        # dataset properties (model.py:39-43)
        semantic_labels = dataset_config.semantic_label_list_without_void
        semantic_n_classes = len(semantic_labels)
        scene_n_classes = len(dataset_config.scene_label_list_without_void)
        panoptic_semantic_classes_is_thing = semantic_labels.classes_is_thing
        panoptic_use_orientation = tuple(semantic_labels.classes_use_orientations)

        # encoders (model.py:46-92)
        def backbone(name, block, n_in):
            if block != 'nonbottleneck1d':
                raise NotImplementedError(f"resnet block '{block}' (hot path is NBt1D)")
            return ResNetNBt1D(name, n_in, args.dropout_p)

        if 'rgbd' in args.input_modalities:
            raise NotImplementedError("single rgbd encoder is not part of the hot path")
        backbone_rgb = backbone(args.rgb_encoder_backbone,
                                args.rgb_encoder_backbone_resnet_block, 3) \
            if 'rgb' in args.input_modalities else None
        backbone_depth = backbone(args.depth_encoder_backbone,
                                  args.depth_encoder_backbone_resnet_block, 1) \
            if 'depth' in args.input_modalities else None
        if getattr(args, 'activation', 'relu') != 'relu':
            raise NotImplementedError("only the default 'relu' activation has kernels")

        # fused encoder (model.py:95-106)
        self.encoder = FusedEncoder(backbone_rgb, backbone_depth, args.encoder_fusion,
                                    args.encoder_decoder_skip_downsamplings)
        enc_downsampling = self.encoder.downsampling
        enc_n_channels_out = self.encoder.n_channels_out
        enc_skips_n_channels = self.encoder.skips_n_channels

        # context module (model.py:109-119)
        if args.context_module != 'ppm':
            raise NotImplementedError(f"context module '{args.context_module}'")
        self.context_module = PyramidPoolingModule(
            enc_n_channels_out, enc_n_channels_out,
            (args.input_height // enc_downsampling, args.input_width // enc_downsampling))

        # decoders (model.py:122-160)
        if args.instance_offset_encoding == 'tanh':
            instance_normalized_offset, instance_tanh_for_offset = True, True
        elif args.instance_offset_encoding == 'relative':
            instance_normalized_offset, instance_tanh_for_offset = True, False
        elif args.instance_offset_encoding == 'deeplab':
            instance_normalized_offset, instance_tanh_for_offset = False, False
        else:
            raise NotImplementedError
        instance_sigmoid_for_center = args.instance_center_encoding == 'sigmoid'

        self.decoders = get_decoders(
            args,
            n_channels_in=enc_n_channels_out,
            downsampling_in=enc_downsampling,
            semantic_n_classes=semantic_n_classes,
            instance_normalized_offset=instance_normalized_offset,
            instance_offset_distance_threshold=args.instance_offset_distance_threshold,
            instance_sigmoid_for_center=instance_sigmoid_for_center,
            instance_tanh_for_offset=instance_tanh_for_offset,
            normal_n_channels_out=3,
            scene_n_channels_in=self.context_module.n_channels_reduction,
            scene_n_classes=scene_n_classes,
            panoptic_semantic_classes_is_thing=panoptic_semantic_classes_is_thing,
            panoptic_has_orientation=panoptic_use_orientation,
            fusion_n_channels=enc_skips_n_channels[::-1],
        )

        # initialisation (model.py:162-190)
        if 'encoder-fusion' in args.he_init and self.encoder.two:
            for m in self.encoder.fusion_modules.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                    if m.bias is not None:
                        nn.init.zeros_(m.bias)
        if not args.no_zero_init_decoder_residuals:
            for m in self.decoders.modules():
                if isinstance(m, NonBottleneck1D):
                    nn.init.zeros_(m.bn2.weight)

        # Dropout2d bookkeeping (counter-based masks, one id per dropout layer)
        self.dropout_seed = 0
        self.dropout_step = 0
        lid = 0
        for m in self.modules():
            if isinstance(m, Dropout2dHash):
                m.layer_id = lid
                m.seed_fn = self._dropout_seed
                lid += 1

        # every convolution's weight transforms (packed / Winograd layouts) in one launch per step
        rts = []
        for m in self.modules():
            rt = getattr(m, '_rt', None)
            if isinstance(rt, ops.NBt1DRT):
                rts += [rt.c31_1, rt.c13_1, rt.c31_2, rt.c13_2] + ([rt.cds] if rt.cds else [])
            crt = getattr(m, '_crt', None)
            if isinstance(crt, ops.ConvRT):
                rts.append(crt)
        self._pack_plan = ops.PackPlan(rts)

    def _dropout_seed(self):
        return (self.dropout_seed + 0x632BE5AB * self.dropout_step) & 0xFFFFFFFF

    def forward(self, batch, do_postprocessing=False) -> Dict[str, Any]:
        # determine input (model.py:194-204)
        enc_inputs = {}
        if 'rgb' in self.args.input_modalities:
            enc_inputs['rgb'] = batch['rgb']
        if 'depth' in self.args.input_modalities:
            enc_inputs['depth'] = batch['depth']
        for k, v in enc_inputs.items():
            if v.dim() != 4 or v.shape[-2] % 32 or v.shape[-1] % 32:
                # five stride-2 encoder stages and x2 decoder upsampling with skip additions
                raise _lib.EmsaError(
                    f"batch['{k}'] has shape {tuple(v.shape)}: height and width must be multiples "
                    "of 32 (encoder downsampling 32, decoder skip connections)")
            if not v.is_cuda:
                raise _lib.EmsaError(
                    f"batch['{k}'] lives on {v.device}: the EMSANet engine only runs on an AMD "
                    "GPU (no CPU fallback)")

        self._pack_plan.refresh()

        # forward (fused) encoder(s) (model.py:206)
        enc_outputs, enc_dec_skips = self.encoder(enc_inputs)

        # context module input (model.py:209-217)
        if len(self.args.input_modalities) == 2:
            con_input = enc_outputs['rgb']
        else:
            assert len(enc_inputs) == 1
            con_input = enc_outputs[list(enc_inputs.keys())[0]]
        con_outputs, con_context_outputs = self.context_module(con_input)

        # decoders (model.py:220-227)
        outputs = []
        for decoder in self.decoders.values():
            outputs.append(decoder((con_outputs, con_context_outputs), enc_dec_skips, batch,
                                   do_postprocessing=do_postprocessing))
        if self.training:
            self.dropout_step += 1

        # simplify output if postprocessing was applied (model.py:230-231)
        if do_postprocessing:
            outputs = dict(ChainMap(*outputs))
        return outputs
