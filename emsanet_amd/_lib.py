# -*- coding: utf-8 -*-
"""
ctypes binding of libemsanet_hip.so -- exactly the symbols `include/emsanet_hip.h` declares.

The library is built in-tree by `__graft_entry__.build()` (or `make -C emsanet_amd/csrc`) into
`emsanet_amd/lib/libemsanet_hip.so`.  There is NO fallback: if the library is missing or a call
is rejected, this module raises -- the product path never computes on the CPU or through torch
operators instead.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int8, c_int32, c_int64, c_uint32, c_void_p

LIB_PATH = os.environ.get('EMSA_LIB') or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), 'lib', 'libemsanet_hip.so')   # EMSA_LIB: tuning builds


class EmsaPackJob(Structure):
    _fields_ = [('src', c_void_p), ('dst0', c_void_p), ('dst1', c_void_p), ('cout', c_int32),
                ('cin', c_int32), ('kh', c_int32), ('kw', c_int32), ('kind', c_int32),
                ('first_block', c_int32), ('cout_total', c_int32), ('cout_off', c_int32),
                ('cin_total', c_int32), ('cin_off', c_int32)]


class EmsaDropoutJob(Structure):
    _fields_ = [('offset', c_int64), ('c', c_int32), ('layer_id', c_uint32), ('p', c_float),
                ('pad_', c_int32)]


class EmsaConvGeom(Structure):
    _fields_ = [
        ('n_img', c_int32),
        ('in_h', c_int32), ('in_w', c_int32),
        ('out_h', c_int32), ('out_w', c_int32),
        ('k_ch', c_int32), ('n_ch', c_int32),
        ('kh', c_int32), ('kw', c_int32),
        ('mul_h', c_int32), ('off_h', c_int32), ('step_h', c_int32), ('div_h', c_int32),
        ('mul_w', c_int32), ('off_w', c_int32), ('step_w', c_int32), ('div_w', c_int32),
        ('in_img_stride', c_int64), ('in_row_stride', c_int64),
        ('in_px_stride', c_int32), ('ld_out', c_int32),
        # optional output pixel map (0 = dense), see include/emsanet_hip.h
        ('out_pix_img', c_int32), ('out_pix_row', c_int32), ('out_pix_px', c_int32),
        ('out_pix_off', c_int32),
    ]


_P = c_void_p          # device pointers travel as integers (tensor.data_ptr())
_GP = POINTER(EmsaConvGeom)

# name -> (restype, argtypes); must stay in sync with include/emsanet_hip.h (tests check it)
SIGNATURES = {
    'emsa_arch': (c_char_p, []),
    'emsa_version': (c_int, []),
    'emsa_set_batch_invariant': (c_int, [c_int]),
    'emsa_conv_igemm': (c_int, [_GP, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int32, c_int32, _P]),
    'emsa_conv_stats_rows': (c_int, [_GP]),
    'emsa_conv1d_wino_supported': (c_int, [_GP]),
    'emsa_conv1d_wino_stats_rows': (c_int, [_GP]),
    'emsa_conv_relu_bits_words': (c_int64, [c_int64, c_int32]),
    'emsa_conv1d_wino_bnb': (c_int, [_GP, _P, _P, _P, _P, c_int32, _P, c_int32, _P, _P, _P, _P, _P,
                                     c_int32, _P]),
    'emsa_conv_igemm_bnb_t': (c_int, [c_int32, _GP, _P, _P, _P, _P, c_int32, _P, c_int32, _P, _P, _P, _P,
                                      _P, c_int32, _P]),
    'emsa_bn_bwd_apply_rows_t': (c_int, [c_int32, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64,
                                         c_int32, c_int32, _P, _P, _P, _P]),
    'emsa_conv1d_wino': (c_int, [_GP, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int32, c_int32, _P,
                                 _P, _P]),
    'emsa_memset_async': (c_int, [_P, c_int32, c_int64, _P]),
    'emsa_graph_count_nodes': (c_int, [_P, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    'emsa_graph_replace_memsets': (c_int, [_P, POINTER(c_int32)]),
    'emsa_dropout2d_mask_batch': (c_int, [_P, _P, c_int32, c_int32, c_int32, c_uint32, _P, _P]),
    'emsa_to_nhwc_t': (c_int, [c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int64, c_int64,
                               c_int64, c_int64, _P]),
    'emsa_conv1d_wino_inbn': (c_int, [_GP, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, _P]),
    'emsa_conv_wgrad_inbn': (c_int, [_GP, _P, _P, _P, _P, _P, _P, _P, _P]),
    'emsa_conv_wgrad_inbn_t': (c_int, [c_int32, _GP, _P, _P, _P, _P, _P, _P, _P, _P]),
    'emsa_conv_wgrad_multi_inbn_t': (c_int, [c_int32, c_int32, _GP, POINTER(c_void_p), POINTER(c_void_p),
                                             POINTER(c_void_p), POINTER(c_void_p), _P,
                                             POINTER(c_void_p), POINTER(c_void_p), _P]),
    'emsa_conv1d_rs_inbn_t': (c_int, [c_int32, _GP, _P, _P, _P, _P, _P, _P, _P, c_int32, _P]),
    'emsa_bn_bwd_reduce_aff_t': (c_int, [c_int32, _P, _P, _P, _P, _P, _P, c_int32, c_int64, c_int32, _P, _P]),
    'emsa_bn_bwd_apply_aff_t': (c_int, [c_int32, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64,
                                        c_int32, _P, _P, _P, _P]),
    'emsa_conv_wgrad_multi_ws_bytes': (c_int64, [c_int32, c_int32, _GP]),
    'emsa_conv_wgrad_multi_t': (c_int, [c_int32, c_int32, _GP, POINTER(c_void_p), POINTER(c_void_p),
                                        POINTER(c_void_p), POINTER(c_void_p), _P, _P]),
    'emsa_pack_wino': (c_int, [_P, _P, _P, c_int32, c_int32, c_int32, _P]),
    'emsa_pack_batch': (c_int, [_P, c_int32, c_int32, _P]),
    'emsa_pack_wino_packed': (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_conv_wgrad_ws_bytes': (c_int64, [_GP]),
    'emsa_conv_wgrad': (c_int, [_GP, _P, _P, _P, _P, _P, _P]),
    'emsa_pack_weight_fwd': (c_int, [_P, _P] + [c_int32] * 8 + [_P]),
    'emsa_pack_weight_dgrad': (c_int, [_P, _P] + [c_int32] * 8 + [_P]),
    'emsa_unpack_wgrad': (c_int, [_P, _P] + [c_int32] * 8 + [_P]),
    'emsa_pack_weight_pair': (c_int, [_P, _P, _P] + [c_int32] * 4 + [_P]),
    'emsa_stem_pack_input': (c_int, [_P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_stem_pack_weight': (c_int, [_P, _P, c_int32, c_int32, _P]),
    'emsa_stem_pack_input_rows_t': (c_int, [c_int32, _P, _P, c_int32, c_int32, c_int32, _P]),
    'emsa_stem_pack_weight_rows_t': (c_int, [c_int32, _P, _P, c_int32, _P]),
    'emsa_stem_unpack_wgrad_rows': (c_int, [_P, _P, c_int32, _P]),
    'emsa_stem_unpack_wgrad': (c_int, [_P, _P, c_int32, c_int32, _P]),
    'emsa_bn_finalize': (c_int, [_P, c_int32, c_int32, c_int64, _P, _P, c_float, c_float,
                                 _P, _P, _P, _P, _P, _P, _P, _P]),
    'emsa_bn_finalize_ws_bytes': (c_int, [c_int32]),
    'emsa_bn_fold': (c_int, [_P, _P, _P, _P, c_float, c_int32, _P, _P, _P, _P]),
    'emsa_relu_mask_words': (c_int64, [c_int64]),
    'emsa_bn_act_fwd': (c_int, [_P, _P, _P, _P, _P, _P, c_int32, c_int64, c_int32, c_int32, _P,
                                _P]),
    'emsa_bn_bwd_reduce': (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int32, c_int64, c_int32, c_int32,
                                   _P, _P]),
    'emsa_bn_bwd_rows': (c_int, [c_int64, c_int32]),
    'emsa_bn_bwd_apply': (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64,
                                  c_int32, c_int32, c_int32, _P, _P, _P, _P, _P]),
    'emsa_dropout2d_mask': (c_int, [_P, c_int32, c_int32, c_float, c_uint32, c_uint32, _P]),
    'emsa_maxpool3x3s2_fwd': (c_int, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_maxpool3x3s2_bwd': (c_int, [_P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_channel_ws_floats': (c_int, [c_int32, c_int64, c_int32]),
    'emsa_channel_mean': (c_int, [_P, _P, _P, c_int32, c_int64, c_int32, _P]),
    'emsa_se_mlp_fwd': (c_int, [_P] * 7 + [c_int32] * 3 + [_P]),
    'emsa_se_mlp_bwd': (c_int, [_P] * 11 + [c_int32] * 3 + [_P]),
    'emsa_se_scale_add_fwd': (c_int, [_P] * 5 + [c_int32, c_int64, c_int32, _P]),
    'emsa_se_scale_bwd_reduce': (c_int, [_P, _P, _P, _P, c_int32, c_int64, c_int32, _P]),
    'emsa_se_scale_bwd_apply': (c_int, [_P] * 5 + [c_int32, c_int64, c_int32, _P]),
    'emsa_up2x_dw3x3_fwd': (c_int, [_P] * 5 + [c_int32] * 4 + [_P]),
    'emsa_up2x_dw3x3_bwd_data': (c_int, [_P] * 3 + [c_int32] * 4 + [_P]),
    'emsa_up2x_dw3x3_bwd_weight': (c_int, [_P] * 4 + [c_int32] * 4 + [_P]),
    'emsa_up2x_dw3x3_bwd': (c_int, [_P] * 6 + [c_int32] * 4 + [_P]),
    'emsa_up2x_dw3x3_bwd_supported': (c_int, [c_int32, c_int32]),
    'emsa_adaptive_avgpool_fwd': (c_int, [_P, _P] + [c_int32] * 5 + [_P]),
    'emsa_adaptive_avgpool_bwd': (c_int, [_P, _P] + [c_int32] * 6 + [_P]),
    'emsa_bilinear_fwd': (c_int, [_P, _P] + [c_int32] * 7 + [_P]),
    'emsa_bilinear_bwd': (c_int, [_P, _P] + [c_int32] * 7 + [_P]),
    'emsa_head_act_fwd': (c_int, [_P, _P, c_int64] + [c_int32] * 5 + [_P]),
    'emsa_head_act_bwd': (c_int, [_P, _P, _P, _P, c_int64] + [c_int32] * 5 + [_P]),
    'emsa_copy_channels': (c_int, [_P, c_int32, _P, c_int32, c_int64, c_int32, _P]),
    'emsa_axpy': (c_int, [_P, _P, c_int64, c_float, _P]),
    'emsa_ce_semantic_blocks': (c_int, [c_int64]),
    'emsa_ce_semantic_fwd': (c_int, [_P, c_int32, _P, _P, c_int32, c_int64, c_float, c_float, _P, _P,
                                     _P]),
    'emsa_ce_semantic_bwd': (c_int, [_P, c_int32, _P, _P, c_int32, c_int64, c_float, c_float, _P, _P,
                                     _P, c_int32, _P]),
    'emsa_softmax_argmax': (c_int, [_P, c_int32, c_int32, c_int64, _P, _P, _P]),
    'emsa_center_candidates_max': (c_int, []),
    'emsa_center_ws_entries': (c_int64, [c_int32, c_int32, c_int32]),
    'emsa_instance_centers': (c_int, [_P, c_int32, c_int32, c_int32, c_int32, c_int32, c_float,
                                      c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    'emsa_instance_assign': (c_int, [_P, c_int32, c_int32, c_int32, c_int32, c_float, c_float, _P,
                                     _P, c_int32, _P, c_float, _P, _P]),
    'emsa_panoptic_merge': (c_int, [_P, _P, _P, c_int32, c_int64, c_int32, c_int32, c_int32, _P, _P,
                                    _P, _P, _P, _P]),
    'emsa_instance_stats': (c_int, [_P, _P, _P, c_int32, c_int64, c_int32, _P, _P, _P]),
    'emsa_panoptic_scores': (c_int, [_P, _P, _P, _P, c_int32, c_int64, c_int32, _P, _P, _P, _P, _P, _P,
                                     _P, _P]),
    'emsa_instance_orientation': (c_int, [_P, c_int32, _P, _P, c_int32, c_int64, c_int32, _P, _P, _P]),
    'emsa_normalize_rgb': (c_int, [_P, _P, c_int32, c_int32, c_int32, c_float, _P, _P, _P]),
    'emsa_normalize_depth': (c_int, [_P, _P, c_int64, c_float, c_float, c_int32, _P]),
    'emsa_sgd_nesterov': (c_int, [_P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_int32,
                                  _P]),
    'emsa_add_t': (c_int, [c_int32, _P, _P, _P, c_int64, _P]),
    'emsa_adam_advance': (c_int, [_P, _P, _P]),
    'emsa_adam_step': (c_int, [_P, _P, _P, _P, c_int64, _P, _P, _P]),
    'emsa_instance_loss_blocks': (c_int, [c_int64]),
    'emsa_instance_loss_fwd': (c_int, [_P, c_int32, _P, c_int32, _P, c_int32, _P, _P, _P, _P, _P, _P,
                                       c_int64, c_float, _P, _P, _P]),
    'emsa_instance_loss_bwd': (c_int, [_P, c_int32, _P, c_int32, _P, c_int32, _P, _P, _P, _P, _P, _P,
                                       c_int64, c_float, _P, _P, _P, c_int32, _P, c_int32, _P,
                                       c_int32, _P]),
    'emsa_bn_act_fwd_t': (c_int, [c_int32, _P, _P, _P, _P, _P, _P, c_int32, c_int64, c_int32, c_int32, _P, _P]),
    'emsa_bn_bwd_reduce_t': (c_int, [c_int32, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int64, c_int32, c_int32, _P, _P]),
    'emsa_bn_bwd_apply_t': (c_int, [c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int64, c_int32, c_int32, c_int32, _P, _P, _P, _P, _P]),
    'emsa_maxpool3x3s2_fwd_t': (c_int, [c_int32, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_maxpool3x3s2_bwd_t': (c_int, [c_int32, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_se_scale_add_fwd_t': (c_int, [c_int32, _P, _P, _P, _P, _P, c_int32, c_int64, c_int32, _P]),
    'emsa_se_scale_bwd_apply_t': (c_int, [c_int32, _P, _P, _P, _P, _P, c_int32, c_int64, c_int32, _P]),
    'emsa_up2x_dw3x3_fwd_t': (c_int, [c_int32, c_int32, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_up2x_dw3x3_bwd_data_t': (c_int, [c_int32, c_int32, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_up2x_dw3x3_bwd_weight_t': (c_int, [c_int32, c_int32, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_up2x_dw3x3_bwd_t': (c_int, [c_int32, c_int32, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_adaptive_avgpool_fwd_t': (c_int, [c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_adaptive_avgpool_bwd_t': (c_int, [c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_bilinear_fwd_t': (c_int, [c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_bilinear_bwd_t': (c_int, [c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_nearest_fwd_t': (c_int, [c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_nearest_bwd_t': (c_int, [c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_head_act_fwd_t': (c_int, [c_int32, c_int32, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_head_act_bwd_gather_t': (c_int, [c_int32, _P, c_int32, c_int32, _P, c_int32, c_int32, _P, c_int32,
                                           c_int32, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32,
                                           c_int32, _P]),
    'emsa_head_act_bwd_t': (c_int, [c_int32, c_int32, _P, _P, _P, _P, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_stem_pack_input_t': (c_int, [c_int32, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    'emsa_channel_mean_t': (c_int, [c_int32, _P, _P, _P, c_int32, c_int64, c_int32, _P]),
    'emsa_se_pair_fwd_t': (c_int, [c_int32] + [_P] * 14 + [c_int32, c_int64, c_int32, c_int32, _P]),
    'emsa_se_scale_bwd_reduce_t': (c_int, [c_int32, _P, _P, _P, _P, c_int32, c_int64, c_int32, _P]),
    'emsa_cast_channels': (c_int, [c_int32, _P, c_int32, c_int32, _P, c_int32, c_int64, c_int32, _P]),
    'emsa_conv_stats_rows_t': (c_int, [c_int32, _GP]),
    'emsa_conv_igemm_t': (c_int, [c_int32, _GP, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int32,
                                  c_int32, _P]),
    'emsa_conv_wgrad_ws_bytes_t': (c_int64, [c_int32, _GP]),
    'emsa_conv_wgrad_t': (c_int, [c_int32, _GP, _P, _P, _P, _P, _P, _P]),
    'emsa_pack_weight_t': (c_int, [c_int32, _P, _P, _P] + [c_int32] * 8 + [_P]),
    'emsa_conv_igemm_splitk_ws_bytes_t': (c_int64, [c_int32, _GP]),
    'emsa_conv_igemm_splitk_t': (c_int, [c_int32, _GP, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P]),
    'emsa_up2x_dw3x3_fwd_pair_t': (c_int, [c_int32] + [_P] * 10 + [c_int32] * 4 + [_P]),
    'emsa_conv_igemm_pair_t': (c_int, [c_int32, _GP] + [_P] * 14 + [c_int32, c_int32, _P, _P, _P]),
    'emsa_conv_rs_set_cu_budget': (c_int, [c_int32]),
    'emsa_nbt_half_block_supported': (c_int, [c_int32, c_int32, c_int32]),
    'emsa_nbt_half_block_t': (c_int, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                      POINTER(c_void_p), c_int32, POINTER(c_void_p), POINTER(c_void_p),
                                      POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                      POINTER(c_void_p), POINTER(c_void_p), c_int32, POINTER(c_void_p),
                                      c_int32, c_int32, _P]),
    'emsa_conv1d_rs_supported': (c_int, [c_int32, _GP]),
    'emsa_conv1d_rs_stats_rows': (c_int, [c_int32, _GP]),
    'emsa_conv1d_rs_t': (c_int, [c_int32, _GP, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int32,
                                 c_int32, _P]),
    'emsa_conv1d_rs_pair_t': (c_int, [c_int32, _GP] + [_P] * 14 + [c_int32, c_int32, _P]),
    'emsa_conv1d_rs_bnb_t': (c_int, [c_int32, _GP, _P, _P, _P, _P, c_int32, _P, c_int32, _P, _P, _P,
                                     _P, _P, c_int32, _P]),
    'emsa_pack_weight_frag_t': (c_int, [c_int32, _P, _P, _P, c_int32, c_int32, _P]),
    'emsa_stem_pack_weight_t': (c_int, [c_int32, _P, _P, c_int32, c_int32, _P]),
    'emsa_dropout2d_mask_dev': (c_int, [_P, c_int32, c_int32, c_float, _P, c_uint32, _P]),
    'emsa_u32_add': (c_int, [_P, c_uint32, _P]),
    'emsa_sgd_nesterov_dev': (c_int, [_P, _P, _P, c_int64, _P, _P]),
    'emsa_prof_enable': (c_int, [c_int32]),
    'emsa_prof_next_flops': (c_int, [ctypes.c_double]),
    'emsa_prof_reset': (c_int, []),
    'emsa_prof_seen': (c_int, [c_int32]),
    'emsa_prof_name': (c_char_p, [c_int32]),
    'emsa_prof_read': (c_int, [c_int32, POINTER(ctypes.c_double), POINTER(ctypes.c_double),
                               POINTER(c_int32)]),
    'emsa_prof_read_bytes': (c_int, [c_int32, POINTER(ctypes.c_double)]),
}

_ERR = {-1: 'EMSA_E_SHAPE (unsupported geometry)', -2: 'EMSA_E_ARG (bad argument)',
        -3: 'EMSA_E_LAUNCH (HIP launch failed)'}


class EmsaError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise EmsaError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or make -C emsanet_amd/csrc). "
            "There is no CPU / torch fallback for the EMSANet engine.")
    # torch first: it ships its own libamdhip64 (same SONAME as /opt/rocm's, which this library
    # is linked against).  Whichever is loaded first serves the whole process, and device memory,
    # streams and kernels must all come from ONE HIP runtime -- torch's, since the tensors are
    # torch's.  (Loading this library before torch made every launch fail with EMSA_E_LAUNCH.)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def check(status, name):
    if status != 0:
        raise EmsaError(f"{name} failed: {_ERR.get(status, status)}")
