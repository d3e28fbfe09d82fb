# -*- coding: utf-8 -*-
"""
Hot-path subset of the reference's argument namespace.

`emsanet.model.EMSANet(args, dataset_config)` takes the argparse namespace produced by
`ArgParserEMSANet` (/root/reference/emsanet/args.py:41-1488) as its model configuration
(/root/reference/emsanet/model.py:27-36).  The fields read on the hot path are listed in
SURVEY.md §8(b); `default_args()` returns a namespace carrying exactly those fields with the
reference's defaults (cited per field), so a namespace coming from the reference's own parser
is accepted unchanged and a caller without the reference can still build the model.
"""
from argparse import Namespace

# field -> (default, line of the flag in /root/reference/emsanet/args.py)
_DEFAULTS = {
    'tasks': (('semantic',), 60),
    'enable_panoptic': (False, 68),
    'input_height': (480, 78),
    'input_width': (640, 84),
    'input_modalities': (('rgb', 'depth'), 90),
    'activation': ('relu', 109),
    'no_pretrained_backbone': (False, 119),
    'encoder_normalization': ('batchnorm', 126),
    'encoder_fusion': ('se-add-uni-rgb', 143),
    'rgb_encoder_backbone': ('resnet34', 152),
    'rgb_encoder_backbone_resnet_block': ('nonbottleneck1d', 159),
    'rgb_encoder_backbone_pretrained_weights_filepath': (None, 174),
    'depth_encoder_backbone': ('resnet34', 185),
    'depth_encoder_backbone_resnet_block': ('nonbottleneck1d', 192),
    'depth_encoder_backbone_pretrained_weights_filepath': (None, 207),
    'rgbd_encoder_backbone': ('resnet34', 218),
    'rgbd_encoder_backbone_resnet_block': ('nonbottleneck1d', 225),
    'rgbd_encoder_backbone_pretrained_weights_filepath': (None, 232),
    'context_module': ('ppm', 244),
    'upsampling_context_module': ('bilinear', 251),
    'encoder_decoder_skip_downsamplings': ((4, 8, 16), 261),
    'upsampling_prediction': ('learned-3x3-zeropad', 290),
    'decoder_normalization': ('batchnorm', 300),
    'semantic_encoder_decoder_fusion': ('add-rgb', 310),
    'semantic_decoder': ('emsanet', 318),
    'semantic_decoder_block': ('nonbottleneck1d', 325),
    'semantic_decoder_block_dropout_p': (0.2, 332),
    'semantic_decoder_n_blocks': (3, 339),
    'semantic_decoder_dropout_p': (0.1, 346),
    'semantic_decoder_n_channels': ((512, 256, 128), 353),
    'semantic_decoder_downsamplings': ((16, 8, 4), 364),
    'semantic_decoder_upsampling': ('learned-3x3-zeropad', 373),
    'instance_encoder_decoder_fusion': ('add-rgb', 386),
    'instance_decoder': ('emsanet', 394),
    'instance_decoder_block': ('nonbottleneck1d', 401),
    'instance_decoder_block_dropout_p': (0.2, 408),
    'instance_decoder_n_blocks': (3, 415),
    'instance_decoder_dropout_p': (0.1, 422),
    'instance_decoder_n_channels': ((512, 256, 128), 429),
    'instance_decoder_downsamplings': ((16, 8, 4), 440),
    'instance_decoder_upsampling': ('learned-3x3-zeropad', 449),
    'normal_encoder_decoder_fusion': ('add-rgb', 543),
    'normal_decoder': ('emsanet', 551),
    'normal_decoder_block': ('nonbottleneck1d', 558),
    'normal_decoder_block_dropout_p': (0.2, 565),
    'normal_decoder_n_blocks': (3, 572),
    'normal_decoder_dropout_p': (0.1, 579),
    'normal_decoder_n_channels': ((512, 256, 128), 586),
    'normal_decoder_downsamplings': ((16, 8, 4), 597),
    'normal_decoder_upsampling': ('learned-3x3-zeropad', 606),
    'instance_center_heatmap_threshold': (0.1, 469),
    'instance_center_heatmap_nms_kernel_size': (17, 478),
    'instance_center_heatmap_apply_foreground_mask': (False, 487),
    'instance_center_heatmap_top_k': (64, 499),
    'instance_offset_encoding': ('tanh', 516),
    'instance_center_encoding': ('sigmoid', 506),
    'instance_offset_distance_threshold': (None, 528),
    'dropout_p': (0.1, 619),
    # training losses; tasks_weighting None -> (1,)*len(tasks), args.py:1346-1348
    'tasks_weighting': (None, 696),
    'semantic_loss_label_smoothing': (0.0, 724),
    'semantic_no_multiscale_supervision': (False, 731),
    'instance_weighting': ((2, 1), 740),
    'instance_center_loss': ('mse', 750),
    'instance_no_multiscale_supervision': (False, 757),
    'orientation_kappa': (1.0, 766),
    'scene_loss_label_smoothing': (0.1, 791),
    # optimizer / schedule
    'optimizer': ('sgd', 661),
    'learning_rate': (0.01, 668),
    'learning_rate_scheduler': ('onecycle', 676),
    'momentum': (0.9, 684),
    'weight_decay': (1e-4, 690),
    'n_epochs': (500, 649),
    'he_init': (('encoder-fusion',), 626),
    'no_zero_init_decoder_residuals': (False, 640),
    'debug': (False, 1116),
    # NOT a reference option (line 0): storage type of the engine's activations, see
    # EMSANet.set_compute_dtype ('float32' = the reference's arithmetic)
    'compute_dtype': ('float32', 0),
}

FULL_TASKS = ('semantic', 'scene', 'instance', 'orientation')


def default_args(**overrides) -> Namespace:
    """Namespace with the reference defaults for every field the hot path reads."""
    ns = Namespace(**{k: v for k, (v, _line) in _DEFAULTS.items()})
    for k, v in overrides.items():
        if k not in _DEFAULTS:
            raise KeyError(f"unknown hot-path argument '{k}'")
        setattr(ns, k, v)
    # post-parse fix-up the reference applies (args.py:1317-1321)
    if len(ns.input_modalities) == 1:
        ns.encoder_fusion = 'none'
    # default task weighting (args.py:1346-1353)
    if ns.tasks_weighting is None:
        ns.tasks_weighting = (1,) * len(ns.tasks)
    if len(ns.tasks_weighting) != len(ns.tasks):
        raise ValueError("Length for given task weighting does not match number of tasks: "
                         f"{len(ns.tasks_weighting)} vs. {len(ns.tasks)}.")
    return ns


def full_args(**overrides) -> Namespace:
    """BASELINE.json config 2: RGB-D, all four task heads."""
    overrides.setdefault('tasks', FULL_TASKS)
    overrides.setdefault('no_pretrained_backbone', True)
    return default_args(**overrides)
