"""CPU: checkpoint surgery of emsanet_amd.weights.load_weights, one test per
case of /root/reference/emsanet/weights.py:11-162 on synthetic state dicts."""
import copy
from argparse import Namespace

import pytest
import torch


def _model(tasks, n_sem=40, n_scene=10):
    from emsanet_amd import DatasetConfig, full_args
    from emsanet_amd.model import EMSANet
    args = full_args(input_height=64, input_width=64, tasks=tasks)
    args.dataset = 'nyuv2'
    return EMSANet(args, DatasetConfig(n_sem, n_scene)), args


def _rand_sd(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {k: (torch.randn(v.shape, generator=g) if v.is_floating_point() else v.clone())
            for k, v in model.state_dict().items()}


def test_plain_round_trip_and_renamed_prefix():
    from emsanet_amd.weights import load_weights
    m, args = _model(('semantic', 'scene', 'instance', 'orientation'))
    sd = _rand_sd(m)
    old = {k.replace('encoder.', 'fused_encoders.', 1) if k.startswith('encoder.') else k: v
           for k, v in sd.items()}
    load_weights(args, m, old, verbose=False)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_orientation_removed_from_instance_head():
    from emsanet_amd.weights import load_weights
    full, _ = _model(('semantic', 'scene', 'instance', 'orientation'))
    sd = _rand_sd(full)
    m, args = _model(('semantic', 'scene', 'instance'))
    load_weights(args, m, copy.deepcopy(sd), verbose=False)
    own = m.state_dict()
    k = 'decoders.instance_decoder.head.shared_conv.conv.weight'
    assert own[k].shape[0] == 64 and torch.equal(own[k], sd[k][:64])
    k = 'decoders.instance_decoder.head.shared_conv.norm.running_var'
    assert torch.equal(own[k], sd[k][:64])
    k = 'decoders.instance_decoder.head.upsampling.1.conv.weight'
    assert own[k].shape[0] == 3 and torch.equal(own[k], sd[k][:3])
    assert not any('task_convs.2' in key for key in own)
    k = 'decoders.instance_decoder.head.task_convs.1.weight'
    assert torch.equal(own[k], sd[k])


def test_semantic_only_from_panoptic_checkpoint_and_extra_keys_dropped():
    from emsanet_amd.weights import load_weights
    full, _ = _model(('semantic', 'scene', 'instance', 'orientation'))
    sd = _rand_sd(full)
    pan = {}
    for k, v in sd.items():
        if k.startswith('decoders.semantic_decoder.'):
            pan[k.replace('decoders.semantic_decoder.',
                          'decoders.panoptic_helper.semantic_decoder.')] = v
        elif k.startswith('decoders.instance_decoder.'):
            pan[k.replace('decoders.instance_decoder.',
                          'decoders.panoptic_helper.instance_decoder.')] = v
        else:
            pan[k] = v
    m, args = _model(('semantic',))
    load_weights(args, m, pan, verbose=False)
    k = 'decoders.semantic_decoder.head.conv.weight'
    assert torch.equal(m.state_dict()[k], sd[k])
    assert not any('instance_decoder' in key or 'scene_decoder' in key for key in m.state_dict())


def test_scene_class_mismatch_keeps_model_weights():
    from emsanet_amd.weights import load_weights
    src, _ = _model(('semantic', 'scene'), n_scene=7)
    sd = _rand_sd(src)
    m, args = _model(('semantic', 'scene'), n_scene=10)
    before = m.state_dict()['decoders.scene_decoder.head.weight'].clone()
    load_weights(args, m, sd, verbose=False)
    assert torch.equal(m.state_dict()['decoders.scene_decoder.head.weight'], before)
    k = 'context_module.final_conv.conv.weight'
    assert torch.equal(m.state_dict()[k], sd[k])


def test_semantic_37_40_classes():
    from emsanet_amd.weights import load_weights
    s37, _ = _model(('semantic',), n_sem=37)
    sd37 = _rand_sd(s37)
    m40, args = _model(('semantic',), n_sem=40)
    own_before = copy.deepcopy(m40.state_dict())
    load_weights(args, m40, copy.deepcopy(sd37), verbose=False)
    k = 'decoders.semantic_decoder.head.conv.weight'
    assert torch.equal(m40.state_dict()[k][:37], sd37[k])
    assert torch.equal(m40.state_dict()[k][37:], own_before[k][37:])
    # side heads (no 'head' fragment) and the 37-channel upsampling keep the model's weights
    k = 'decoders.semantic_decoder.head.upsampling.0.conv.weight'
    assert m40.state_dict()[k].shape[0] == 40
    # 40 -> 37 (sunrgbd)
    s40, _ = _model(('semantic',), n_sem=40)
    sd40 = _rand_sd(s40, 1)
    m37, args37 = _model(('semantic',), n_sem=37)
    args37.dataset = 'sunrgbd'
    load_weights(args37, m37, copy.deepcopy(sd40), verbose=False)
    k = 'decoders.semantic_decoder.head.conv.bias'
    assert torch.equal(m37.state_dict()[k], sd40[k][:37])
