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


def test_scannet_benchmark_class_remapping():
    """case 7 (/root/reference/emsanet/weights.py:121-145): 40 dataset classes -> 20 benchmark
    classes through the mapping that is passed in (void = key 0, ignored class = value 0)"""
    from emsanet_amd.weights import load_weights
    full, _ = _model(('semantic',), n_sem=40)
    sd = _rand_sd(full)
    m, args = _model(('semantic',), n_sem=20)
    args.dataset = 'scannet'
    args.validation_scannet_benchmark_mode = False
    kept = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]
    mapping = {0: 0}
    for c in range(1, 41):
        mapping[c] = kept.index(c) + 1 if c in kept else 0
    load_weights(args, m, copy.deepcopy(sd), verbose=False, scannet_mapping=mapping)
    own = m.state_dict()
    idx = torch.tensor([c - 1 for c in kept])
    n = 0
    for k in own:
        if all(f in k for f in ('semantic_decoder', 'head', 'conv')):
            assert own[k].shape[0] == 20 and torch.equal(own[k], sd[k][idx]), k
            n += 1
    assert n >= 2          # main head weight + bias (+ side heads)
    # without the table the 40-class head cannot be cut down: loud error ...
    m2, _ = _model(('semantic',), n_sem=20)
    with pytest.raises(NotImplementedError):
        load_weights(args, m2, copy.deepcopy(sd), verbose=False)
    # ... but a checkpoint that does not need the table (37 classes into 20) keeps the model's
    # own head like the reference does (weights.py:147-160)
    ck37, _ = _model(('semantic',), n_sem=37)
    sd37 = _rand_sd(ck37, seed=3)
    before = {k: v.clone() for k, v in m2.state_dict().items()}
    load_weights(args, m2, sd37, verbose=False)
    k = 'decoders.semantic_decoder.head.conv.weight'
    assert torch.equal(m2.state_dict()[k], before[k])
    k = 'encoder.backbone_rgb.conv1.weight'
    assert torch.equal(m2.state_dict()[k], sd37[k])
