"""Drop-in boundary, driven the way the reference's own interface tests drive it:

  * `get_decoders(args, ...)` stand-alone with exactly the tensors of
    /root/reference/emsanet/tests/test_interface_decoders.py:36-131 -- contiguous NCHW `torch.rand`
    features (bs 3, 480x640 -> /32), the `(B, 256, 1, 1)` GAP branch, the skip dict with both
    modalities -- for train / eval x do_postprocessing, and the `(outputs, side_outputs)` contract;
  * `EMSANet(args, dataset_config)(batch, do_postprocessing)` with the inputs of
    /root/reference/emsanet/tests/test_interface_model.py:53-101 (bs 3, `torch.randn`) over the
    supported task / modality / backbone grid;
  * foreign layouts cost ONE pass of the library's layout kernel (`emsa_to_nhwc_t`), never a torch
    `permute().contiguous()`: inputs and cotangents in contiguous NCHW (what NCHW losses return),
    expanded scalars (what `.sum().backward()` produces), sliced views -- values bit-identical to
    the channels-last path.
"""
import pytest
import torch

from util import DEV, rnd, to_act

pytestmark = pytest.mark.gpu

H, W, DS = 480, 640, 32


def _decoder_inputs(seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.rand(*s, generator=g).to(DEV)      # noqa: E731  contiguous NCHW, like the reference
    x = (r(3, 512, H // DS, W // DS), (r(3, 512 // 2, 1, 1),))
    skips = {str(d): {'rgb': r(3, c, H // d, W // d), 'depth': r(3, c, H // d, W // d)}
             for d, c in ((16, 256), (8, 128), (4, 64))}
    batch = {'instance_foreground': torch.ones((3, H, W), dtype=torch.bool, device=DEV),
             'instance': torch.ones((3, H, W), dtype=torch.bool, device=DEV),
             'orientation_foreground': torch.ones((3, H, W), dtype=torch.bool, device=DEV)}
    return x, skips, batch


def _flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, dict):
        return [t for v in o.values() for t in _flat(v)]
    if isinstance(o, (list, tuple)):
        return [t for v in o for t in _flat(v)]
    return []


@pytest.mark.parametrize('tasks,panoptic', [(('semantic',), False),
                                            (('semantic', 'instance'), False),
                                            (('semantic', 'instance', 'orientation', 'scene'), False),
                                            (('semantic', 'instance', 'orientation', 'scene'), True),
                                            (('semantic', 'instance', 'orientation', 'scene', 'normal'),
                                             False)])
@pytest.mark.parametrize('training', [False, True])
@pytest.mark.parametrize('do_postprocessing', [False, True])
def test_get_decoders_standalone_reference_inputs(tasks, panoptic, training, do_postprocessing):
    from emsanet_amd import default_args
    from emsanet_amd.decoder import get_decoders
    args = default_args(tasks=tasks, enable_panoptic=panoptic, input_height=H, input_width=W)
    torch.manual_seed(1)
    # the reference's call (test_interface_decoders.py:40-50), extra keywords included
    decoders = get_decoders(args, n_channels_in=512, downsampling_in=32, semantic_n_blocks=3,
                            instance_n_blocks=2, normal_n_blocks=1, scene_n_channels_in=512 // 2,
                            fusion_n_channels=(256, 128, 64), debug=True).to(DEV)
    n_decoders = len(tasks) - ('orientation' in tasks) - (1 if panoptic else 0)
    assert len(decoders) == n_decoders
    decoders.train(training)
    x, skips, batch = _decoder_inputs()
    with torch.set_grad_enabled(training):
        outs = [d(x, skips, batch, do_postprocessing=do_postprocessing) for d in decoders.values()]
        # the same VALUES handed over as channels-last tensors: identical results, i.e. the
        # layout kernel at the boundary moves every element to the right place
        x_cl = (to_act(x[0].cpu()), (to_act(x[1][0].cpu()),))
        skips_cl = {k: {m: to_act(t.cpu()) for m, t in v.items()} for k, v in skips.items()}
        # (dropout masks are a function of (seed, step, layer), not of the call count)
        outs_cl = [d(x_cl, skips_cl, batch, do_postprocessing=do_postprocessing)
                   for d in decoders.values()]
    torch.cuda.synchronize()
    for name, o in zip(decoders.keys(), outs):
        if not do_postprocessing:
            assert isinstance(o, tuple) and len(o) == 2, name
            main, sides = o
            if name == 'scene_decoder':
                assert sides == () and main.shape == (3, 10)
                continue
            # side outputs at /32, /16, /8 in training mode only (test_semantic_loss.py:80-93)
            n_sides = 3 if training else 0
            side_lists = sides if name != 'panoptic_helper' else sides[0]
            assert len(side_lists) == n_sides, (name, len(side_lists))
            if name == 'semantic_decoder':
                assert main.shape == (3, 40, H, W)
                for s, d in zip(sides, (32, 16, 8)):
                    assert s.shape == (3, 40, H // d, W // d)
            if name == 'normal_decoder':
                assert main.shape == (3, 3, H, W)
                for s, d in zip(sides, (32, 16, 8)):
                    assert s.shape == (3, 3, H // d, W // d)
            if name == 'instance_decoder':
                assert [t.shape[1] for t in main] == ([1, 2, 2] if 'orientation' in tasks else [1, 2])
                assert all(t.shape[2:] == (H, W) for t in main)
        else:
            assert isinstance(o, dict) and len(o), name
            if name == 'normal_decoder':
                assert set(o) == {'normal_output', 'normal_side_outputs'}
    a, b = _flat(outs), _flat(outs_cl)
    assert len(a) == len(b) and len(a) > 0
    for t, u in zip(a, b):
        assert t.shape == u.shape and torch.isfinite(t.float()).all()
        if t.is_floating_point():
            assert torch.equal(t, u), "NCHW-contiguous inputs changed the result"


def test_decoder_backward_with_nchw_cotangents():
    """cotangents as a reference-style NCHW loss returns them (contiguous NCHW), as `.sum()` returns
    them (expanded scalar, all strides 0) and as channels-last tensors: same parameter gradients"""
    from emsanet_amd import default_args
    from emsanet_amd.decoder import get_decoders
    args = default_args(tasks=('semantic', 'instance', 'orientation', 'scene'), input_height=H,
                        input_width=W)
    torch.manual_seed(2)
    decoders = get_decoders(args, n_channels_in=512, downsampling_in=32,
                            scene_n_channels_in=256, fusion_n_channels=(256, 128, 64)).to(DEV).train()
    for m in decoders.modules():                      # (decoder blocks start as identities otherwise)
        if isinstance(m, torch.nn.BatchNorm2d):
            torch.nn.init.uniform_(m.weight, 0.5, 1.5)
    x, skips, batch = _decoder_inputs(seed=3)

    def grads(make_cot):
        for p in decoders.parameters():
            p.grad = None
        outs = _flat([d(x, skips, batch) for d in decoders.values()])
        torch.autograd.backward(outs, [make_cot(i, t) for i, t in enumerate(outs)])
        return [p.grad.clone() for p in decoders.parameters()]

    base = [rnd(*s, seed=50 + i, scale=1e-2) for i, s in enumerate(
        [tuple(t.shape) for t in _flat([d(x, skips, batch) for d in decoders.values()])])]
    g_cl = grads(lambda i, t: to_act(base[i]) if base[i].dim() == 4 else base[i].to(DEV))
    g_nchw = grads(lambda i, t: base[i].to(DEV).contiguous())
    gmax = max(float(g.abs().max()) for g in g_cl)
    for a, b in zip(g_cl, g_nchw):
        # (identical cotangent VALUES; a few weight gradients are summed with fp32 atomics)
        assert float((a - b).abs().max()) <= 2e-5 * gmax
    # expanded scalars: d(sum)/dt = 1 everywhere
    g_sum = grads(lambda i, t: torch.ones((), device=DEV).expand(t.shape))
    g_one = grads(lambda i, t: to_act(torch.ones(t.shape)) if t.dim() == 4
                  else torch.ones(t.shape, device=DEV))
    gmax = max(float(g.abs().max()) for g in g_one)
    for a, b in zip(g_sum, g_one):
        assert float((a - b).abs().max()) <= 2e-5 * gmax


@pytest.mark.parametrize('shape', [(3, 40, 37, 53), (2, 1, 64, 64), (1, 5, 8, 8), (2, 512, 15, 20),
                                   (4, 8, 1, 1), (2, 96, 3, 5)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16, torch.float16])
def test_to_nhwc_is_one_library_pass(shape, dtype, monkeypatch):
    """`to_nhwc` on foreign layouts == permute reference bit for bit, through emsa_to_nhwc_t and
    WITHOUT any torch .contiguous() (patched to raise)"""
    from emsanet_amd import _lib, functional as Fn
    calls = []
    real = _lib.lib().emsa_to_nhwc_t

    class Counting:
        def __getattr__(self, name):
            if name == 'emsa_to_nhwc_t':
                def wrapped(*a):
                    calls.append(a)
                    return real(*a)
                return wrapped
            return getattr(_lib._lib, name)
    x = rnd(*shape, seed=4).to(DEV).to(dtype)
    cases = {
        'nchw': x.contiguous(),
        'expanded': x[:, :, :1, :1].expand(shape),
        'sliced': torch.cat([x, x], 1)[:, 1:shape[1] + 1] if shape[1] > 1 else x.contiguous(),
        'transposed': x.transpose(2, 3).contiguous().transpose(2, 3),
    }
    ref = {k: v.permute(0, 2, 3, 1).contiguous() for k, v in cases.items()}
    monkeypatch.setattr(Fn._lib, 'lib', lambda: Counting())

    def boom(*a, **k):
        raise AssertionError("torch .contiguous() on the boundary path")
    monkeypatch.setattr(torch.Tensor, 'contiguous', boom)
    outs = {}
    for k, v in cases.items():
        n_before = len(calls)
        outs[k] = Fn.to_nhwc(v)
        already = Fn.ld_of(outs[k]) == shape[1] and outs[k].data_ptr() == v.data_ptr()
        assert already or len(calls) == n_before + 1, k      # exactly one library pass
    monkeypatch.undo()
    torch.cuda.synchronize()
    for k, o in outs.items():
        assert Fn.ld_of(o) == shape[1]
        assert torch.equal(o.permute(0, 2, 3, 1), ref[k]), k


@pytest.mark.parametrize('tasks', [('semantic',), ('semantic', 'instance'),
                                   ('semantic', 'instance', 'orientation'),
                                   ('semantic', 'instance', 'orientation', 'scene')])
@pytest.mark.parametrize('modalities', [('rgb',), ('depth',), ('rgb', 'depth')])
@pytest.mark.parametrize('training,do_postprocessing', [(True, False), (False, False),
                                                        (False, True), (True, True)])
def test_model_interface_reference_inputs(tasks, modalities, training, do_postprocessing):
    """/root/reference/emsanet/tests/test_interface_model.py:18-101 for the supported grid (ReLU,
    ResNet-18/34 NBt1D): bs 3, 480x640 `torch.randn` inputs, list / dict return contract"""
    from emsanet_amd import default_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    backbone = 'resnet18' if len(tasks) % 2 else 'resnet34'
    args = default_args(
        tasks=tasks, input_modalities=modalities, input_height=H, input_width=W,
        rgb_encoder_backbone=backbone, depth_encoder_backbone=backbone, no_pretrained_backbone=True,
        semantic_encoder_decoder_fusion='add-rgb' if len(modalities) > 1 else 'add',
        instance_encoder_decoder_fusion='add-rgb' if len(modalities) > 1 else 'add')
    torch.manual_seed(0)
    model = EMSANet(args, nyuv2_config()).to(DEV)
    model.train(training)
    bs = 3
    batch = {}
    if 'rgb' in modalities:
        batch['rgb'] = torch.randn((bs, 3, H, W), device=DEV)
    if 'depth' in modalities:
        batch['depth'] = torch.randn((bs, 1, H, W), device=DEV)
    if 'instance' in tasks:
        batch['instance_foreground'] = torch.ones((bs, 1, H, W), dtype=torch.bool, device=DEV)
    if 'orientation' in tasks:
        batch['instance'] = torch.ones((bs, 1, H, W), dtype=torch.bool, device=DEV)
        batch['orientation_foreground'] = torch.ones((bs, 1, H, W), dtype=torch.bool, device=DEV)
    if not training and do_postprocessing:
        for m in modalities:
            batch[f'{m}_fullres'] = batch[m].clone()
    with torch.set_grad_enabled(training):
        outputs = model(batch, do_postprocessing=do_postprocessing)
    torch.cuda.synchronize()
    if do_postprocessing:
        assert isinstance(outputs, dict) and outputs
        assert outputs['semantic_output'].shape == (bs, 40, H, W)
        if not training:
            assert outputs['semantic_segmentation_idx'].shape == (bs, H, W)
            if 'instance' in tasks:
                # the key the reference's consumers read (visualization.py:607-620), grouped inside
                # batch['instance_foreground']
                assert outputs['instance_segmentation_gt_foreground'].shape == (bs, H, W)
                assert len(outputs['instance_segmentation_gt_meta']) == bs          # visualization.py:622-624
            if 'orientation' in tasks:
                # batch['instance'] all ones, (N,1,H,W) bool as the reference's test hands it over: ONE
                # ground-truth instance per image (visualization.py:752-765 zips it with batch['instance'])
                o = outputs['orientations_gt_instance_gt_orientation_foreground']
                assert len(o) == bs and all(list(d) == [1] and 0.0 <= d[1] < 6.2832 for d in o)
    else:
        assert isinstance(outputs, list) and outputs
        assert len(outputs) == len(tasks) - ('orientation' in tasks)
        for o in outputs:
            assert isinstance(o, tuple) and len(o) == 2
    for t in _flat(outputs):
        if t.is_floating_point():
            assert torch.isfinite(t).all()


@pytest.mark.parametrize('training', [True, False])
def test_model_interface_resnet50_bottleneck(training):
    """the `resnet50` case of /root/reference/emsanet/tests/test_interface_model.py:133 (the block its
    ResNet-50 comes with: bottleneck, inference_time.bash:8,13) with the reference test's inputs: bs 3,
    480x640, all four tasks, both modalities -- list contract in train mode (+ a backward pass with
    finite gradients everywhere), merged dict with the post-processed keys in eval mode"""
    from emsanet_amd import default_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    tasks = ('semantic', 'instance', 'orientation', 'scene')
    args = default_args(
        tasks=tasks, input_modalities=('rgb', 'depth'), input_height=H, input_width=W,
        rgb_encoder_backbone='resnet50', depth_encoder_backbone='resnet50',
        rgb_encoder_backbone_resnet_block='bottleneck', depth_encoder_backbone_resnet_block='bottleneck',
        no_pretrained_backbone=True)
    torch.manual_seed(0)
    model = EMSANet(args, nyuv2_config()).to(DEV)
    model.train(training)
    bs = 3
    batch = {'rgb': torch.randn((bs, 3, H, W), device=DEV), 'depth': torch.randn((bs, 1, H, W), device=DEV),
             'instance_foreground': torch.ones((bs, 1, H, W), dtype=torch.bool, device=DEV),
             'instance': torch.ones((bs, 1, H, W), dtype=torch.bool, device=DEV),
             'orientation_foreground': torch.ones((bs, 1, H, W), dtype=torch.bool, device=DEV)}
    if training:
        outputs = model(batch)
        assert isinstance(outputs, list) and len(outputs) == 3
        flat = [t for t in _flat(outputs) if t.is_floating_point()]
        sum((t * t).mean() for t in flat).backward()
        for k, p in model.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
    else:
        batch['rgb_fullres'] = batch['rgb'].clone()
        with torch.no_grad():
            r = model(batch, do_postprocessing=True)
        assert r['semantic_segmentation_idx_fullres'].shape == (bs, H, W)
        assert r['instance_segmentation_gt_foreground'].shape == (bs, H, W)
        assert len(r['orientations_gt_instance_gt_orientation_foreground']) == bs
        assert r['scene_class_idx'].shape == (bs,)
    torch.cuda.synchronize()
