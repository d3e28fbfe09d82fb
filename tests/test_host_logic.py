"""CPU tests of the host side: state-dict layout, argument handling, and a dry run of the whole
forward/backward orchestration with the C-ABI replaced by a recording stub (no arithmetic):
checks that every parameter receives a gradient of its own shape, that saved tensors are
released, and which kernels a training step launches."""
import collections
import types

import pytest
import torch


class _FakeLib:
    def __init__(self):
        self.calls = collections.Counter()

    def __getattr__(self, name):
        if not name.startswith('emsa_'):
            raise AttributeError(name)

        def fn(*a):
            self.calls[name] += 1
            if name in ('emsa_conv_stats_rows', 'emsa_conv1d_wino_stats_rows', 'emsa_bn_bwd_rows'):
                return 3
            if name in ('emsa_channel_ws_floats', 'emsa_bn_finalize_ws_bytes',
                        'emsa_ce_semantic_blocks', 'emsa_instance_loss_blocks',
                        'emsa_center_ws_entries'):
                return 64
            if name == 'emsa_center_candidates_max':
                return 1024
            return 0
        return fn


@pytest.fixture
def fake_lib(monkeypatch):
    from emsanet_amd import _lib, functional
    fake = _FakeLib()
    monkeypatch.setattr(_lib, '_lib', fake)
    monkeypatch.setattr(functional, '_stream', lambda: 0)
    return fake


def _model(args):
    from emsanet_amd import nyuv2_config
    from emsanet_amd.model import EMSANet
    return EMSANet(args, nyuv2_config())


def _bypass_device_check(model):
    # the product refuses CPU tensors; the dry run feeds CPU tensors to the stubbed library
    import emsanet_amd.model as M

    class _T(torch.Tensor):
        pass
    orig = M.EMSANet.forward

    def fwd(self, batch, do_postprocessing=False):
        class B(dict):
            pass
        return orig(self, {k: _Cuda(v) for k, v in batch.items()}, do_postprocessing)
    return fwd


class _Cuda:
    """wraps a CPU tensor and claims is_cuda (only the attributes the stem touches)"""

    def __init__(self, t):
        self.t = t
        self.is_cuda = True
        self.shape = t.shape
        self.device = t.device

    def dim(self):
        return self.t.dim()

    def detach(self):
        return self.t.detach()

    def float(self):
        return self.t.float()


def test_state_dict_matches_oracle_layout():
    from emsanet_amd import full_args, nyuv2_config
    from emsanet_amd.model import EMSANet
    from oracle.emsanet_oracle import EMSANetOracle
    args = full_args()
    m, o = EMSANet(args, nyuv2_config()), EMSANetOracle(args, nyuv2_config())
    sm, so = m.state_dict(), o.state_dict()
    assert list(sm.keys()) == list(so.keys())
    assert all(sm[k].shape == so[k].shape for k in sm)
    assert sum(p.numel() for p in m.parameters()) == 63501474
    # key fragments pinned by /root/reference/emsanet/weights.py:22-26,39-56,82-119
    keys = list(sm.keys())
    assert any(k.startswith('encoder.') for k in keys)
    assert any(k.startswith('context_module.') for k in keys)
    sc = [k for k in keys if 'instance_decoder' in k and 'head' in k and 'shared_conv' in k]
    assert any('norm.num_batches_tracked' in k for k in sc)
    assert sm['decoders.instance_decoder.head.shared_conv.conv.weight'].shape[0] == 96
    assert sm['decoders.instance_decoder.head.task_convs.2.weight'].shape == (2, 32, 3, 3)
    up = [k for k in keys if 'instance_decoder' in k and 'head' in k and 'upsampling' in k]
    assert up and all(sm[k].shape[0] == 5 for k in up)
    sem = [k for k in keys if 'semantic_decoder' in k and 'head' in k and 'conv' in k]
    assert sem and all(sm[k].shape[0] == 40 for k in sem)
    scene = [k for k in keys if 'scene_decoder' in k and 'head' in k]
    assert scene and all(sm[k].shape[0] == 10 for k in scene)


def test_reference_init_rules():
    from emsanet_amd import full_args
    from emsanet_amd.nn import NonBottleneck1D
    m = _model(full_args())
    # zero_residual_initialization of the decoders (/root/reference/emsanet/model.py:188-190)
    for mod in m.decoders.modules():
        if isinstance(mod, NonBottleneck1D):
            assert float(mod.bn2.weight.detach().abs().sum()) == 0.0
    for mod in m.encoder.modules():
        if isinstance(mod, NonBottleneck1D):
            assert float(mod.bn2.weight.detach().abs().sum()) > 0.0
    # learned upsampling starts as the bilinear kernel
    w = m.decoders['semantic_decoder'].decoder_modules[0].upsampling.conv.weight
    assert torch.allclose(w[0, 0], torch.tensor([[1., 2, 1], [2, 4, 2], [1, 2, 1]]) / 16)


def test_unsupported_configurations_raise():
    from emsanet_amd import default_args, full_args
    with pytest.raises(NotImplementedError):
        _model(full_args(rgb_encoder_backbone='resnet34se'))
    with pytest.raises(NotImplementedError):
        _model(full_args(rgb_encoder_backbone_resnet_block='bottleneck-se'))
    with pytest.raises(NotImplementedError):
        _model(full_args(semantic_decoder='segformermlp'))
    with pytest.raises(NotImplementedError):
        _model(full_args(instance_offset_encoding='bogus'))
    # arguments the reference hands on to its factories and the engine has ONE implementation for:
    # refused loudly, never silently replaced (round 6: `upsampling_context_module` used to be ignored)
    with pytest.raises(NotImplementedError):
        _model(full_args(encoder_normalization='layernorm'))
    with pytest.raises(NotImplementedError):
        _model(full_args(semantic_decoder_downsamplings=(8, 4, 2)))
    with pytest.raises(NotImplementedError):
        _model(full_args(upsampling_context_module='bicubic'))
    with pytest.raises(NotImplementedError):
        _model(full_args(activation='swish'))
    assert _model(full_args(upsampling_context_module='nearest')).context_module.upsampling == 'nearest'
    with pytest.raises(NotImplementedError, match='n_channels'):
        _model(full_args(semantic_decoder_n_channels=(512, 256)))
    with pytest.raises(NotImplementedError, match='skip_downsamplings'):
        _model(full_args(encoder_decoder_skip_downsamplings=(4, 8)))
    # decoder blocks (args.py:325-331,401-407): NBt1D (default) and basic block built, the x4 bottleneck refused
    with pytest.raises(NotImplementedError, match='decoder block'):
        _model(full_args(semantic_decoder_block='bottleneck'))
    from emsanet_amd.nn import BasicBlock
    assert isinstance(_model(full_args(instance_decoder_block='basicblock'))
                      .decoders['instance_decoder'].decoder_modules[0].blocks[0], BasicBlock)
    # decoder / prediction up-sampling (args.py:280-298,363-372): the learned one and the two weight-free
    # modes are built, the library's 'learned-3x3' (padding rule unknown) is refused by name
    for field in ('semantic_decoder_upsampling', 'instance_decoder_upsampling', 'upsampling_prediction'):
        with pytest.raises(NotImplementedError, match='learned-3x3'):
            _model(full_args(**{field: 'learned-3x3'}))
    from emsanet_amd.nn import PlainUpsampling
    m = _model(full_args(semantic_decoder_upsampling='bilinear', upsampling_prediction='nearest'))
    d = m.decoders['semantic_decoder']
    assert isinstance(d.decoder_modules[0].upsampling, PlainUpsampling) and d.decoder_modules[0].upsampling.mode == 'bilinear'
    assert [u.mode for u in d.head.upsampling] == ['nearest', 'nearest']
    assert not isinstance(m.decoders['instance_decoder'].decoder_modules[0].upsampling, PlainUpsampling)
    with pytest.raises(KeyError):
        default_args(not_a_field=1)
    a = default_args(input_modalities=('rgb',))
    assert a.encoder_fusion == 'none'       # /root/reference/emsanet/args.py:1317-1321


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from emsanet_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.EmsaError):
        _lib.lib()
    from emsanet_amd import full_args
    with pytest.raises(_lib.EmsaError):
        _model(full_args())


def test_rejected_call_raises():
    from emsanet_amd import _lib
    with pytest.raises(_lib.EmsaError):
        _lib.check(-1, 'emsa_conv_igemm')


@pytest.mark.parametrize('train', [True, False])
def test_dry_run_orchestration(fake_lib, train, monkeypatch):
    import emsanet_amd.model as M
    from emsanet_amd import full_args, functional as Fn
    from oracle.emsanet_oracle import synthetic_batch
    monkeypatch.setattr(Fn, 'BN1_FOLD', True)     # (default rule: tensors >= 24 MiB only)
    args = full_args(input_height=64, input_width=96)
    model = _model(args)
    model.train(train)
    monkeypatch.setattr(M.EMSANet, 'forward', _bypass_device_check(model))
    batch = synthetic_batch(2, 64, 96)
    outs = model(batch)
    assert isinstance(outs, list) and len(outs) == 3
    sem, inst, scene = outs
    assert sem[0].shape == (2, 40, 64, 96)
    assert [t.shape[1] for t in inst[0]] == [1, 2, 2] and inst[0][0].shape[2:] == (64, 96)
    assert scene[0].shape == (2, 10)
    if train:
        assert [s.shape for s in sem[1]] == [(2, 40, 2, 3), (2, 40, 4, 6), (2, 40, 8, 12)]
        assert len(inst[1]) == 3 and len(inst[1][0]) == 3
    else:
        assert sem[1] == () and inst[1] == ()
    flat = [sem[0], *inst[0], scene[0]] + list(sem[1]) + [t for s in inst[1] for t in s]
    torch.autograd.backward(flat, [torch.zeros_like(t) for t in flat])
    for k, p in model.named_parameters():
        if not train and 'side_output_heads' in k:
            continue      # side heads are evaluated in training mode only
        assert p.grad is not None and p.grad.shape == p.shape, k
    c = fake_lib.calls
    # 2 encoders x 16 + 2 decoders x 9 NBt1D blocks, 4 MFMA convs each (+3 downsample per encoder)
    assert c['emsa_conv_wgrad'] + c['emsa_conv_wgrad_inbn'] >= 50 * 4
    wino = c['emsa_conv1d_wino'] + c['emsa_conv1d_wino_bnb'] + c['emsa_conv1d_wino_inbn']
    assert c['emsa_conv_igemm'] + wino > c['emsa_conv_wgrad'] + c['emsa_conv_wgrad_inbn']
    # every stride-1 3x1/1x3 conv runs on the Winograd kernel, forward and data gradient.  fp32
    # training (batch statistics): bn1 of every NBt1D block lives in the loaders of conv3x1_2
    # (forward + weight gradient) and in the epilogue of its data gradient -- no separate
    # normalise pass, no separate reduction pass; with frozen statistics nothing is folded
    folded = 50 if train else 0
    assert c['emsa_conv1d_wino_inbn'] == folded and c['emsa_conv_wgrad_inbn'] == folded
    assert c['emsa_conv1d_wino_bnb'] == folded and c['emsa_bn_bwd_apply_rows_t'] == folded
    assert wino >= 2 * (50 * 4 - 2 * 3 * 2) - 2
    assert c['emsa_se_pair_fwd_t'] == 5 and c['emsa_se_mlp_fwd'] == 0 and c['emsa_maxpool3x3s2_fwd'] == 2
    assert c['emsa_up2x_dw3x3_fwd'] == 2 * 3 + 2 * 2
    # merged dict variant (do_postprocessing=True), /root/reference/emsanet/model.py:230-231
    d = model(batch, do_postprocessing=True)
    assert isinstance(d, dict) and 'semantic_output' in d and 'instance_centers' in d


def test_dry_run_fused_bn_reduction(fake_lib, monkeypatch):
    """with the fused form forced on, the data gradient of every NBt1D block's conv3x1_2 carries
    bn1's backward reduction and bn1's backward is the apply-from-rows entry point"""
    import emsanet_amd.model as M
    from emsanet_amd import full_args, functional as Fn
    from oracle.emsanet_oracle import synthetic_batch
    monkeypatch.setattr(Fn, '_BN_FUSE_ENV', '1')
    monkeypatch.setattr(Fn, 'BN1_FOLD', False)       # (the folded forward implies the fused form)
    model = _model(full_args(input_height=64, input_width=96)).train()
    monkeypatch.setattr(M.EMSANet, 'forward', _bypass_device_check(model))
    outs = model(synthetic_batch(2, 64, 96))
    flat = [t for o, sides in outs for t in (list(o) if isinstance(o, tuple) else [o])]
    flat += [t for _, sides in outs for s in sides for t in (list(s) if isinstance(s, tuple) else [s])]
    torch.autograd.backward(flat, [torch.zeros_like(t) for t in flat])
    c = fake_lib.calls
    assert c['emsa_conv1d_wino_bnb'] == 50
    assert c['emsa_bn_bwd_apply_rows_t'] == 50
    assert c['emsa_conv1d_wino_inbn'] == 0 and c['emsa_conv_wgrad_inbn'] == 0
    assert all(p.grad is not None for p in model.parameters())


def test_dry_run_fast_eval_uses_folded_kernels(fake_lib, monkeypatch):
    import emsanet_amd.model as M
    from emsanet_amd import full_args
    from oracle.emsanet_oracle import synthetic_batch
    model = _model(full_args(input_height=64, input_width=96)).eval()
    monkeypatch.setattr(M.EMSANet, 'forward', _bypass_device_check(model))
    with torch.no_grad():
        model(synthetic_batch(1, 64, 96))
    c = fake_lib.calls
    assert c['emsa_bn_finalize'] == 0 and c['emsa_bn_fold'] > 0
    # NBt1D blocks: exactly 4 conv launches each, BatchNorm folded, no separate bn_act pass
    assert c['emsa_bn_act_fwd'] == 0


@pytest.mark.parametrize('panoptic', [False, True])
def test_dry_run_twin_launch_orchestration(fake_lib, monkeypatch, panoptic):
    """16-bit eval fast path on the stubbed C-ABI: the rgb | depth encoder blocks and the semantic |
    instance decoder modules (also inside a PanopticHelper) issue ONE twin launch per conv pair and per
    up-sampling pair (the stub takes no geometry on the register-stationary kernel, so every pair goes
    to emsa_conv_igemm_pair_t); EMSA_TWIN=0 / fp32 storage / a batch beyond nn.TWIN_MAX_PIXELS launch
    per module as before"""
    import emsanet_amd.model as M
    from emsanet_amd import full_args, nn as enn
    from oracle.emsanet_oracle import synthetic_batch
    model = _model(full_args(input_height=64, input_width=96, compute_dtype='bfloat16',
                             enable_panoptic=panoptic)).eval()
    monkeypatch.setattr(M.EMSANet, 'forward', _bypass_device_check(model))
    c = fake_lib.calls
    with torch.no_grad():
        model(synthetic_batch(1, 64, 96))
    # encoders: 16 blocks x 4 convs + 3 down-sampling convs; decoders: 3 x (3x3 + 3 blocks x 4 + skip 1x1)
    n_pairs = c['emsa_conv_igemm_pair_t']
    assert n_pairs == (16 * 4 + 3) + 3 * (1 + 12 + 1), n_pairs
    assert c['emsa_up2x_dw3x3_fwd_pair_t'] == 3
    assert c['emsa_conv1d_rs_pair_t'] == 0 and c['emsa_conv1d_rs_t'] == 0
    singles = c['emsa_conv_igemm_t'] + c['emsa_conv_igemm_splitk_t']
    for key in ('emsa_conv_igemm_pair_t', 'emsa_up2x_dw3x3_fwd_pair_t'):
        c[key] = 0
    monkeypatch.setattr(enn, 'TWIN', False)
    before = c['emsa_conv_igemm_t'] + c['emsa_conv_igemm_splitk_t']
    with torch.no_grad():
        model(synthetic_batch(1, 64, 96))
    assert c['emsa_conv_igemm_pair_t'] == 0 and c['emsa_up2x_dw3x3_fwd_pair_t'] == 0
    assert c['emsa_conv_igemm_t'] + c['emsa_conv_igemm_splitk_t'] - before == singles + 2 * n_pairs
    # default rule: a batch with more input pixels than nn.TWIN_MAX_PIXELS keeps one launch per module
    monkeypatch.setattr(enn, 'TWIN', None)
    monkeypatch.setattr(enn, 'TWIN_MAX_PIXELS', 64 * 96)
    with torch.no_grad():
        model(synthetic_batch(2, 64, 96))
    assert c['emsa_conv_igemm_pair_t'] == 0
    with torch.no_grad():
        model(synthetic_batch(1, 64, 96))
    assert c['emsa_conv_igemm_pair_t'] == n_pairs


def test_dry_run_training_losses(fake_lib, monkeypatch):
    """TrainingLosses on the engine's raw training outputs: every supervised scale reaches its
    loss kernel, the weighted total back-propagates into every parameter (stubbed C-ABI)"""
    import emsanet_amd.model as M
    from emsanet_amd import full_args
    from emsanet_amd.loss import TrainingLosses
    from oracle.emsanet_oracle import synthetic_batch
    args = full_args(input_height=64, input_width=96, tasks_weighting=(1.0, 0.25, 3.0, 0.5))
    model = _model(args).train()
    monkeypatch.setattr(M.EMSANet, 'forward', _bypass_device_check(model))
    outs = model(synthetic_batch(2, 64, 96))
    sizes = [(64, 96), (2, 3), (4, 6), (8, 12)]
    targets = {'semantic': [torch.ones(2, h, w, dtype=torch.long) for h, w in sizes],
               'instance': [dict(center=torch.zeros(2, 1, h, w), offset=torch.zeros(2, 2, h, w),
                                 foreground=torch.ones(2, h, w, dtype=torch.bool),
                                 orientation=torch.zeros(2, h, w),
                                 orientation_foreground=torch.ones(2, h, w, dtype=torch.bool))
                            for h, w in sizes],
               'scene': torch.ones(2, dtype=torch.long)}
    crit = TrainingLosses(args, torch.ones(40), 10)
    total, losses = crit(outs, targets)
    assert set(losses) == {'semantic', 'scene', 'instance_center', 'instance_offset',
                           'instance_orientation'}
    c = fake_lib.calls
    assert c['emsa_ce_semantic_fwd'] == 4 + 1 and c['emsa_instance_loss_fwd'] == 4
    total.backward()
    assert c['emsa_ce_semantic_bwd'] == 5 and c['emsa_instance_loss_bwd'] == 4
    for k, p in model.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, k


def test_input_size_must_be_multiple_of_32():
    from emsanet_amd import _lib, full_args
    model = _model(full_args(input_height=64, input_width=96))
    with pytest.raises(_lib.EmsaError, match='multiples of 32'):
        model({'rgb': torch.zeros(1, 3, 72, 104), 'depth': torch.zeros(1, 1, 72, 104)})


def test_dry_run_panoptic_helper(fake_lib, monkeypatch):
    """--enable-panoptic (/root/reference/emsanet/decoder.py:141-158): both decoders under
    `decoders.panoptic_helper`, nested raw outputs, losses still reach every parameter"""
    import emsanet_amd.model as M
    from emsanet_amd import full_args
    from emsanet_amd.loss import TrainingLosses
    from oracle.emsanet_oracle import synthetic_batch
    args = full_args(input_height=64, input_width=96, enable_panoptic=True)
    model = _model(args).train()
    keys = list(model.state_dict())
    assert any(k.startswith('decoders.panoptic_helper.semantic_decoder.') for k in keys)
    assert any(k.startswith('decoders.panoptic_helper.instance_decoder.') for k in keys)
    assert not any(k.startswith('decoders.semantic_decoder.') for k in keys)
    assert list(model.decoders.keys()) == ['panoptic_helper', 'scene_decoder']
    monkeypatch.setattr(M.EMSANet, 'forward', _bypass_device_check(model))
    outs = model(synthetic_batch(2, 64, 96))
    (sem, inst), (sem_side, inst_side) = outs[0]
    assert sem.shape == (2, 40, 64, 96) and len(inst) == 3 and len(sem_side) == 3
    sizes = [(64, 96), (2, 3), (4, 6), (8, 12)]
    targets = {'semantic': [torch.ones(2, h, w, dtype=torch.long) for h, w in sizes],
               'instance': [dict(center=torch.zeros(2, 1, h, w), offset=torch.zeros(2, 2, h, w),
                                 foreground=torch.ones(2, h, w, dtype=torch.bool),
                                 orientation=torch.zeros(2, h, w),
                                 orientation_foreground=torch.ones(2, h, w, dtype=torch.bool))
                            for h, w in sizes],
               'scene': torch.ones(2, dtype=torch.long)}
    total, losses = TrainingLosses(args, torch.ones(40), 10)(outs, targets)
    total.backward()
    for k, p in model.named_parameters():
        assert p.grad is not None, k
    # the surgery case "semantic-only model fed a panoptic checkpoint" still applies to these keys
    from emsanet_amd.weights import load_weights        # noqa: F401


def test_he_init_parts_and_bias_defaults():
    """/root/reference/emsanet/model.py:162-185 + args.py:626-638: every whitelisted part is
    He-initialised (weights only), unknown parts raise"""
    from emsanet_amd import full_args
    torch.manual_seed(0)
    base = _model(full_args(he_init=()))
    torch.manual_seed(0)
    m = _model(full_args(he_init=('encoder-fusion', 'encoder-decoder-fusion', 'context-module',
                                  'decoder')))
    sb, sm = base.state_dict(), m.state_dict()

    def changed(frag, kind):
        ks = [k for k in sm if frag in k and k.endswith(kind) and sm[k].dim() > 1]
        assert ks, frag
        return [not torch.equal(sm[k], sb[k]) for k in ks]
    assert all(changed('fusion_modules', 'weight'))
    assert all(changed('skip_fusion.conv', 'weight'))
    assert all(changed('context_module', 'conv.weight'))
    assert all(changed('decoder_modules.0.conv3x3.conv', 'weight'))
    # biases keep PyTorch's default; the learned upsampling keeps its bilinear kernel
    for k in sm:
        if k.endswith('.bias') and 'fusion_modules' in k:
            assert torch.equal(sm[k], sb[k]), k
        if 'upsampling' in k:
            assert torch.equal(sm[k], sb[k]), k
        if k.startswith('encoder.backbone'):
            assert torch.equal(sm[k], sb[k]), k
    with pytest.raises(ValueError):
        _model(full_args(he_init=('bogus',)))


def test_spec_switches_change_the_structure_like_the_oracle(monkeypatch):
    """the five [U] switches that used to be hard-coded in the engine are table-driven on both
    sides: after a flip the state dicts still agree key for key"""
    from emsanet_amd import full_args, nn as enn, nyuv2_config
    from emsanet_amd.model import EMSANet
    from oracle import emsanet_oracle as O
    flips = dict(STEM_BIAS=True, DW_UPSAMPLE_BIAS=False, SIDE_OUTPUT_KERNEL=3,
                 SKIP_FUSION_1X1='always', ORIENTATION_L2_NORMALIZE=True)
    for name, val in flips.items():
        monkeypatch.setattr(enn.Spec, name, val)
        monkeypatch.setattr(O.Spec, name, val)
        args = full_args(input_height=64, input_width=64)
        if name == 'SKIP_FUSION_1X1':
            args.semantic_decoder_n_channels = (256, 128, 64)       # == skip channels
            args.instance_decoder_n_channels = (256, 128, 64)
        m, o = EMSANet(args, nyuv2_config()), O.EMSANetOracle(args, nyuv2_config())
        sm, so = m.state_dict(), o.state_dict()
        assert list(sm) == list(so), name
        assert all(sm[k].shape == so[k].shape for k in sm), name
        if name == 'STEM_BIAS':
            assert 'encoder.backbone_rgb.conv1.bias' in sm
        if name == 'DW_UPSAMPLE_BIAS':
            assert not any('upsampling' in k and k.endswith('bias') for k in sm)
        if name == 'SIDE_OUTPUT_KERNEL':
            assert sm['decoders.semantic_decoder.side_output_heads.0.conv.weight'].shape[-1] == 3
        if name == 'SKIP_FUSION_1X1':
            assert 'decoders.semantic_decoder.decoder_modules.0.skip_fusion.conv.weight' in sm
        monkeypatch.undo()


@pytest.mark.parametrize('cfg', [
    (3, 4, (3, 1), (2, 1), (1, 0), 2, 15, 6), (3, 4, (1, 3), (1, 2), (0, 1), 1, 7, 13),
    (3, 4, (1, 1), (2, 2), (0, 0), 1, 5, 9), (2, 3, (3, 3), (2, 2), (1, 1), 1, 9, 11),
    (2, 2, (1, 3), (1, 3), (0, 1), 1, 3, 11), (2, 2, (3, 3), (2, 2), (1, 1), 1, 4, 4),
    (2, 2, (3, 1), (1, 1), (1, 0), 1, 5, 4)])
def test_strided_dgrad_phase_plan(cfg):
    """host logic of the phase-decomposed strided data gradient (functional._dgrad_phases): every
    output pixel belongs to exactly one phase, and evaluating each phase as the dense stride-1
    convolution of dy it describes (tap subset, row / column offsets) reproduces autograd's dx"""
    import torch.nn.functional as F
    from emsanet_amd import functional as Fn
    cin, cout, k, s, p, n, h, w = cfg
    torch.manual_seed(0)
    x = torch.randn(n, cin, h, w, dtype=torch.double, requires_grad=True)
    wt = torch.randn(cout, cin, *k, dtype=torch.double)
    y = F.conv2d(x, wt, None, stride=s, padding=p)
    dy = torch.randn_like(y)
    y.backward(dy)
    spec = Fn.ConvSpec(cin, cout, k, s, p)
    oh, ow = y.shape[2:]
    dx = torch.full((n, cin, h, w), float('nan'), dtype=torch.double)
    for ph, pw, khs, kws, off_h, off_w, rows, cols in Fn._dgrad_phases(spec, h, w):
        for j in range(rows):
            for i in range(cols):
                acc = torch.zeros(n, cin, dtype=torch.double)
                for th, kh in enumerate(khs):
                    for tw, kw in enumerate(kws):
                        r, c = j + off_h - th, i + off_w - tw
                        if 0 <= r < oh and 0 <= c < ow:
                            acc += dy[:, :, r, c] @ wt[:, :, kh, kw]
                assert torch.isnan(dx[:, :, s[0] * j + ph, s[1] * i + pw]).all()      # written once
                dx[:, :, s[0] * j + ph, s[1] * i + pw] = acc
    assert not torch.isnan(dx).any()                                                   # covered
    assert float((dx - x.grad).abs().max()) < 1e-12


def test_bn1_fold_default_rule(monkeypatch):
    """bn1 is folded into the conv loaders where the saved passes outweigh the loader work: the
    bs=32 640x480 stages /4, /8, /16 (157 / 79 / 39 MB), not /32 (20 MB), never in 16-bit storage"""
    from emsanet_amd import functional as Fn
    monkeypatch.setattr(Fn, 'BN1_FOLD', None)
    monkeypatch.setattr(Fn, '_BN1_FOLD_ENV', None)
    t = lambda c, h, w, dt=torch.float32: torch.empty(32, c, h, w, device='meta', dtype=dt)   # noqa: E731
    assert Fn.bn1_fold(t(64, 120, 160)) and Fn.bn1_fold(t(128, 60, 80)) and Fn.bn1_fold(t(256, 30, 40))
    assert not Fn.bn1_fold(t(512, 15, 20))
    assert not Fn.bn1_fold(t(64, 120, 160, torch.bfloat16))
    monkeypatch.setattr(Fn, '_BN1_FOLD_ENV', '0')
    assert not Fn.bn1_fold(t(64, 120, 160))


def test_segment_parameter_groups_partition_the_model():
    """the backward segments of SegmentedGraphedTrainStep: every parameter in exactly one group,
    decoders + context module first, the stem in the last one; buckets never straddle two groups
    and the tail bucket of the LAST group stays small"""
    from emsanet_amd import full_args
    from emsanet_amd.graph import segment_parameter_groups
    from emsanet_amd.parallel import GradientBuckets
    model = _model(full_args(input_height=64, input_width=96))
    params = [p for p in model.parameters() if p.requires_grad]
    groups = segment_parameter_groups(model, (2, 1))
    assert len(groups) == 4
    ids = [id(p) for g in groups for p in g]
    assert len(ids) == len(set(ids)) == len(params)
    names = {id(p): n for n, p in model.named_parameters()}
    assert all(names[id(p)].startswith(('decoders.', 'context_module.')) for p in groups[0])
    assert all('layer3' in names[id(p)] or 'layer4' in names[id(p)] or 'fusion_modules.3' in names[id(p)]
               or 'fusion_modules.4' in names[id(p)] for p in groups[1])
    assert any(names[id(p)].endswith('backbone_rgb.conv1.weight') for p in groups[3])
    b = GradientBuckets(params, bucket_bytes=8 << 20, groups=groups, manual=True, tail_bytes=1 << 20)
    assert len(b.group_buckets) == 4 and sum(len(g) for g in b.group_buckets) == len(b.buckets)
    for gi, bis in enumerate(b.group_buckets):
        members = {id(p) for bi in bis for p in b.buckets[bi][1]}
        assert members == {id(p) for p in groups[gi]}
    last = b.buckets[b.group_buckets[-1][-1]][0]
    assert last.numel() * 4 <= 1 << 20
    # arrival order + tail without groups: order respected, every parameter once
    order = list(reversed(params))
    b2 = GradientBuckets(params, bucket_bytes=8 << 20, order=order, tail_bytes=2 << 20)
    flat = [p for _, ps, _ in b2.buckets for p in ps]
    assert [id(p) for p in flat] == [id(p) for p in order]
    assert b2.buckets[-1][0].numel() * 4 <= 2 << 20
    with pytest.raises(ValueError):
        GradientBuckets(params, groups=groups[:-1])
    # decoder_cut: the decoder group in two -- later modules + heads | first modules, first side heads,
    # context module, scene head -- still a partition
    g5 = segment_parameter_groups(model, (2, 1), decoder_cut=True)
    assert len(g5) == 5 and [id(p) for g in g5[2:] for p in g] == [id(p) for g in groups[1:] for p in g]
    assert {id(p) for p in g5[0]} | {id(p) for p in g5[1]} == {id(p) for p in groups[0]}
    assert not ({id(p) for p in g5[0]} & {id(p) for p in g5[1]})
    n0, n1 = [names[id(p)] for p in g5[0]], [names[id(p)] for p in g5[1]]
    assert all('decoder_modules.0.' not in n and 'side_output_heads.0.' not in n and
               not n.startswith(('context_module.', 'decoders.scene_decoder.')) for n in n0)
    assert any(n.startswith('context_module.') for n in n1) and any('decoder_modules.0.' in n for n in n1)
    assert any(n.startswith('decoders.scene_decoder.') for n in n1)
    assert any('.head.' in n for n in n0) and any('decoder_modules.2.' in n for n in n0)


def test_cut_plan_dry_run(fake_lib, monkeypatch):
    """forward with autograd cuts (segmented backward): the decoders and the encoder stages behind a
    cut run on detached leaves; records carry (original, leaf, producing stage, group)"""
    import emsanet_amd.model as M
    from emsanet_amd import full_args
    from emsanet_amd.nn import CutPlan
    from oracle.emsanet_oracle import synthetic_batch
    model = _model(full_args(input_height=64, input_width=96)).train()
    monkeypatch.setattr(M.EMSANet, 'forward', _bypass_device_check(model))
    plan = CutPlan((2, 1))
    model._cut_plan = plan
    outs = model(synthetic_batch(2, 64, 96))
    model._cut_plan = None
    groups = [g for _, _, _, g in plan.records]
    assert groups.count(1) == 2 and groups.count(2) == 2          # rgb + depth behind each cut
    dec = [(o, c, st) for o, c, st, g in plan.records if g == CutPlan.DECODERS]
    assert sorted(st for _, _, st in dec) == [1, 2, 3, 4, 4]      # skips /4 /8 /16, deep rgb + depth
    for o, c, _, _ in plan.records:
        assert c.is_leaf and c.requires_grad and c.data_ptr() == o.data_ptr()
    flat = [t for o, sides in outs for t in (list(o) if isinstance(o, tuple) else [o])]
    # backward from the outputs stops at the decoder leaves: no encoder parameter gets a gradient
    torch.autograd.backward(flat, [torch.zeros_like(t) for t in flat])
    got = {n for n, p in model.named_parameters() if p.grad is not None}
    assert got and all(n.startswith(('decoders.', 'context_module.')) for n in got)
    with pytest.raises(ValueError):
        CutPlan((4,))
    # decoder_cut: one more leaf per dense decoder behind its first module; the deep features and
    # the /16 skip (what the first modules and the context module read) are marked "late"
    plan = CutPlan((2, 1), decoder_cut=True)
    model._cut_plan = plan
    model(synthetic_batch(2, 64, 96))
    model._cut_plan = None
    assert [g for _, _, _, g in plan.records].count(CutPlan.DECODER_MID) == 2
    assert plan.late_stages == {3, 4}


def test_pretrained_backbone_argument_is_honoured(tmp_path):
    """`no_pretrained_backbone=False` (the reference's default, /root/reference/emsanet/args.py:119-123;
    handed to `get_backbone` at model.py:58-59,72-73): with a weights file the backbones carry its
    tensors (a 3-channel stem summed for the depth backbone), without one the constructor refuses --
    it used to build a randomly initialised model without a word"""
    from emsanet_amd import full_args
    from emsanet_amd.nn import ResNetNBt1D
    with pytest.raises(NotImplementedError, match='pretrained'):
        _model(full_args(no_pretrained_backbone=False))
    torch.manual_seed(3)
    src = ResNetNBt1D('resnet34', 3, 0.1)
    sd = {('module.' + k): v.clone() for k, v in src.state_dict().items()}
    sd['module.fc.weight'] = torch.zeros(1000, 512)
    fp = str(tmp_path / 'r34_nbt1d.pth')
    torch.save({'state_dict': sd}, fp)
    m = _model(full_args(no_pretrained_backbone=False,
                         rgb_encoder_backbone_pretrained_weights_filepath=fp,
                         depth_encoder_backbone_pretrained_weights_filepath=fp))
    own = src.state_dict()
    for k, v in m.encoder.backbone_rgb.state_dict().items():
        assert torch.equal(v, own[k]), k
    d = m.encoder.backbone_depth.state_dict()
    assert torch.equal(d['conv1.weight'], own['conv1.weight'].sum(1, keepdim=True))
    assert torch.equal(d['layer3.2.conv1x3_2.weight'], own['layer3.2.conv1x3_2.weight'])
    bad = str(tmp_path / 'bad.pth')
    torch.save({k: v for k, v in list(sd.items())[:10]}, bad)
    with pytest.raises(RuntimeError, match='missing'):
        _model(full_args(no_pretrained_backbone=False,
                         rgb_encoder_backbone_pretrained_weights_filepath=bad,
                         depth_encoder_backbone_pretrained_weights_filepath=bad))


def test_bottleneck_resnet50_layout():
    """`--*-encoder-backbone resnet50 --*-encoder-backbone-resnet-block bottleneck`
    (/root/reference/inference_time.bash:8,13; emsanet/tests/test_interface_model.py:133): torchvision's
    layout -- (3, 4, 6, 3) blocks of 1x1 / 3x3 (stride) / 1x1 with expansion 4, a 1x1 down-sampling
    branch in the first block of every stage -- so torchvision-named backbone weights load key for key;
    stage widths 256 .. 2048 reach the fusion modules, the skip connections and the context module"""
    from emsanet_amd import full_args
    m = _model(full_args(rgb_encoder_backbone='resnet50', depth_encoder_backbone='resnet50',
                         rgb_encoder_backbone_resnet_block='bottleneck',
                         depth_encoder_backbone_resnet_block='bottleneck'))
    bb = m.encoder.backbone_rgb
    assert [len(getattr(bb, f'layer{i}')) for i in (1, 2, 3, 4)] == [3, 4, 6, 3]
    assert bb.stage_channels == (64, 256, 512, 1024, 2048)
    b0, b1 = bb.layer2[0], bb.layer2[1]
    assert tuple(b0.conv1.weight.shape) == (128, 256, 1, 1) and b0.conv2.stride == (2, 2)
    assert tuple(b0.conv3.weight.shape) == (512, 128, 1, 1) and b0.downsample[0].stride == (2, 2)
    assert b1.downsample is None and bb.layer1[0].downsample is not None
    keys = set(bb.state_dict())
    for k in ('conv1.weight', 'bn1.running_var', 'layer1.0.conv3.weight', 'layer1.0.downsample.1.weight',
              'layer3.5.bn3.bias', 'layer4.2.conv2.weight'):
        assert k in keys, k
    sd = m.state_dict()
    assert tuple(sd['encoder.fusion_modules.4.se_rgb.fc.0.weight'].shape) == (128, 2048, 1, 1)
    assert m.decoders['semantic_decoder'].decoder_modules[0].conv3x3.conv.in_channels == 2048
    assert m.decoders['semantic_decoder'].decoder_modules[0].skip_fusion.conv.in_channels == 1024
