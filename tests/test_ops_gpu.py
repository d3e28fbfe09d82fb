"""GPU parity of every C-ABI kernel against a plain PyTorch CPU reference of the same op
(fp64 reference, tolerance 1e-4 relative to the tensor's max magnitude: fp32 kernels with a
different summation order than the reference)."""
import pytest
import torch
import torch.nn.functional as F

from util import DEV, close, rnd, to_act

pytestmark = pytest.mark.gpu


def _fn():
    from emsanet_amd import functional as Fn
    return Fn


CONVS = [
    # cin, cout, kernel, stride, padding, n, h, w
    (64, 64, (3, 1), (1, 1), (1, 0), 2, 12, 20),
    (64, 64, (1, 3), (1, 1), (0, 1), 2, 12, 20),
    (128, 128, (1, 3), (1, 1), (0, 1), 3, 9, 13),
    (64, 128, (3, 1), (2, 1), (1, 0), 2, 12, 20),
    (128, 128, (1, 3), (1, 2), (0, 1), 2, 6, 20),
    (64, 128, (1, 1), (2, 2), (0, 0), 2, 12, 20),
    (256, 128, (3, 3), (1, 1), (1, 1), 2, 8, 10),
    (128, 40, (3, 3), (1, 1), (1, 1), 2, 8, 10),
    (96, 8, (3, 3), (1, 1), (1, 1), 2, 8, 10),
    (512, 256, (1, 1), (1, 1), (0, 0), 2, 5, 5),
    (256, 12, (1, 1), (1, 1), (0, 0), 4, 1, 1),
    (64, 64, (1, 3), (1, 1), (0, 1), 1, 30, 40),
    # ragged: channel counts that are no multiple of the 64-wide tiles / 16-wide K chunks, odd and
    # very short lines, fewer pixels than one K step
    (40, 72, (1, 3), (1, 1), (0, 1), 1, 5, 7),
    (72, 40, (3, 1), (1, 1), (1, 0), 2, 7, 5),
    (8, 8, (1, 3), (1, 1), (0, 1), 1, 1, 2),
    (64, 64, (3, 1), (1, 1), (1, 0), 1, 3, 3),
    (24, 136, (3, 3), (1, 1), (1, 1), 1, 5, 3),
]


@pytest.mark.parametrize('tile', [-1, 0, 1, 2, 3])
@pytest.mark.parametrize('cfg', CONVS)
def test_conv_fwd(cfg, tile, monkeypatch):
    Fn = _fn()
    if tile >= 0:
        monkeypatch.setenv('EMSA_CONV_TILE', str(tile))
    cin, cout, k, s, p, n, h, w = cfg
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, *k, seed=2, scale=0.1)
    b = rnd(cout, seed=3)
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=s, padding=p)
    spec = Fn.ConvSpec(cin, cout, k, s, p)
    wp = Fn.pack_weight(wt.to(DEV), 'fwd')
    y, stats = Fn.conv_fwd(to_act(x), wp, spec, bias=b.to(DEV), want_stats=True)
    torch.cuda.synchronize()
    close(y, ref, what='conv')
    cnt = ref.numel() / cout
    assert float(stats[2][:, 0].sum()) == cnt
    mean = stats[0].sum(0) / cnt
    close(mean, ref.mean((0, 2, 3)), what='stats mean')
    tile_mean = stats[0] / stats[2]
    m2 = stats[1].sum(0) + (stats[2] * (tile_mean - mean[None]) ** 2).sum(0)
    close(m2 / cnt, ref.var((0, 2, 3), unbiased=False), tol=2e-4, what='stats var')
    # fused epilogue: folded BN + residual + relu
    sc, sh = rnd(cout, seed=4), rnd(cout, seed=5)
    res = rnd(*ref.shape, seed=6)
    y2 = Fn.conv_fwd(to_act(x), wp, spec, bias=b.to(DEV), scale=sc.to(DEV), shift=sh.to(DEV),
                     residual=to_act(res), act=Fn.ACT_RELU)
    ref2 = F.relu(ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None] + res)
    close(y2, ref2, what='conv epilogue')


@pytest.mark.parametrize('tile', [-1, 0, 2])
@pytest.mark.parametrize('cfg', CONVS)
def test_conv_dgrad(cfg, tile, monkeypatch):
    Fn = _fn()
    if tile >= 0:
        monkeypatch.setenv('EMSA_CONV_TILE', str(tile))
    cin, cout, k, s, p, n, h, w = cfg
    x = rnd(n, cin, h, w, seed=1).double().requires_grad_(True)
    wt = rnd(cout, cin, *k, seed=2, scale=0.1)
    y = F.conv2d(x, wt.double(), None, stride=s, padding=p)
    dy = rnd(*y.shape, seed=7)
    y.backward(dy.double())
    spec = Fn.ConvSpec(cin, cout, k, s, p)
    wpd = Fn.pack_weight(wt.to(DEV), 'dgrad')
    dx = Fn.conv_dgrad(to_act(dy), wpd, spec, (h, w))
    torch.cuda.synchronize()
    close(dx, x.grad, what='dgrad')
    mask = rnd(n, cin, h, w, seed=8)
    res = rnd(n, cin, h, w, seed=9)
    dx2 = Fn.conv_dgrad(to_act(dy), wpd, spec, (h, w), mask_src=to_act(mask))
    close(dx2, x.grad * (mask > 0), what='dgrad mask')
    dx3 = Fn.conv_dgrad(to_act(dy), wpd, spec, (h, w), residual=to_act(res))
    close(dx3, x.grad + res, what='dgrad residual')


STRIDED = [
    # cin, cout, kernel, stride, padding, n, h, w: odd sizes (phases of unequal length), 2-D strides,
    # a 3x3 stride-2 kernel (four phases with 4 / 2 / 2 / 1 taps), stride 3
    (64, 128, (3, 1), (2, 1), (1, 0), 2, 15, 20),
    (64, 64, (1, 3), (1, 2), (0, 1), 1, 7, 13),
    (64, 128, (1, 1), (2, 2), (0, 0), 2, 15, 9),
    (16, 32, (3, 3), (2, 2), (1, 1), 2, 9, 11),
    (8, 8, (3, 3), (2, 2), (1, 1), 1, 4, 4),
    (16, 16, (1, 3), (1, 3), (0, 1), 1, 3, 11),
    (64, 128, (3, 1), (2, 1), (1, 0), 4, 60, 80),
]


@pytest.mark.parametrize('cfg', STRIDED)
def test_conv_dgrad_strided_phases(cfg, monkeypatch):
    """strided data gradient as one dense stride-1 launch per output phase (emsa_conv_igemm with
    the output pixel map) == fp64 autograd == the single launch over all taps, incl. the fused
    mask / residual epilogue (a phase without taps -- 1x1 stride 2 -- falls back to that one when
    an epilogue operand is given)"""
    Fn = _fn()
    cin, cout, k, s, p, n, h, w = cfg
    x = rnd(n, cin, h, w, seed=1).double().requires_grad_(True)
    wt = rnd(cout, cin, *k, seed=2, scale=0.1)
    y = F.conv2d(x, wt.double(), None, stride=s, padding=p)
    dy = rnd(*y.shape, seed=7)
    y.backward(dy.double())
    spec = Fn.ConvSpec(cin, cout, k, s, p)
    wpd = Fn.pack_weight(wt.to(DEV), 'dgrad')
    mask = rnd(n, cin, h, w, seed=8)
    res = rnd(n, cin, h, w, seed=9)
    outs = {}
    for phases in (True, False):
        monkeypatch.setattr(Fn, 'DGRAD_PHASES', phases)
        outs[phases] = (Fn.conv_dgrad(to_act(dy), wpd, spec, (h, w)),
                        Fn.conv_dgrad(to_act(dy), wpd, spec, (h, w), mask_src=to_act(mask)),
                        Fn.conv_dgrad(to_act(dy), wpd, spec, (h, w), residual=to_act(res)),
                        Fn.conv_dgrad(to_act(dy), wpd, spec, (h, w), mask_src=to_act(mask),
                                      residual=to_act(res)))
    torch.cuda.synchronize()
    for phases in (True, False):
        d0, d1, d2, d3 = outs[phases]
        close(d0, x.grad, what=f'dgrad (phases={phases})')
        close(d1, x.grad * (mask > 0), what=f'dgrad mask (phases={phases})')
        close(d2, x.grad + res, what=f'dgrad residual (phases={phases})')
        close(d3, (x.grad + res) * (mask > 0), what=f'dgrad residual+mask (phases={phases})')


WINO = [
    # cin, cout, kernel, n, h, w   (stride 1, "same" padding)
    (64, 64, (3, 3), 2, 12, 20),         # 3x3: Winograd along W, kernel rows in the GEMM K
    (128, 40, (3, 3), 1, 9, 13),
    (96, 8, (3, 3), 2, 5, 7),
    (8, 96, (3, 3), 1, 6, 6),
    (72, 64, (3, 3), 1, 3, 40),
    (64, 64, (3, 1), 2, 12, 20),
    (64, 64, (1, 3), 2, 12, 20),
    (128, 128, (1, 3), 3, 9, 13),        # odd line length: last pair of a line is half empty
    (128, 128, (3, 1), 3, 9, 13),
    (256, 128, (3, 1), 1, 15, 7),
    (512, 512, (1, 3), 2, 3, 5),
    (64, 40, (1, 3), 1, 30, 41),         # cout not a multiple of the 64-channel tile
    (72, 64, (3, 1), 1, 5, 40),          # cin not a multiple of the K step
    (64, 64, (1, 3), 1, 1, 1),           # single pixel
]


@pytest.mark.parametrize('cfg', WINO)
def test_conv1d_winograd_fwd(cfg):
    """F(2,3) Winograd kernel == direct convolution (fp64 reference) incl. the fused epilogues
    and the BatchNorm statistics partials"""
    Fn = _fn()
    cin, cout, k, n, h, w = cfg
    p = (k[0] // 2, k[1] // 2)
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, *k, seed=2, scale=0.1)
    b = rnd(cout, seed=3)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=p)
    spec = Fn.ConvSpec(cin, cout, k, (1, 1), p)
    assert Fn.wino_eligible(spec)
    u = Fn.pack_wino(wt.to(DEV))[0]
    y, stats = Fn.conv_fwd(to_act(x), None, spec, bias=b.to(DEV), want_stats=True, wino_u=u)
    torch.cuda.synchronize()
    close(y, ref, what='wino conv')
    cnt = ref.numel() / cout
    assert float(stats[2][:, 0].sum()) == cnt
    assert bool((stats[2] == stats[2][:, :1]).all())
    mean = stats[0].sum(0) / cnt
    close(mean, ref.mean((0, 2, 3)), what='stats mean')
    tile_mean = stats[0] / stats[2].clamp(min=1)
    m2 = stats[1].sum(0) + (stats[2] * (tile_mean - mean[None]) ** 2).sum(0)
    close(m2 / cnt, ref.var((0, 2, 3), unbiased=False), tol=2e-4, what='stats var')
    sc, sh = rnd(cout, seed=4), rnd(cout, seed=5)
    res = rnd(*ref.shape, seed=6)
    y2 = Fn.conv_fwd(to_act(x), None, spec, bias=b.to(DEV), scale=sc.to(DEV), shift=sh.to(DEV),
                     residual=to_act(res), act=Fn.ACT_RELU, wino_u=u)
    ref2 = F.relu(ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None] + res)
    close(y2, ref2, what='wino epilogue')
    # same result as the implicit-GEMM kernel to fp32 rounding
    wp = Fn.pack_weight(wt.to(DEV), 'fwd')
    y3 = Fn.conv_fwd(to_act(x), wp, spec, bias=b.to(DEV))
    close(y, y3.double(), tol=2e-5, what='wino vs igemm')
    # Winograd weights from the packed layout (merged head convs) == from the OIHW parameter
    u2 = Fn.pack_wino_packed(wp, cout, cin, Fn.wino_rows(spec), flip=False)
    assert torch.equal(u, u2)


@pytest.mark.parametrize('cfg', WINO)
def test_conv1d_winograd_dgrad(cfg):
    Fn = _fn()
    cin, cout, k, n, h, w = cfg
    p = (k[0] // 2, k[1] // 2)
    x = rnd(n, cin, h, w, seed=1).double().requires_grad_(True)
    wt = rnd(cout, cin, *k, seed=2, scale=0.1)
    y = F.conv2d(x, wt.double(), None, padding=p)
    dy = rnd(*y.shape, seed=7)
    y.backward(dy.double())
    spec = Fn.ConvSpec(cin, cout, k, (1, 1), p)
    ud = Fn.pack_wino(wt.to(DEV), fwd=False, dgrad=True)[1]
    dx = Fn.conv_dgrad(to_act(dy), None, spec, (h, w), wino_u=ud)
    torch.cuda.synchronize()
    close(dx, x.grad, what='wino dgrad')
    mask = rnd(n, cin, h, w, seed=8)
    res = rnd(n, cin, h, w, seed=9)
    dx2 = Fn.conv_dgrad(to_act(dy), None, spec, (h, w), mask_src=to_act(mask), wino_u=ud)
    close(dx2, x.grad * (mask > 0), what='wino dgrad mask')
    dx3 = Fn.conv_dgrad(to_act(dy), None, spec, (h, w), residual=to_act(res), wino_u=ud)
    close(dx3, x.grad + res, what='wino dgrad residual')
    ud2 = Fn.pack_wino_packed(Fn.pack_weight(wt.to(DEV), 'dgrad'), cin, cout, Fn.wino_rows(spec),
                              flip=True)
    assert torch.equal(ud, ud2)


@pytest.mark.parametrize('cfg', [(64, 64, (3, 1), 2, 12, 20), (128, 128, (1, 3), 3, 9, 13),
                                 (64, 40, (1, 3), 1, 7, 41), (64, 64, (3, 3), 1, 6, 10)])
def test_conv1d_winograd_relu_bit_masks(cfg):
    """forward with fused ReLU emits (out > 0) as bits; the data gradient of the NEXT conv masked
    by those bits == masked by the float tensor (bit-identical)"""
    Fn = _fn()
    cin, cout, k, n, h, w = cfg
    p = (k[0] // 2, k[1] // 2)
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, *k, seed=2, scale=0.1)
    spec = Fn.ConvSpec(cin, cout, k, (1, 1), p)
    u = Fn.pack_wino(wt.to(DEV))[0]
    y, bits = Fn.conv_fwd(to_act(x), None, spec, act=Fn.ACT_RELU, wino_u=u, want_relu_bits=True)
    assert bits is not None and bits.numel() == n * h * w * ((cout + 63) // 64)
    # decode on the host: word per (pixel, 64-channel tile), bit = (c % 4) * 16 + (c % 64) // 4
    yb = y.permute(0, 2, 3, 1).reshape(-1, cout).cpu() > 0
    words = bits.cpu().numpy().view('uint64').reshape(n * h * w, -1)
    import numpy as np
    for c in (0, 1, 5, cout - 1, cout // 2):
        got = (words[:, c // 64] >> np.uint64((c % 4) * 16 + (c % 64) // 4)) & np.uint64(1)
        assert np.array_equal(got.astype(bool), yb[:, c].numpy()), c
    # consumer: a conv whose INPUT has `cout` channels; its data gradient is masked by y > 0
    k2 = (k[1], k[0])
    spec2 = Fn.ConvSpec(cout, 64, k2, (1, 1), (k2[0] // 2, k2[1] // 2))
    w2 = rnd(64, cout, *k2, seed=4, scale=0.1)
    ud2 = Fn.pack_wino(w2.to(DEV), fwd=False, dgrad=True)[1]
    dy = to_act(rnd(n, 64, h, w, seed=5))
    a = Fn.conv_dgrad(dy, None, spec2, (h, w), mask_src=y, wino_u=ud2)
    b = Fn.conv_dgrad(dy, None, spec2, (h, w), wino_u=ud2, mask_bits=bits)
    assert torch.equal(a, b)
    assert float(a.abs().max()) > 0


def test_conv1d_winograd_channel_slice_views():
    """input and output that are channel slices of wider NHWC tensors (pixel stride > channels)"""
    Fn = _fn()
    n, h, w = 2, 7, 11
    wide_in = to_act(rnd(n, 192, h, w, seed=1))
    x = wide_in[:, 64:128]
    wt = rnd(64, 64, 1, 3, seed=2, scale=0.1)
    spec = Fn.ConvSpec(64, 64, (1, 3), (1, 1), (0, 1))
    wide_out = Fn.act_empty(n, 128, h, w, DEV).zero_()
    Fn.conv_fwd(x, None, spec, out=wide_out[:, 64:], wino_u=Fn.pack_wino(wt.to(DEV))[0])
    ref = F.conv2d(x.cpu().double(), wt.double(), padding=(0, 1))
    close(wide_out[:, 64:], ref, what='wino slice')
    assert float(wide_out[:, :64].abs().max()) == 0.0


@pytest.mark.parametrize('cfg', CONVS)
def test_conv_wgrad(cfg):
    Fn = _fn()
    cin, cout, k, s, p, n, h, w = cfg
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, *k, seed=2, scale=0.1).double().requires_grad_(True)
    b = rnd(cout, seed=3).double().requires_grad_(True)
    y = F.conv2d(x.double(), wt, b, stride=s, padding=p)
    dy = rnd(*y.shape, seed=7)
    y.backward(dy.double())
    spec = Fn.ConvSpec(cin, cout, k, s, p)
    dwp, db, packed = Fn.conv_wgrad(to_act(x), to_act(dy), spec, True)
    assert packed
    dw = Fn.unpack_wgrad(dwp, wt.detach().float().to(DEV))
    torch.cuda.synchronize()
    close(dw, wt.grad, what='wgrad')
    close(db, b.grad, what='bgrad')
    # deterministic two-pass form (stride-1 3-tap and 3x3 convs): OIHW directly, bit-reproducible
    like = wt.detach().float().to(DEV)
    dw2, db2, packed2 = Fn.conv_wgrad(to_act(x), to_act(dy), spec, True, like=like, two_pass=True)
    if not packed2:
        assert k in ((3, 1), (1, 3), (3, 3)) and s == (1, 1)
        close(dw2, wt.grad, what='wgrad two-pass')
        close(db2, b.grad, what='bgrad two-pass')
        dw3, db3, _ = Fn.conv_wgrad(to_act(x), to_act(dy), spec, True, like=like, two_pass=True)
        assert torch.equal(dw2, dw3) and torch.equal(db2, db3)
    else:
        assert not (k in ((3, 1), (1, 3), (3, 3)) and s == (1, 1))


def test_conv_wgrad_split_k_large():
    """long pixel axis (split-K with atomics) at the 1-D conv shape of the /4 stage"""
    Fn = _fn()
    cin = cout = 64
    n, h, w = 2, 120, 160
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, 1, 3, seed=2, scale=0.1).double().requires_grad_(True)
    y = F.conv2d(x.double(), wt, None, padding=(0, 1))
    dy = rnd(*y.shape, seed=7, scale=0.1)
    y.backward(dy.double())
    spec = Fn.ConvSpec(cin, cout, (1, 3), 1, (0, 1))
    dwp, _, _ = Fn.conv_wgrad(to_act(x), to_act(dy), spec, False)
    dw = Fn.unpack_wgrad(dwp, wt.detach().float().to(DEV))
    close(dw, wt.grad, tol=2e-4, what='wgrad split-k')
    like = wt.detach().float().to(DEV)
    dw2, _, packed = Fn.conv_wgrad(to_act(x), to_act(dy), spec, False, like=like, two_pass=True)
    assert not packed
    close(dw2, wt.grad, tol=2e-4, what='wgrad split-k two-pass')
    dw3, _, _ = Fn.conv_wgrad(to_act(x), to_act(dy), spec, False, like=like, two_pass=True)
    assert torch.equal(dw2, dw3), 'two-pass weight gradient is not bit-reproducible'


@pytest.mark.parametrize('shape', [(2, 32, 48), (1, 22, 26)])
@pytest.mark.parametrize('cin,rows', [(3, True), (1, True), (1, False)])
def test_stem(cin, rows, shape, monkeypatch):
    """rows: the one-channel stem in the rows-as-channels layout (two super-taps) vs the generic
    NHWC4 layout (seven row taps)"""
    Fn = _fn()
    monkeypatch.setattr(Fn, 'STEM_ROWS', rows)
    n, h, w = shape
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(64, cin, 7, 7, seed=2, scale=0.1).double().requires_grad_(True)
    y = F.conv2d(x.double(), wt, None, stride=2, padding=3)
    dy = rnd(*y.shape, seed=3)
    y.backward(dy.double())
    spec = Fn.StemSpec(cin)
    xp = Fn.stem_pack_input(x.to(DEV))
    wp = Fn.stem_pack_weight(wt.detach().float().to(DEV))
    out, stats = Fn.stem_fwd(xp, wp, spec, n, h, w)
    close(out, y, what='stem fwd')
    dw = Fn.stem_wgrad(xp, to_act(dy), spec, n, h, w, wt.detach().float().to(DEV))
    close(dw, wt.grad, what='stem wgrad')


def test_pack_roundtrip_and_blockdiag():
    Fn = _fn()
    w = rnd(5, 7, 3, 3, seed=1).to(DEV)
    wp = Fn.pack_weight(w, 'fwd', 8, 2, 12, 4)
    ref = torch.zeros(9, 8, 12)
    ref[:, 2:7, 4:11] = w.cpu().permute(2, 3, 0, 1).reshape(9, 5, 7)
    close(wp.view(9, 8, 12), ref, tol=0, what='pack fwd')
    wd = Fn.pack_weight(w, 'dgrad', 8, 2, 12, 4)
    close(wd.view(9, 12, 8), ref.transpose(1, 2), tol=0, what='pack dgrad')
    back = Fn.unpack_wgrad(wp, w, 8, 2, 12, 4)
    close(back, w, tol=0, what='unpack')


# (1024 / 2048: the widest stages of the bottleneck ResNets; fp32 with 2048 channels has more channel
#  vectors than a workgroup has threads and takes bn_bwd_reduce_wide_kernel)
@pytest.mark.parametrize('c', [64, 128, 512, 40, 1024, 2048])
@pytest.mark.parametrize('act', [0, 1])
@pytest.mark.parametrize('train', [True, False])
def test_bn_fwd_bwd(c, act, train):
    Fn = _fn()
    n, h, w = 3, 7, 9
    x = rnd(n, c, h, w, seed=1).double().requires_grad_(True)
    res = rnd(n, c, h, w, seed=2).double().requires_grad_(True)
    gamma = (rnd(c, seed=3) * 0.2 + 1).double().requires_grad_(True)
    beta = rnd(c, seed=4).double().requires_grad_(True)
    drop = (torch.rand(n, c, generator=torch.Generator().manual_seed(0)) > 0.3).double() / 0.7
    rm, rv = rnd(c, seed=5) * 0.1, rnd(c, seed=6).abs() + 0.5
    eps = 1e-3
    rm_ref, rv_ref = rm.clone().double(), rv.clone().double()
    yb = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, training=train, momentum=0.1, eps=eps)
    out = yb * drop[:, :, None, None] + res
    if act:
        out = F.relu(out)
    dy = rnd(n, c, h, w, seed=7)
    out.backward(dy.double())

    g, b = gamma.detach().float().to(DEV), beta.detach().float().to(DEV)
    xa = to_act(x.detach().float())
    rmg, rvg = rm.clone().to(DEV), rv.clone().to(DEV)
    if train:
        # statistics the way the conv epilogue delivers them: [2][rows][c] partial sums
        xs = x.detach().float().permute(0, 2, 3, 1).reshape(-1, c)
        rows = 5
        chunks = torch.chunk(xs, rows, 0)
        stats = torch.stack([torch.stack([ch.sum(0) for ch in chunks]),
                             torch.stack([((ch - ch.mean(0)) ** 2).sum(0) for ch in chunks]),
                             torch.stack([torch.full((c,), float(ch.shape[0])) for ch in chunks])
                             ]).to(DEV)
        scale, shift, mean, invstd = Fn.bn_finalize(stats, xs.shape[0], g, b, eps, 0.1, rmg, rvg)
        close(rmg, rm_ref, what='running mean')
        close(rvg, rv_ref, what='running var')
    else:
        scale, shift, invstd = Fn.bn_fold(g, b, rmg, rvg, eps)
        mean = rmg
    y = Fn.bn_act(xa, scale, shift, drop.float().to(DEV), to_act(res.detach().float()), act)
    close(y, out, what='bn_act fwd')
    dx, dres, dg, db = Fn.bn_bwd(to_act(dy), y, xa, g, mean, invstd, drop.float().to(DEV), act,
                                 train, want_dres=True)
    close(dx, x.grad, what='bn dx')
    close(dres, res.grad, what='bn dres')
    close(dg, gamma.grad, what='bn dgamma')
    close(db, beta.grad, what='bn dbeta')
    # ReLU mask as 1 bit per element instead of re-reading y: bit-identical results
    y2, bits = Fn.bn_act(xa, scale, shift, drop.float().to(DEV), to_act(res.detach().float()), act,
                         want_mask=True)
    assert torch.equal(y2, y)
    if act == Fn.ACT_RELU:
        assert bits.dtype == torch.int64 and bits.numel() * 64 >= y.numel()
        r2 = Fn.bn_bwd(to_act(dy), bits, xa, g, mean, invstd, drop.float().to(DEV), act, train,
                       want_dres=True)
        for a, b_, name in zip(r2, (dx, dres, dg, db), ('dx', 'dres', 'dgamma', 'dbeta')):
            assert torch.equal(a, b_), f'bit-mask path differs: {name}'
    else:
        assert bits is None


def test_dropout_mask_matches_oracle():
    Fn = _fn()
    from oracle.emsanet_oracle import dropout2d_scale_mask
    for seed, lid, p in ((0, 0, 0.1), (123456789, 7, 0.2), (0xFFFFFFFF, 49, 0.5)):
        m = Fn.dropout2d_mask(32, 512, p, seed, lid, DEV)
        ref = torch.from_numpy(dropout2d_scale_mask(seed, lid, 32, 512, p))
        assert torch.equal(m.cpu(), ref)
        assert 0.0 < (ref == 0).float().mean() < 2 * p + 0.05


@pytest.mark.parametrize('shape', [(2, 64, 12, 20), (1, 64, 7, 9)])
def test_maxpool(shape):
    Fn = _fn()
    x = rnd(*shape, seed=1).double().requires_grad_(True)
    y = F.max_pool2d(x, 3, stride=2, padding=1)
    dy = rnd(*y.shape, seed=2)
    y.backward(dy.double())
    yg, idx = Fn.maxpool_fwd(to_act(x.detach().float()))
    close(yg, y, tol=0, what='maxpool')
    dx = Fn.maxpool_bwd(to_act(dy), idx, shape[2:])
    close(dx, x.grad, tol=1e-6, what='maxpool bwd')


@pytest.mark.parametrize('c', [64, 512, 1024, 2048])       # (2048 fp32: channel_dot_wide_kernel)
def test_se_fusion(c):
    Fn = _fn()
    from emsanet_amd import ops
    from emsanet_amd.nn import SEAddUniRGB
    torch.manual_seed(0)
    n, h, w = 3, 6, 10
    mod = SEAddUniRGB(c)
    ref = mod.double()
    rgb = rnd(n, c, h, w, seed=1).double().requires_grad_(True)
    dep = rnd(n, c, h, w, seed=2).double().requires_grad_(True)

    def se(m, x):
        return x * m.fc(F.adaptive_avg_pool2d(x, 1))
    out = se(ref.se_rgb, rgb) + se(ref.se_depth, dep)
    dy = rnd(n, c, h, w, seed=3)
    out.backward(dy.double())
    ref_grads = [p.grad.clone() for p in ref.parameters()]

    gm = SEAddUniRGB(c)
    gm.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    gm.to(DEV)
    r = to_act(rgb.detach().float()).requires_grad_(True)
    d = to_act(dep.detach().float()).requires_grad_(True)
    o, _ = gm(r, d)
    close(o, out, what='se fwd')
    o.backward(to_act(dy))
    close(r.grad, rgb.grad, what='se d_rgb')
    close(d.grad, dep.grad, what='se d_depth')
    for (k, p), g in zip(gm.named_parameters(), ref_grads):
        close(p.grad, g, tol=2e-4, what=f'se {k}')
    # the depth stream continues on the Function's pass-through output: a gradient arriving there
    # (the next depth stage's) is added inside the SE backward kernel, not by a torch add
    r.grad = d.grad = None
    o, d_next = gm(r, d)
    assert d_next.data_ptr() == d.data_ptr()
    dn = rnd(n, c, h, w, seed=9)
    torch.autograd.backward([o, d_next], [to_act(dy), to_act(dn)])
    close(r.grad, rgb.grad, what='se d_rgb (pass-through)')
    close(d.grad, dep.grad + dn.double(), what='se d_depth + next stage')


@pytest.mark.parametrize('c,cp', [(64, 64), (40, 40), (5, 8)])
def test_upsample_dw(c, cp):
    Fn = _fn()
    from emsanet_amd import ops
    n, h, w = 2, 5, 7
    x = torch.zeros(n, cp, h, w)
    x[:, :c] = rnd(n, c, h, w, seed=1)
    x = x.double().requires_grad_(True)
    wt = torch.zeros(cp, 1, 3, 3)
    wt[:c] = rnd(c, 1, 3, 3, seed=2)
    wt = wt.double().requires_grad_(True)
    b = torch.zeros(cp)
    b[:c] = rnd(c, seed=3)
    b = b.double().requires_grad_(True)
    skip = rnd(n, cp, 2 * h, 2 * w, seed=4).double().requires_grad_(True)
    y = F.conv2d(F.interpolate(x, scale_factor=2, mode='nearest'), wt, b, padding=1, groups=cp) + skip
    dy = rnd(*y.shape, seed=5)
    y.backward(dy.double())
    xg = to_act(x.detach().float()).requires_grad_(True)
    wg = wt.detach().float().to(DEV).requires_grad_(True)
    bg = b.detach().float().to(DEV).requires_grad_(True)
    sg = to_act(skip.detach().float()).requires_grad_(True)
    yg = ops.UpsampleDWFunction.apply(xg, wg, bg, sg)
    close(yg, y, what='up fwd')
    yg.backward(to_act(dy))
    close(xg.grad, x.grad, what='up dx')
    close(wg.grad, wt.grad, what='up dw')
    close(bg.grad, b.grad, what='up db')
    close(sg.grad, skip.grad, what='up dskip')


@pytest.mark.parametrize('c', [8, 16, 32, 40, 64, 128])
@pytest.mark.parametrize('shape', [(2, 5, 7), (1, 9, 33), (3, 4, 4)])
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_upsample_dw_backward_one_pass(c, shape, dtype, monkeypatch):
    """the LDS-tiled single-pass backward (dx, dw, db from one read of dy) for every instantiated
    (channel chunk, tile width) pair, tiles ragged in both directions: against the fp64 reference
    and against the two separate passes; dy in the feature type and as fp32 (model boundary)"""
    Fn = _fn()
    n, h, w = shape
    x = rnd(n, c, h, w, seed=1)
    wt = rnd(c, 1, 3, 3, seed=2, scale=0.3)
    dy = rnd(n, c, 2 * h, 2 * w, seed=5)
    if dtype != torch.float32:
        x, dy16 = x.to(dtype).float(), dy.to(dtype).float()
    else:
        dy16 = dy
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    cases = [(dy16, to_act(dy16).to(dtype))]
    if dtype != torch.float32:
        cases.append((dy, to_act(dy)))               # fp32 cotangent of a 16-bit head
    for dy_ref, dy_dev in cases:
        xr = x.double().requires_grad_(True)
        wr = wt.double().requires_grad_(True)
        br = torch.zeros(c).double().requires_grad_(True)
        y = F.conv2d(F.interpolate(xr, scale_factor=2, mode='nearest'), wr, br, padding=1, groups=c)
        y.backward(dy_ref.double())
        xd = to_act(x).to(dtype)
        assert Fn._lib.lib().emsa_up2x_dw3x3_bwd_supported(c, xd.element_size()) == 1
        monkeypatch.setattr(Fn, 'UP2X_FUSED_BWD', True)
        dx, dw, db = Fn.up2x_dw_bwd(dy_dev, xd, wt.to(DEV).contiguous())
        monkeypatch.setattr(Fn, 'UP2X_FUSED_BWD', False)
        dx0, dw0, db0 = Fn.up2x_dw_bwd(dy_dev, xd, wt.to(DEV).contiguous())
        assert dx.dtype == dtype
        close(dx, xr.grad, tol=tol, what='one-pass dx')
        close(dw.view(c, 1, 3, 3), wr.grad, tol=2e-4, what='one-pass dw')
        close(db, br.grad, tol=2e-4, what='one-pass db')
        close(dx, dx0.float().cpu(), tol=tol, what='one-pass dx vs separate passes')
        close(dw, dw0.cpu(), tol=2e-4, what='one-pass dw vs separate passes')
        # weight gradient only (dx not requested)
        monkeypatch.setattr(Fn, 'UP2X_FUSED_BWD', True)
        nodx, dw1, db1 = Fn.up2x_dw_bwd(dy_dev, xd, wt.to(DEV).contiguous(), need_dx=False)
        assert nodx is None
        close(dw1, dw.cpu(), tol=2e-4, what='one-pass dw without dx')


def test_ppm_pieces():
    Fn = _fn()
    from emsanet_amd import ops
    n, c, h, w = 2, 32, 15, 20
    x = rnd(n, c, h, w, seed=1).double().requires_grad_(True)
    y1 = rnd(n, 16, 1, 1, seed=2).double().requires_grad_(True)
    y5 = rnd(n, 16, 5, 5, seed=3).double().requires_grad_(True)
    for bins in (1, 5, 4):
        xx = x.detach().clone().requires_grad_(True)
        p = F.adaptive_avg_pool2d(xx, bins)
        dp = rnd(*p.shape, seed=4)
        p.backward(dp.double())
        xg = to_act(x.detach().float()).requires_grad_(True)
        pg = ops.AdaptiveAvgPoolFunction.apply(xg, bins)
        close(pg, p, what=f'pool{bins}')
        pg.backward(to_act(dp))
        close(xg.grad, xx.grad, what=f'pool{bins} bwd')
    cat = torch.cat([x, F.interpolate(y1, (h, w), mode='bilinear', align_corners=False),
                     F.interpolate(y5, (h, w), mode='bilinear', align_corners=False)], 1)
    dc = rnd(*cat.shape, seed=5)
    cat.backward(dc.double())
    xg = to_act(x.detach().float()).requires_grad_(True)
    y1g = to_act(y1.detach().float()).requires_grad_(True)
    y5g = to_act(y5.detach().float()).requires_grad_(True)
    cg = ops.PPMConcatFunction.apply(xg, y1g, y5g)
    close(cg, cat, what='ppm cat')
    cg.backward(to_act(dc))
    close(xg.grad, x.grad, what='ppm dx')
    close(y1g.grad, y1.grad, what='ppm dy1')
    close(y5g.grad, y5.grad, what='ppm dy5')


@pytest.mark.parametrize('ih,iw,oh,ow', [(5, 5, 15, 20), (1, 1, 15, 20), (7, 9, 30, 41), (30, 40, 15, 20),
                                         (15, 20, 480, 640), (3, 640, 6, 640)])
def test_bilinear_bwd_gather(ih, iw, oh, ow):
    """the gather-form bilinear backward (round 6: one thread per dx element over a WINDOW of output
    rows / columns, fixed summation order) against torch autograd, up- and down-sampling, clamped
    borders, a one-pixel source, a large map; and bit-reproducible run to run"""
    Fn = _fn()
    n, c = 2, 8
    x = rnd(n, c, ih, iw, seed=1).double().requires_grad_(True)
    y = F.interpolate(x, (oh, ow), mode='bilinear', align_corners=False)
    dy = rnd(n, c, oh, ow, seed=2)
    y.backward(dy.double())
    got = Fn.bilinear_bwd(to_act(dy), (ih, iw))
    close(got, x.grad, tol=2e-5, what='bilinear bwd')
    again = Fn.bilinear_bwd(to_act(dy), (ih, iw))
    assert torch.equal(got, again)
    out = Fn.act_empty(n, c, oh, ow, DEV)
    close(Fn.bilinear_fwd(to_act(x.detach().float()), out), y, what='bilinear fwd')


@pytest.mark.parametrize('ih,iw,oh,ow', [(5, 5, 15, 20), (1, 1, 15, 20), (7, 9, 30, 41), (30, 40, 15, 20), (3, 5, 23, 31)])
def test_nearest_up(ih, iw, oh, ow):
    """'nearest' up-sampling of the pyramid-pooling branches (--upsampling-context-module nearest,
    /root/reference/emsanet/args.py:250-256) and its gather-form backward vs torch"""
    Fn = _fn()
    n, c = 2, 8
    x = rnd(n, c, ih, iw, seed=1).double().requires_grad_(True)
    y = F.interpolate(x, (oh, ow), mode='nearest')
    dy = rnd(n, c, oh, ow, seed=2)
    y.backward(dy.double())
    out = Fn.act_empty(n, c, oh, ow, DEV)
    close(Fn.nearest_fwd(to_act(x.detach().float()), out), y, tol=0.0, what='nearest fwd')
    close(Fn.nearest_bwd(to_act(dy), (ih, iw)), x.grad, tol=2e-6, what='nearest bwd')


def test_head_act():
    from emsanet_amd import ops
    x = rnd(2, 8, 6, 7, seed=1).double().requires_grad_(True)
    y = torch.cat([torch.sigmoid(x[:, :1]), torch.tanh(x[:, 1:3]), x[:, 3:]], 1)
    dy = rnd(*y.shape, seed=2)
    y.backward(dy.double())
    xg = to_act(x.detach().float()).requires_grad_(True)
    c, o, r = ops.HeadActFunction.apply(xg, 1, 2, (1, 2, 2))      # task slices of one tensor
    assert c.shape[1] == 1 and o.shape[1] == 2 and r.shape[1] == 2
    close(torch.cat([c, o, r], 1), y[:, :5], what='head act')
    # gradients arrive per task (contiguous NCHW like a loss would produce them); the padding
    # channels 5..7 receive none
    torch.autograd.backward([c, o, r], [dy[:, :1].contiguous().to(DEV), dy[:, 1:3].contiguous().to(DEV),
                                        dy[:, 3:5].contiguous().to(DEV)])
    ref = x.grad.clone()       # activation-wise gradient; no cotangent reached channels 5..7
    ref[:, 5:] = 0
    close(xg.grad, ref, what='head act bwd')


def test_head_act_l2_normalised_orientation():
    """[U] switch ORIENTATION_L2_NORMALIZE: the orientation pair is F.normalize(dim=1)'d"""
    from emsanet_amd import ops
    x = rnd(2, 8, 6, 7, seed=1).double()
    x[0, 3:5, 0, 0] = 0.0                        # a zero vector: eps branch
    x.requires_grad_(True)
    y = torch.cat([torch.sigmoid(x[:, :1]), torch.tanh(x[:, 1:3]),
                   torch.nn.functional.normalize(x[:, 3:5], dim=1), x[:, 5:]], 1)
    dy = rnd(*y.shape, seed=2)
    dy[0, 3:5, 0, 0] = 0.0                       # (the eps branch's 1e12 gain is not a parity case)
    y.backward(dy.double())
    xg = to_act(x.detach().float()).requires_grad_(True)
    c, o, r = ops.HeadActFunction.apply(xg, 1, 2, (1, 2, 2), 2)
    close(torch.cat([c, o, r], 1), y[:, :5], what='head act + normalise')
    torch.autograd.backward([c, o, r], [dy[:, :1].contiguous().to(DEV), dy[:, 1:3].contiguous().to(DEV),
                                        dy[:, 3:5].contiguous().to(DEV)])
    ref = x.grad.clone()
    ref[:, 5:] = 0
    close(xg.grad, ref, what='head act + normalise bwd')


def test_copy_axpy():
    Fn = _fn()
    x = rnd(2, 12, 3, 5, seed=1)
    xa = to_act(x)
    dst = Fn.act_zeros(2, 20, 3, 5, DEV)
    Fn.copy_channels(xa[:, 4:12], dst[:, 2:10])
    ref = torch.zeros(2, 20, 3, 5)
    ref[:, 2:10] = x[:, 4:12]
    close(dst, ref, tol=0, what='copy')
    a = rnd(1003, seed=2).to(DEV)
    b = rnd(1003, seed=3).to(DEV)
    b0 = b.clone()
    Fn.axpy_(b, a, 0.5)
    close(b, b0.cpu() + 0.5 * a.cpu(), tol=1e-6, what='axpy')


def test_conv1d_winograd_bf16_mfma_subprocess():
    """EMSA_BF16_MFMA=1 (opt-in mixed precision, BASELINE config 3): operands rounded to bf16 for
    the matrix instruction, fp32 accumulate / storage.  Checked in a subprocess (the mode is a
    process-wide switch) against the fp64 convolution with a bf16-sized tolerance."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, torch
sys.path.insert(0, "tests")
import torch.nn.functional as F
from util import DEV, rnd, to_act
from emsanet_amd import functional as Fn
worst = 0.0
for cin, cout, k in ((64, 64, (1, 3)), (256, 128, (3, 1)), (128, 40, (3, 3))):
    x = rnd(2, cin, 12, 20, seed=1); wt = rnd(cout, cin, *k, seed=2, scale=0.1)
    ref = F.conv2d(x.double(), wt.double(), padding=(k[0] // 2, k[1] // 2))
    spec = Fn.ConvSpec(cin, cout, k, (1, 1), (k[0] // 2, k[1] // 2))
    y = Fn.conv_fwd(to_act(x), None, spec, wino_u=Fn.pack_wino(wt.to(DEV))[0])
    err = (y.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    worst = max(worst, err)
    assert 1e-5 < err < 2e-2, err      # really bf16 (not the fp32 path), and bf16-accurate
    # weight gradient (Winograd F(3,2) kernel, operands rounded to bf16 at the LDS store)
    wt64 = wt.double().requires_grad_(True)
    b64 = torch.zeros(cout, dtype=torch.float64, requires_grad=True)
    yy = F.conv2d(x.double(), wt64, b64, padding=(k[0] // 2, k[1] // 2))
    dy = rnd(*yy.shape, seed=7)
    yy.backward(dy.double())
    like = wt.to(DEV)
    dw, db, packed = Fn.conv_wgrad(to_act(x), to_act(dy), spec, True, like=like, two_pass=True)
    assert not packed
    errw = (dw.cpu().double() - wt64.grad).abs().max().item() / wt64.grad.abs().max().item()
    errb = (db.cpu().double() - b64.grad).abs().max().item() / b64.grad.abs().max().item()
    assert 1e-5 < errw < 2e-2, errw
    assert errb < 1e-4, errb           # the bias gradient is summed in fp32
    worst = max(worst, errw)
print("BF16_OK %.2e" % worst)
'''
    env = dict(os.environ, EMSA_BF16_MFMA='1')
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True,
                       text=True, timeout=300)
    assert 'BF16_OK' in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('kernel', [(3, 1), (1, 3)])
@pytest.mark.parametrize('c,shape', [(64, (2, 9, 11)), (128, (3, 8, 20)), (72, (1, 5, 7))])
@pytest.mark.parametrize('with_res', [False, True])
def test_dgrad_with_fused_bn_backward_sums(dtype, kernel, c, shape, with_res):
    """conv3x1_2 -> bn1 of the NBt1D backward: the data gradient whose epilogue applies the
    BatchNorm's ReLU mask and emits its backward reduction (emsa_conv1d_wino_bnb /
    emsa_conv_igemm_bnb_t) + emsa_bn_bwd_apply_rows_t == fp64 autograd of
    conv(relu(batch_norm(t))), and == the separate passes (dgrad, bn_bwd_reduce, bn_bwd_apply)"""
    Fn = _fn()
    n, h, w = shape
    pad = (1, 0) if kernel == (3, 1) else (0, 1)
    spec = Fn.ConvSpec(c, c, kernel, 1, pad)
    lo = dtype != torch.float32
    q = (lambda t: t.to(dtype).float()) if lo else (lambda t: t)
    t = q(rnd(n, c, h, w, seed=1))
    wt = q(rnd(c, c, *kernel, seed=2, scale=0.1))
    dy = q(rnd(n, c, h, w, seed=3))
    res = q(rnd(n, c, h, w, seed=4)) if with_res else None
    gamma = rnd(c, seed=5) * 0.2 + 1
    beta = rnd(c, seed=6) * 0.3
    eps = 1e-3
    # fp64 reference: a = relu(bn(t)) (batch statistics), z = conv(a) (+ a second consumer whose
    # gradient arrives as the residual operand)
    tr = t.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.relu(F.batch_norm(tr, None, None, gr, br, training=True, eps=eps))
    z = F.conv2d(a, wt.double(), padding=pad)
    loss = (z * dy.double()).sum()
    if with_res:
        loss = loss + (a * res.double()).sum()
    loss.backward()
    # engine: statistics of t the way bn_finalize produces them
    g, b = gamma.to(DEV), beta.to(DEV)
    xs = t.permute(0, 2, 3, 1).reshape(-1, c)
    stats = torch.stack([xs.sum(0, keepdim=True), ((xs - xs.mean(0)) ** 2).sum(0, keepdim=True),
                         torch.full((1, c), float(xs.shape[0]))]).to(DEV)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    scale, shift, mean, invstd = Fn.bn_finalize(stats, xs.shape[0], g, b, eps, 0.1, rm, rv)
    ta = to_act(t).to(dtype)
    dya = to_act(dy).to(dtype)
    resa = to_act(res).to(dtype) if with_res else None
    if lo:
        wpd = Fn.pack_weight_t(wt.to(DEV), dtype, fwd=False, dgrad=True)[1]
        gm, partial, rows = Fn.conv_dgrad_bnb(dya, wpd, spec, (h, w), ta, scale, shift, mean, invstd,
                                              residual=resa)
    else:
        ud = Fn.pack_wino(wt.to(DEV), fwd=False, dgrad=True)[1]
        gm, partial, rows = Fn.conv_dgrad_bnb(dya, None, spec, (h, w), ta, scale, shift, mean, invstd,
                                              residual=resa, wino_u=ud)
    dx, dg, db = Fn.bn_bwd_from_rows(gm, ta, g, mean, invstd, partial, rows, True)
    tol = 2e-2 if lo else 2e-4
    close(dx, tr.grad, tol=tol, what='fused dx')
    close(dg, gr.grad, tol=1e-2 if lo else 2e-4, what='fused dgamma')
    close(db, br.grad, tol=1e-2 if lo else 2e-4, what='fused dbeta')
    # the separate passes on the same inputs
    a_dev, bits = Fn.bn_act(ta, scale, shift, None, None, Fn.ACT_RELU, want_mask=True)
    if lo:
        da = Fn.conv_dgrad(dya, wpd, spec, (h, w), residual=resa)
    else:
        da = Fn.conv_dgrad(dya, None, spec, (h, w), residual=resa, wino_u=ud)
    dx0, _, dg0, db0 = Fn.bn_bwd(da, bits, ta, g, mean, invstd, None, Fn.ACT_RELU, True,
                                 want_dres=False)
    close(dx, dx0.float().cpu(), tol=tol, what='fused vs separate dx')
    close(dg, dg0.cpu(), tol=1e-2 if lo else 2e-4, what='fused vs separate dgamma')
    close(db, db0.cpu(), tol=1e-2 if lo else 2e-4, what='fused vs separate dbeta')


WINO_1D = [c for c in WINO if c[2] != (3, 3)]


@pytest.mark.parametrize('cfg', WINO_1D)
def test_conv1d_winograd_folded_input_bn(cfg):
    """the BatchNorm + ReLU in front of a 1-D conv formed in the conv's loader
    (emsa_conv1d_wino_inbn) and again in the loader of its weight gradient (emsa_conv_wgrad_inbn):
    == conv(relu(x * scale + shift)) with ZERO padding of the normalised tensor (a padding row
    must not turn into relu(shift)), odd line lengths, channel tails, statistics, ReLU bits."""
    Fn = _fn()
    cin, cout, k, n, h, w = cfg
    p = (k[0] // 2, k[1] // 2)
    x = rnd(n, cin, h, w, seed=1)
    sc = rnd(cin, seed=11).abs() + 0.5
    sh = rnd(cin, seed=12) + 0.3            # mostly positive shifts: padding would show
    wt = rnd(cout, cin, *k, seed=2, scale=0.1).double().requires_grad_(True)
    b = rnd(cout, seed=3).double().requires_grad_(True)
    # the kernels form a with ONE fused multiply-add in fp32; the reference takes the same values
    a = F.relu(torch.addcmul(sh[None, :, None, None], x, sc[None, :, None, None])).double()
    ref = F.conv2d(a, wt, b, padding=p)
    spec = Fn.ConvSpec(cin, cout, k, (1, 1), p)
    u = Fn.pack_wino(wt.detach().float().to(DEV))[0]
    aff = (sc.to(DEV), sh.to(DEV))
    (y, stats), bits = Fn.conv_fwd(to_act(x), None, spec, bias=b.detach().float().to(DEV),
                                   want_stats=True, act=Fn.ACT_RELU, wino_u=u,
                                   want_relu_bits=True, in_affine=aff)
    torch.cuda.synchronize()
    close(y, F.relu(ref), tol=2e-4, what='folded-bn conv')
    # the same launch without the fold on the materialised tensor: bit-identical epilogue products
    (y0, stats0), bits0 = Fn.conv_fwd(to_act(a.float()), None, spec,
                                      bias=b.detach().float().to(DEV), want_stats=True,
                                      act=Fn.ACT_RELU, wino_u=u, want_relu_bits=True)
    close(y, y0.double(), tol=2e-5, what='folded vs materialised')
    close(stats[0].sum(0), stats0[0].sum(0).double(), tol=2e-4, what='stats sums')
    assert bits is not None and bits.shape == bits0.shape
    # weight gradient with the same fold
    dy = rnd(*ref.shape, seed=7)
    ref.backward(dy.double())
    like = wt.detach().float().to(DEV)
    dw, db, packed = Fn.conv_wgrad(to_act(x), to_act(dy), spec, True, like=like, two_pass=True,
                                   in_affine=aff)
    torch.cuda.synchronize()
    assert not packed
    close(dw, wt.grad, tol=2e-4, what='folded-bn wgrad')
    close(db, b.grad, tol=2e-4, what='folded-bn bgrad')
    dw2, _, _ = Fn.conv_wgrad(to_act(x), to_act(dy), spec, True, like=like, two_pass=True,
                              in_affine=aff)
    assert torch.equal(dw, dw2)                               # deterministic two-pass form


def test_folded_input_bn_wgrad_under_bf16_mfma_subprocess():
    """with the opt-in EMSA_BF16_MFMA=1 the weight gradient of a conv whose forward folded
    its input's BatchNorm must still be taken against relu(x * scale + shift) (the exact-fp32
    Winograd F(3,2) kernel with the loader fold), not against the raw BatchNorm input"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = '''
import sys, torch, torch.nn.functional as F
sys.path.insert(0, "tests")
from util import DEV, rnd, to_act
from emsanet_amd import functional as Fn
cin = cout = 64; k = (3, 1); p = (1, 0)
x = rnd(2, cin, 12, 20, seed=1)
sc = rnd(cin, seed=11).abs() + 0.5
sh = rnd(cin, seed=12) + 0.3
wt = rnd(cout, cin, *k, seed=2, scale=0.1).double().requires_grad_(True)
a = F.relu(torch.addcmul(sh[None, :, None, None], x, sc[None, :, None, None])).double()
ref = F.conv2d(a, wt, None, padding=p)
dy = rnd(*ref.shape, seed=7)
ref.backward(dy.double())
spec = Fn.ConvSpec(cin, cout, k, (1, 1), p)
like = wt.detach().float().to(DEV)
dw, db, packed = Fn.conv_wgrad(to_act(x), to_act(dy), spec, False, like=like, two_pass=True,
                               in_affine=(sc.to(DEV), sh.to(DEV)))
torch.cuda.synchronize()
err = (dw.cpu().double() - wt.grad).abs().max().item() / wt.grad.abs().max().item()
assert err < 2e-4, err
print("INBN_BF16_OK %.2e" % err)
'''
    env = dict(os.environ, EMSA_BF16_MFMA='1')
    r = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True,
                       text=True, timeout=300)
    assert 'INBN_BF16_OK' in r.stdout, r.stderr[-2000:]
    # the direct form (EMSA_WGRAD_WINO=0) has no loader fold: bn1_fold() says so up front
    env = dict(os.environ, EMSA_WGRAD_WINO='0')
    r = subprocess.run([sys.executable, '-c',
                        'import torch\nfrom emsanet_amd import functional as Fn\n'
                        'print("FOLD", Fn.bn1_fold(torch.empty(64 << 20)))'],
                       cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert 'FOLD False' in r.stdout, r.stdout + r.stderr[-1000:]


def test_folded_input_bn_rejects_unsupported():
    """3x3 / strided convs have no folded loader: EMSA_E_SHAPE, never a silent plain conv"""
    Fn = _fn()
    from emsanet_amd._lib import EmsaError
    x = to_act(rnd(1, 64, 8, 8, seed=1))
    wt = rnd(64, 64, 3, 3, seed=2, scale=0.1).to(DEV)
    spec = Fn.ConvSpec(64, 64, (3, 3), (1, 1), (1, 1))
    aff = (torch.ones(64, device=DEV), torch.zeros(64, device=DEV))
    with pytest.raises(EmsaError):
        Fn.conv_fwd(x, None, spec, wino_u=Fn.pack_wino(wt)[0], in_affine=aff)
    with pytest.raises(EmsaError):
        Fn.conv_wgrad(x, x, spec, False, like=wt, two_pass=True, in_affine=aff)


@pytest.mark.parametrize('c', [8, 40, 64, 128])
@pytest.mark.parametrize('shape', [(2, 3, 130), (1, 4, 62), (2, 2, 63), (1, 1, 1)])
@pytest.mark.parametrize('with_skip', [False, True])
def test_upsample_dw_forward_row_tiles(c, shape, with_skip):
    """forward of the learned x2 up-sampling on WIDE maps (more columns than one wave covers, odd
    widths, single pixel), with and without skip / bias, against nearest x2 + zero-padded
    depth-wise 3x3 in fp64"""
    Fn = _fn()
    n, h, w = shape
    x = rnd(n, c, h, w, seed=1)
    wt = rnd(c, 1, 3, 3, seed=2)
    b = rnd(c, seed=3)
    skip = rnd(n, c, 2 * h, 2 * w, seed=4) if with_skip else None
    ref = F.conv2d(F.interpolate(x.double(), scale_factor=2, mode='nearest'), wt.double(),
                   b.double(), padding=1, groups=c)
    if with_skip:
        ref = ref + skip.double()
    y = Fn.up2x_dw_fwd(to_act(x), wt.to(DEV), b.to(DEV), to_act(skip) if with_skip else None)
    torch.cuda.synchronize()
    close(y, ref, what='up2x rows')
    y0 = Fn.up2x_dw_fwd(to_act(x), wt.to(DEV), None, None)
    close(y0, ref - b.double()[None, :, None, None] - (skip.double() if with_skip else 0),
          what='up2x rows, no bias')


def test_dropout_masks_in_one_launch():
    """all Dropout2d masks of a step from one launch == the per-layer kernel (and with it the
    oracle's counter-based hash), host seed and device {seed, step} state"""
    Fn = _fn()
    from emsanet_amd import _lib
    n = 5
    layers = [(64, 3, 0.1), (128, 7, 0.2), (40, 11, 0.5), (8, 0, 0.05)]
    jobs = (_lib.EmsaDropoutJob * len(layers))()
    off = 0
    for j, (c, lid, p) in enumerate(layers):
        jobs[j] = _lib.EmsaDropoutJob(off, c, lid, p, 0)
        off += (n * c + 3) // 4 * 4
    table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(DEV)
    seed = 0x1234ABCD
    buf = Fn.dropout2d_mask_batch(table, len(layers), off, n, 128, seed, torch.device(DEV))
    state = torch.tensor([77, 5], dtype=torch.int32, device=DEV)
    buf2 = Fn.dropout2d_mask_batch(table, len(layers), off, n, 128, state, torch.device(DEV))
    o = 0
    for c, lid, p in layers:
        ref = Fn.dropout2d_mask(n, c, p, seed, lid, DEV)
        assert torch.equal(buf[o:o + n * c].view(n, c), ref)
        ref2 = Fn.dropout2d_mask(n, c, p, (77 + 0x632BE5AB * 5) & 0xFFFFFFFF, lid, DEV)
        assert torch.equal(buf2[o:o + n * c].view(n, c), ref2)
        o += (n * c + 3) // 4 * 4
