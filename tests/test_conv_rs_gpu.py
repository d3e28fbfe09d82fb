"""GPU parity of the register-stationary streaming kernel (csrc/conv_rs.hip: the stride-1 3-tap 1-D
convs of the NBt1D blocks, /root/reference/emsanet/model.py:47-58, in 16-bit storage) against plain
PyTorch fp64 references computed from the SAME 16-bit-rounded inputs and weights, through the C-ABI
(emsa_conv1d_rs_t / emsa_conv1d_rs_bnb_t / emsa_pack_weight_frag_t) -- same bars as
tests/test_ops16_gpu.py: bf16 6e-3 / fp16 1e-3 of the tensor magnitude for stored results, 2e-4 /
4e-4 for the fp32 statistics.  Every case asserts that the new kernel took the launch (no silent
fallback to the implicit GEMM) and cross-checks the implicit GEMM on the same inputs.
"""
import pytest
import torch
import torch.nn.functional as F

from util import DEV, close, rnd, to_act

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]
TOL = {torch.bfloat16: 6e-3, torch.float16: 1e-3}

# channels, kernel, n, h, w -- every (C, direction) of the model, image sizes whose tiles are
# ragged in both directions (odd heights / widths, pixel counts off the 32 / 64 / 128-pixel tiles,
# images smaller than a tile row), the /32 map of BASELINE configs[3] (23 x 30)
RS_CONVS = [
    (64, (1, 3), 2, 40, 48),
    (64, (3, 1), 2, 40, 48),
    (64, (1, 3), 3, 23, 30),
    (64, (3, 1), 3, 23, 30),
    (128, (1, 3), 2, 30, 40),
    (128, (3, 1), 2, 30, 40),
    (128, (3, 1), 5, 11, 13),
    (256, (1, 3), 2, 15, 20),
    (256, (3, 1), 2, 15, 20),
    (256, (1, 3), 3, 9, 11),
    (512, (1, 3), 2, 15, 20),
    (512, (3, 1), 2, 15, 20),
    (512, (3, 1), 1, 23, 30),
    (512, (1, 3), 17, 5, 7),
]


def _fn():
    from emsanet_amd import functional as Fn
    return Fn


def q(t, dtype):
    return t.to(dtype).double()


def act16(t_nchw, dtype):
    return t_nchw.to(dtype).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def _spec(Fn, c, k):
    return Fn.ConvSpec(c, c, k, 1, (1, 0) if k == (3, 1) else (0, 1))


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', RS_CONVS)
def test_conv_rs_fwd(cfg, dtype):
    Fn = _fn()
    c, k, n, h, w = cfg
    spec = _spec(Fn, c, k)
    pad = (spec.ph, spec.pw)
    x = rnd(n, c, h, w, seed=1)
    wt = rnd(c, c, *k, seed=2, scale=0.1)
    b = rnd(c, seed=3)
    ref = F.conv2d(q(x, dtype), q(wt, dtype), b.double(), padding=pad)
    wf, _ = Fn.pack_weight_frag_t(wt.to(DEV), dtype, fwd=True)
    wp, _ = Fn.pack_weight_t(wt.to(DEV), dtype, fwd=True)
    xa = act16(x, dtype)
    g = spec.geom_fwd(n, h, w, c, c)
    assert Fn.rs_eligible(spec) and Fn.rs_supported(Fn.dt(xa), g), "the rs kernel must take this case"
    y, stats = Fn.conv_fwd(xa, wp, spec, bias=b.to(DEV), want_stats=True, wfrag=wf)
    torch.cuda.synchronize()
    assert y.dtype == dtype and stats.shape[1] == g._rs[2]
    close(y, ref, tol=TOL[dtype], what='conv_rs')
    # BatchNorm statistics rows (sum, M2 about the row's mean, count) from the fp32 values
    cnt = ref.numel() / c
    assert float(stats[2][:, 0].sum()) == cnt
    mean = stats[0].sum(0) / cnt
    close(mean, ref.mean((0, 2, 3)), tol=2e-4, what='stats mean')
    row_mean = stats[0] / stats[2].clamp(min=1)
    m2 = stats[1].sum(0) + (stats[2] * (row_mean - mean[None]) ** 2).sum(0)
    close(m2 / cnt, ref.var((0, 2, 3), unbiased=False), tol=4e-4, what='stats var')
    # the rows feed emsa_bn_finalize like the implicit GEMM's
    gamma, beta = rnd(c, seed=8).abs() + 0.5, rnd(c, seed=9)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    scale, shift, bmean, invstd = Fn.bn_finalize(stats, int(cnt), gamma.to(DEV), beta.to(DEV), 1e-3, 0.1, rm, rv)
    close(bmean, ref.mean((0, 2, 3)), tol=2e-4, what='bn_finalize mean')
    close(invstd, 1 / torch.sqrt(ref.var((0, 2, 3), unbiased=False) + 1e-3), tol=4e-4, what='bn_finalize invstd')
    # against the implicit GEMM on the same operands: both round the same fp32 sums once
    y0 = Fn.conv_fwd(xa, wp, spec, bias=b.to(DEV))
    close(y, y0.double().cpu(), tol=TOL[dtype], what='conv_rs vs conv_h')
    # fused epilogue: folded BatchNorm + residual + ReLU (an NBt1D block's last conv in eval mode)
    sc, sh = rnd(c, seed=4), rnd(c, seed=5)
    res = rnd(*ref.shape, seed=6)
    y2 = Fn.conv_fwd(xa, wp, spec, bias=b.to(DEV), scale=sc.to(DEV), shift=sh.to(DEV),
                     residual=act16(res, dtype), act=Fn.ACT_RELU, wfrag=wf)
    ref2 = F.relu(ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None] + q(res, dtype))
    close(y2, ref2, tol=TOL[dtype], what='conv_rs epilogue')
    # bias + ReLU (the block's first conv)
    y3 = Fn.conv_fwd(xa, wp, spec, bias=b.to(DEV), act=Fn.ACT_RELU, wfrag=wf)
    close(y3, F.relu(ref), tol=TOL[dtype], what='conv_rs relu')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', RS_CONVS)
def test_conv_rs_dgrad(cfg, dtype):
    Fn = _fn()
    c, k, n, h, w = cfg
    spec = _spec(Fn, c, k)
    x = rnd(n, c, h, w, seed=1).double().requires_grad_(True)
    wt = rnd(c, c, *k, seed=2, scale=0.1)
    y = F.conv2d(x, q(wt, dtype), None, padding=(spec.ph, spec.pw))
    dy = rnd(*y.shape, seed=7)
    y.backward(q(dy, dtype))
    _, wfd = Fn.pack_weight_frag_t(wt.to(DEV), dtype, fwd=False, dgrad=True)
    _, wpd = Fn.pack_weight_t(wt.to(DEV), dtype, fwd=False, dgrad=True)
    dya = act16(dy, dtype)
    g = spec.geom_dgrad(n, h, w, c, c)
    assert Fn.rs_supported(Fn.dt(dya), g)
    dx = Fn.conv_dgrad(dya, wpd, spec, (h, w), wfrag=wfd)
    torch.cuda.synchronize()
    close(dx, x.grad, tol=TOL[dtype], what='rs dgrad')
    mask = rnd(n, c, h, w, seed=8)
    res = rnd(n, c, h, w, seed=9)
    dx2 = Fn.conv_dgrad(dya, wpd, spec, (h, w), mask_src=act16(mask, dtype), wfrag=wfd)
    close(dx2, x.grad * (q(mask, dtype) > 0), tol=TOL[dtype], what='rs dgrad mask')
    dx3 = Fn.conv_dgrad(dya, wpd, spec, (h, w), residual=act16(res, dtype), mask_src=act16(mask, dtype),
                        wfrag=wfd)
    close(dx3, (x.grad + q(res, dtype)) * (q(mask, dtype) > 0), tol=TOL[dtype], what='rs dgrad res+mask')


@pytest.mark.parametrize('cfg', [RS_CONVS[0], RS_CONVS[3], RS_CONVS[5], RS_CONVS[7], RS_CONVS[12]])
@pytest.mark.parametrize('with_res', [False, True])
def test_conv_rs_dgrad_with_fused_bn_backward_sums(cfg, with_res):
    """conv3x1_2 -> bn1 of the NBt1D backward on the rs kernel (emsa_conv1d_rs_bnb_t): == fp64
    autograd of conv(relu(batch_norm(t))) and == the implicit GEMM's fused form"""
    Fn = _fn()
    dtype = torch.bfloat16
    c, k, n, h, w = cfg
    spec = _spec(Fn, c, k)
    pad = (spec.ph, spec.pw)
    qq = lambda t: t.to(dtype).float()   # noqa: E731
    t = qq(rnd(n, c, h, w, seed=1))
    wt = qq(rnd(c, c, *k, seed=2, scale=0.1))
    dy = qq(rnd(n, c, h, w, seed=3))
    res = qq(rnd(n, c, h, w, seed=4)) if with_res else None
    gamma = rnd(c, seed=5) * 0.2 + 1
    beta = rnd(c, seed=6) * 0.3
    eps = 1e-3
    tr = t.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    a = F.relu(F.batch_norm(tr, None, None, gr, br, training=True, eps=eps))
    z = F.conv2d(a, wt.double(), padding=pad)
    loss = (z * dy.double()).sum()
    if with_res:
        loss = loss + (a * res.double()).sum()
    loss.backward()
    gd, bd = gamma.to(DEV), beta.to(DEV)
    xs = t.permute(0, 2, 3, 1).reshape(-1, c)
    stats = torch.stack([xs.sum(0, keepdim=True), ((xs - xs.mean(0)) ** 2).sum(0, keepdim=True),
                         torch.full((1, c), float(xs.shape[0]))]).to(DEV)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    scale, shift, mean, invstd = Fn.bn_finalize(stats, xs.shape[0], gd, bd, eps, 0.1, rm, rv)
    ta, dya = to_act(t).to(dtype), to_act(dy).to(dtype)
    resa = to_act(res).to(dtype) if with_res else None
    wpd = Fn.pack_weight_t(wt.to(DEV), dtype, fwd=False, dgrad=True)[1]
    wfd = Fn.pack_weight_frag_t(wt.to(DEV), dtype, fwd=False, dgrad=True)[1]
    assert Fn.rs_supported(Fn.dt(dya), spec.geom_dgrad(n, h, w, c, c))
    gm, partial, rows = Fn.conv_dgrad_bnb(dya, wpd, spec, (h, w), ta, scale, shift, mean, invstd,
                                          residual=resa, wfrag=wfd)
    dx, dg, db = Fn.bn_bwd_from_rows(gm, ta, gd, mean, invstd, partial, rows, True)
    close(dx, tr.grad, tol=2e-2, what='rs fused dx')
    close(dg, gr.grad, tol=1e-2, what='rs fused dgamma')
    close(db, br.grad, tol=1e-2, what='rs fused dbeta')
    gm0, partial0, rows0 = Fn.conv_dgrad_bnb(dya, wpd, spec, (h, w), ta, scale, shift, mean, invstd,
                                             residual=resa)
    dx0, dg0, db0 = Fn.bn_bwd_from_rows(gm0, ta, gd, mean, invstd, partial0, rows0, True)
    close(dx, dx0.float().cpu(), tol=2e-2, what='rs vs conv_h dx')
    close(dg, dg0.cpu(), tol=1e-2, what='rs vs conv_h dgamma')
    close(db, db0.cpu(), tol=1e-2, what='rs vs conv_h dbeta')


def test_conv_rs_tiny_maps_and_refusals():
    """a map with fewer pixel tiles than persistent workgroups (batch-1 inference at /32; here ONE
    tile) runs on the rs kernel -- the surplus workgroups write empty statistics rows -- and gives
    the reference's result; geometries outside the kernel (strided, 3x3, other channel counts) are
    reported unsupported and a direct call is rejected with EMSA_E_SHAPE, never computed wrongly"""
    from emsanet_amd import _lib
    Fn = _fn()
    dtype = torch.bfloat16
    for c, k, n, h, w in ((64, (1, 3), 1, 4, 6), (512, (3, 1), 1, 5, 4), (256, (1, 3), 1, 3, 3)):
        spec = _spec(Fn, c, k)
        g = spec.geom_fwd(n, h, w, c, c)
        assert Fn.rs_supported(1, g)
        x = rnd(n, c, h, w, seed=1)
        wt = rnd(c, c, *k, seed=2, scale=0.1)
        b = rnd(c, seed=3)
        wf, _ = Fn.pack_weight_frag_t(wt.to(DEV), dtype, fwd=True)
        wp, _ = Fn.pack_weight_t(wt.to(DEV), dtype, fwd=True)
        y, stats = Fn.conv_fwd(act16(x, dtype), wp, spec, bias=b.to(DEV), want_stats=True, wfrag=wf)
        ref = F.conv2d(q(x, dtype), q(wt, dtype), b.double(), padding=(spec.ph, spec.pw))
        close(y, ref, tol=TOL[dtype], what='tiny map')
        cnt = ref.numel() / c
        assert float(stats[2][:, 0].sum()) == cnt and int((stats[2][:, 0] > 0).sum()) >= 1
        close(stats[0].sum(0) / cnt, ref.mean((0, 2, 3)), tol=2e-4, what='tiny map stats')
    # refusals
    s2 = Fn.ConvSpec(64, 128, (3, 1), (2, 1), (1, 0))
    assert not Fn.rs_eligible(s2) and not Fn.rs_eligible(Fn.ConvSpec(72, 72, (1, 3), 1, (0, 1)))
    assert not Fn.rs_eligible(Fn.ConvSpec(64, 64, (3, 3), 1, 1))
    g2 = Fn.ConvSpec(64, 64, (3, 1), (2, 1), (1, 0)).geom_fwd(2, 12, 20, 64, 64)      # strided
    assert _lib.lib().emsa_conv1d_rs_supported(1, g2) == 0
    x = act16(rnd(2, 64, 12, 20, seed=1), dtype)
    wf, _ = Fn.pack_weight_frag_t(rnd(64, 64, 3, 1, seed=2).to(DEV), dtype, fwd=True)
    out = torch.empty(2 * 6 * 20 * 64, device=DEV, dtype=dtype)
    rc = _lib.lib().emsa_conv1d_rs_t(1, g2, x.data_ptr(), wf.data_ptr(), out.data_ptr(),
                                     None, None, None, None, None, 0, None, 0, 0, None)
    assert rc == -1


@pytest.mark.parametrize('cfg', [(64, (1, 3), 2, 24, 40), (128, (3, 1), 2, 24, 40), (256, (1, 3), 2, 12, 20),
                                 (512, (3, 1), 2, 12, 20)])
def test_conv_rs_channel_slices(cfg):
    """input, output, residual and mask tensors that are channel slices of wider NHWC buffers
    (pixel stride > C: concat / split without copies): bit-identical to the dense launch,
    and the neighbouring channels of the output buffer stay untouched"""
    Fn = _fn()
    dtype = torch.bfloat16
    c, k, n, h, w = cfg
    spec = _spec(Fn, c, k)
    x = rnd(n, c, h, w, seed=1)
    wt = rnd(c, c, *k, seed=2, scale=0.1)
    b = rnd(c, seed=3)
    res, mask = rnd(n, c, h, w, seed=4), rnd(n, c, h, w, seed=5)
    wf, wfd = Fn.pack_weight_frag_t(wt.to(DEV), dtype, fwd=True, dgrad=True)
    wp, wpd = Fn.pack_weight_t(wt.to(DEV), dtype, fwd=True, dgrad=True)
    xd, rd, md = act16(x, dtype), act16(res, dtype), act16(mask, dtype)

    def wide(t, ld, off):
        buf = torch.full((n, h, w, ld), 7.0, device=DEV, dtype=dtype)
        buf[..., off:off + c] = t.permute(0, 2, 3, 1)
        return buf, buf[..., off:off + c].permute(0, 3, 1, 2)

    _, xs = wide(xd, c + 64, 32)
    _, rs_ = wide(rd, c + 8, 8)
    _, ms = wide(md, 2 * c, c)
    obuf = torch.full((n, h, w, c + 16), 3.0, device=DEV, dtype=dtype)
    outs = obuf[..., 8:8 + c].permute(0, 3, 1, 2)
    assert Fn.ld_of(xs) == c + 64 and Fn.ld_of(outs) == c + 16
    assert Fn.rs_supported(1, spec.geom_fwd(n, h, w, c + 64, c + 16))
    y_dense = Fn.conv_fwd(xd, wp, spec, bias=b.to(DEV), residual=rd, act=Fn.ACT_RELU, wfrag=wf)
    Fn.conv_fwd(xs, wp, spec, bias=b.to(DEV), residual=rs_, act=Fn.ACT_RELU, wfrag=wf, out=outs)
    torch.cuda.synchronize()
    assert torch.equal(outs, y_dense)
    assert bool((obuf[..., :8] == 3.0).all()) and bool((obuf[..., 8 + c:] == 3.0).all())
    d_dense = Fn.conv_dgrad(xd, wpd, spec, (h, w), mask_src=md, residual=rd, wfrag=wfd)
    obuf.fill_(3.0)
    Fn.conv_dgrad(xs, wpd, spec, (h, w), mask_src=ms, residual=rs_, wfrag=wfd, out=outs)
    torch.cuda.synchronize()
    assert torch.equal(outs, d_dense)
    assert bool((obuf[..., :8] == 3.0).all()) and bool((obuf[..., 8 + c:] == 3.0).all())
    close(y_dense, F.relu(F.conv2d(q(x, dtype), q(wt, dtype), b.double(), padding=(spec.ph, spec.pw))
                          + q(res, dtype)), tol=TOL[dtype], what='sliced conv_rs')


def test_conv_rs_switch(monkeypatch):
    """Fn.CONV_RS = False (EMSA_CONV_RS=0) keeps every conv on the implicit GEMM"""
    Fn = _fn()
    monkeypatch.setattr(Fn, 'CONV_RS', False)
    spec = _spec(Fn, 64, (1, 3))
    assert not Fn.rs_supported(1, spec.geom_fwd(2, 40, 48, 64, 64))


# batch-32 maps (every persistent workgroup busy in both halves) next to the ragged small ones
PAIR_CONVS = RS_CONVS + [(64, (1, 3), 8, 120, 160), (128, (3, 1), 8, 60, 80), (512, (1, 3), 32, 15, 20)]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', PAIR_CONVS)
def test_conv_rs_pair_is_two_launches(cfg, dtype):
    """emsa_conv1d_rs_pair_t (twin launch, grid.y = 2: the rgb | depth encoder blocks and the
    semantic | instance decoder blocks of /root/reference/emsanet/model.py:95-160): each half is
    bit-identical to its own emsa_conv1d_rs_t launch, for the three epilogues of an eval-mode
    NBt1D block (bias+ReLU, bias+folded BN+ReLU, bias+folded BN+residual+ReLU) and the plain one"""
    Fn = _fn()
    c, k, n, h, w = cfg
    spec = _spec(Fn, c, k)
    xs = [act16(rnd(n, c, h, w, seed=11 + i), dtype) for i in range(2)]
    wts = [rnd(c, c, *k, seed=21 + i, scale=0.1).to(DEV) for i in range(2)]
    wfs = [Fn.pack_weight_frag_t(wt, dtype, fwd=True)[0] for wt in wts]
    wps = [Fn.pack_weight_t(wt, dtype, fwd=True)[0] for wt in wts]
    bs = [rnd(c, seed=31 + i).to(DEV) for i in range(2)]
    scs = [rnd(c, seed=41 + i).to(DEV) for i in range(2)]
    shs = [rnd(c, seed=51 + i).to(DEV) for i in range(2)]
    rs = [act16(rnd(n, c, h, w, seed=61 + i), dtype) for i in range(2)]
    N2 = (None, None)
    cases = [dict(), dict(biases=bs, act=Fn.ACT_RELU),
             dict(biases=bs, scales=scs, shifts=shs, act=Fn.ACT_RELU),
             dict(biases=bs, scales=scs, shifts=shs, residuals=rs, act=Fn.ACT_RELU)]
    for kw in cases:
        pair = Fn.conv_fwd_pair(xs, wfs, spec, **kw)
        assert pair is not None, "the twin launch must take this case"
        torch.cuda.synchronize()
        for i in range(2):
            one = Fn.conv_fwd(xs[i], wps[i], spec, bias=kw.get('biases', N2)[i],
                              scale=kw.get('scales', N2)[i], shift=kw.get('shifts', N2)[i],
                              residual=kw.get('residuals', N2)[i], act=kw.get('act', Fn.ACT_NONE),
                              wfrag=wfs[i])
            assert torch.equal(pair[i], one), f"half {i} of the twin launch, case {sorted(kw)}"
    # operands given for one half only / other geometry: refused (the caller launches twice)
    assert Fn.conv_fwd_pair(xs, wfs, spec, biases=(bs[0], None)) is None
    assert Fn.conv_fwd_pair(xs, (wfs[0], None), spec) is None
    rc = _lib_rc_pair_bad_args(Fn, xs, wfs, spec)
    assert rc != 0


def _lib_rc_pair_bad_args(Fn, xs, wfs, spec):
    """the C entry point itself refuses a residual for one half only"""
    from emsanet_amd import _lib
    n, c, h, w = xs[0].shape
    g = spec.geom_fwd(n, h, w, c, c)
    out = [torch.empty_like(x) for x in xs]
    p = lambda t: None if t is None else t.data_ptr()   # noqa: E731
    return _lib.lib().emsa_conv1d_rs_pair_t(Fn.dt(xs[0]), g, p(xs[0]), p(xs[1]), p(wfs[0]), p(wfs[1]),
                                            p(out[0]), p(out[1]), None, None, None, None, None, None,
                                            p(xs[0]), None, c, 0, torch.cuda.current_stream().cuda_stream)


# cin, cout, kernel, stride, padding, n, h, w -- the convs of the twin eval path that the implicit GEMM
# runs: the strided 3x1 / 1x3 and 1x1 down-sampling convs of the first block of an encoder stage, the
# 3x3 and 1x1 skip-fusion convs of the decoder modules (batch 1: tap-split form at 512 -> 512)
IGEMM_PAIRS = [
    (64, 128, (3, 1), (2, 1), (1, 0), 1, 120, 160),
    (128, 128, (1, 3), (1, 2), (0, 1), 1, 60, 160),
    (64, 128, (1, 1), (2, 2), (0, 0), 1, 120, 160),
    (512, 512, (3, 3), (1, 1), (1, 1), 1, 15, 20),
    (512, 256, (3, 3), (1, 1), (1, 1), 1, 30, 40),
    (256, 512, (1, 1), (1, 1), (0, 0), 2, 30, 40),
    (256, 128, (3, 3), (1, 1), (1, 1), 3, 23, 31),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', IGEMM_PAIRS)
def test_conv_igemm_pair_is_two_launches(cfg, dtype):
    """emsa_conv_igemm_pair_t: each half of the twin launch == its own emsa_conv_igemm_t /
    emsa_conv_igemm_splitk_t launch bit for bit (folded BatchNorm + ReLU, + residual, plain; shared
    input as the decoders' skip-fusion convs have it)"""
    Fn = _fn()
    cin, cout, k, st, pad, n, h, w = cfg
    spec = Fn.ConvSpec(cin, cout, k, st, pad)
    xs = [act16(rnd(n, cin, h, w, seed=11 + i), dtype) for i in range(2)]
    wps = [Fn.pack_weight_t(rnd(cout, cin, *k, seed=21 + i, scale=0.1).to(DEV), dtype, fwd=True)[0]
           for i in range(2)]
    oh, ow = spec.out_hw(h, w)
    bs = [rnd(cout, seed=31 + i).to(DEV) for i in range(2)]
    scs = [rnd(cout, seed=41 + i).to(DEV) for i in range(2)]
    shs = [rnd(cout, seed=51 + i).to(DEV) for i in range(2)]
    rs = [act16(rnd(n, cout, oh, ow, seed=61 + i), dtype) for i in range(2)]
    N2 = (None, None)
    cases = [(xs, dict()), (xs, dict(scales=scs, shifts=shs, act=Fn.ACT_RELU)),
             (xs, dict(biases=bs, scales=scs, shifts=shs, residuals=rs, act=Fn.ACT_RELU)),
             ([xs[0], xs[0]], dict(scales=scs, shifts=shs, act=Fn.ACT_RELU))]
    for xin, kw in cases:
        pair = Fn.conv_igemm_pair(xin, wps, spec, **kw)
        assert pair is not None, "the twin launch must take this case"
        torch.cuda.synchronize()
        for i in range(2):
            one = Fn.conv_fwd(xin[i], wps[i], spec, bias=kw.get('biases', N2)[i],
                              scale=kw.get('scales', N2)[i], shift=kw.get('shifts', N2)[i],
                              residual=kw.get('residuals', N2)[i], act=kw.get('act', Fn.ACT_NONE))
            assert torch.equal(pair[i], one), f"half {i} of the twin launch, case {sorted(kw)}"
    assert Fn.conv_igemm_pair(xs, wps, spec, scales=(scs[0], None), shifts=(shs[0], None)) is None
    x32 = [x.float() for x in xs]
    assert Fn.conv_igemm_pair(x32, wps, spec) is None          # fp32 storage: two launches


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', [(512, 1, 15, 20), (256, 1, 30, 40), (128, 2, 23, 31), (64, 3, 7, 5)])
def test_up2x_pair_is_two_launches(cfg, dtype):
    """emsa_up2x_dw3x3_fwd_pair_t: the learned x2 up-sampling (+ skip) of the two decoders in one
    launch, each half == emsa_up2x_dw3x3_fwd_t bit for bit"""
    Fn = _fn()
    c, n, h, w = cfg
    xs = [act16(rnd(n, c, h, w, seed=71 + i), dtype) for i in range(2)]
    sk = [act16(rnd(n, c, 2 * h, 2 * w, seed=73 + i), dtype) for i in range(2)]
    ws = [rnd(c, 1, 3, 3, seed=75 + i).to(DEV).contiguous() for i in range(2)]
    bs = [rnd(c, seed=77 + i).to(DEV) for i in range(2)]
    for biases, skips in (((None, None), (None, None)), (bs, sk), ((None, None), sk)):
        pair = Fn.up2x_dw_fwd_pair(xs, ws, biases, skips)
        assert pair is not None
        torch.cuda.synchronize()
        for i in range(2):
            one = Fn.up2x_dw_fwd(xs[i], ws[i], biases[i], skips[i])
            assert torch.equal(pair[i], one)
    assert Fn.up2x_dw_fwd_pair(xs, ws, (bs[0], None), (None, None)) is None
    assert Fn.up2x_dw_fwd_pair([x.float() for x in xs], ws, (None, None), (None, None)) is None


# channels, n, h, w: the /4 and /8 maps of the batch-1 forward (120 x 160, 60 x 80), widths off the
# 32-pixel blocks, one-row and one-column maps, a second image
HALF_BLOCKS = [(64, 1, 120, 160), (128, 1, 60, 80), (64, 2, 7, 33), (128, 3, 5, 30), (64, 1, 1, 9),
               (128, 1, 9, 1), (64, 1, 23, 240)]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('twin', [False, True])
@pytest.mark.parametrize('cfg', HALF_BLOCKS)
def test_nbt_half_block_is_two_launches(cfg, twin, dtype):
    """the fused half-block (csrc/conv_hb.hip: conv3x1 + ReLU -> conv1x3 + folded BatchNorm (+ residual)
    + ReLU, /root/reference/emsanet/model.py:47-58) == the two emsa_conv1d_rs_t launches it replaces,
    bit for bit, with and without the residual / the affine, for one tensor set and for a twin pair;
    and both against the fp64 reference that rounds the intermediate tensor like the engine stores it"""
    Fn = _fn()
    c, n, h, w = cfg
    sa, sb = _spec(Fn, c, (3, 1)), _spec(Fn, c, (1, 3))
    assert Fn._lib.lib().emsa_nbt_half_block_supported(Fn.DT[dtype], c, w) == 1
    sets = []
    for s in range(2 if twin else 1):
        x = rnd(n, c, h, w, seed=10 + s)
        wa, wb = rnd(c, c, 3, 1, seed=20 + s, scale=0.1), rnd(c, c, 1, 3, seed=30 + s, scale=0.1)
        ba, bb = rnd(c, seed=40 + s), rnd(c, seed=50 + s)
        sc, sh = rnd(c, seed=60 + s).abs() + 0.5, rnd(c, seed=70 + s)
        res = rnd(n, c, h, w, seed=80 + s)
        sets.append(dict(x=x, wa=wa, wb=wb, ba=ba, bb=bb, sc=sc, sh=sh, res=res,
                         xa=act16(x, dtype), ra=act16(res, dtype),
                         wfa=Fn.pack_weight_frag_t(wa.to(DEV), dtype, fwd=True)[0],
                         wfb=Fn.pack_weight_frag_t(wb.to(DEV), dtype, fwd=True)[0],
                         wpa=Fn.pack_weight_t(wa.to(DEV), dtype, fwd=True)[0],
                         wpb=Fn.pack_weight_t(wb.to(DEV), dtype, fwd=True)[0]))
    # (maps conv_rs refuses -- fewer pixels than its smallest plan -- run the two-launch twin on the
    #  implicit GEMM, whose accumulation order differs: storage tolerance instead of bit equality)
    code = Fn.dt(sets[0]['xa'])
    bitwise = Fn.rs_supported(code, sa.geom_fwd(n, h, w, c, c)) and Fn.rs_supported(code, sb.geom_fwd(n, h, w, c, c))
    for with_res, with_affine in ((True, True), (False, True), (False, False)):
        outs = Fn.nbt_half_block(
            [s['xa'] for s in sets], [s['wfa'] for s in sets], [s['ba'].to(DEV) for s in sets],
            [s['wfb'] for s in sets], [s['bb'].to(DEV) for s in sets],
            [s['sc'].to(DEV) if with_affine else None for s in sets],
            [s['sh'].to(DEV) if with_affine else None for s in sets],
            [s['ra'] if with_res else None for s in sets], Fn.ACT_RELU)
        torch.cuda.synchronize()
        for s, out in zip(sets, outs):
            y1 = Fn.conv_fwd(s['xa'], s['wpa'], sa, bias=s['ba'].to(DEV), act=Fn.ACT_RELU, wfrag=s['wfa'])
            two = Fn.conv_fwd(y1, s['wpb'], sb, bias=s['bb'].to(DEV),
                              scale=s['sc'].to(DEV) if with_affine else None,
                              shift=s['sh'].to(DEV) if with_affine else None,
                              residual=s['ra'] if with_res else None, act=Fn.ACT_RELU, wfrag=s['wfb'])
            torch.cuda.synchronize()
            assert out.dtype == dtype
            if bitwise:
                assert torch.equal(out, two), (cfg, with_res, with_affine)
            else:
                close(out, two.double().cpu(), tol=TOL[dtype], what=f'half-block vs two launches {cfg}')
            mid = F.relu(F.conv2d(q(s['x'], dtype), q(s['wa'], dtype), s['ba'].double(), padding=(1, 0)))
            ref = F.conv2d(q(mid, dtype), q(s['wb'], dtype), s['bb'].double(), padding=(0, 1))
            if with_affine:
                ref = ref * s['sc'].double().view(1, -1, 1, 1) + s['sh'].double().view(1, -1, 1, 1)
            if with_res:
                ref = ref + q(s['res'], dtype)
            close(out, F.relu(ref), tol=TOL[dtype], what=f'half-block {cfg}')


def test_nbt_half_block_refusals():
    Fn = _fn()
    L = Fn._lib.lib()
    assert L.emsa_nbt_half_block_supported(Fn.DT[torch.bfloat16], 256, 40) == 0      # channels
    assert L.emsa_nbt_half_block_supported(Fn.DT[torch.bfloat16], 64, 4000) == 0     # row does not fit LDS
    assert L.emsa_nbt_half_block_supported(0, 64, 40) == 0                           # fp32 storage
    x = act16(rnd(1, 64, 4, 8, seed=1), torch.bfloat16)
    assert not Fn.half_block_ok(x.float(), 64)
