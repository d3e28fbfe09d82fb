"""GPU parity of the 16-bit kernel family (BASELINE configs[2] bf16 training, configs[4] 16-bit
inference) against plain PyTorch CPU references in fp64.

The reference of every op is computed in fp64 FROM THE SAME 16-BIT-ROUNDED INPUTS (and, for the
convolutions, 16-bit-rounded weights) the kernel reads, so what is left is the kernel's own error:
fp32 accumulation (negligible) + ONE rounding of the result to the storage type.  Stated tolerance:
    bf16 storage: 2^-8 relative per element  -> 6e-3 of the tensor's max magnitude
    fp16 storage: 2^-11                      -> 1e-3
    fp32 results computed from 16-bit inputs (statistics, weight gradients, SE vectors): 2e-4.
"""
import pytest
import torch
import torch.nn.functional as F

from util import DEV, close, rnd

pytestmark = pytest.mark.gpu

DTYPES = [torch.bfloat16, torch.float16]
TOL = {torch.bfloat16: 6e-3, torch.float16: 1e-3}


def _fn():
    from emsanet_amd import functional as Fn
    return Fn


def q(t, dtype):
    """value of `t` after rounding to `dtype`, as fp64 on the CPU"""
    return t.to(dtype).double()


def act16(t_nchw, dtype):
    """CPU NCHW fp32 tensor -> GPU activation in `dtype` (logical NCHW, NHWC memory)"""
    return t_nchw.to(dtype).to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


CONVS = [
    # cin, cout, kernel, stride, padding, n, h, w
    (64, 64, (3, 1), (1, 1), (1, 0), 2, 12, 20),
    (64, 64, (1, 3), (1, 1), (0, 1), 2, 12, 20),
    (128, 128, (1, 3), (1, 1), (0, 1), 3, 9, 13),
    (64, 128, (3, 1), (2, 1), (1, 0), 2, 12, 20),
    (128, 128, (1, 3), (1, 2), (0, 1), 2, 6, 20),
    (64, 128, (1, 1), (2, 2), (0, 0), 2, 12, 20),
    (64, 128, (3, 1), (2, 1), (1, 0), 2, 15, 20),      # odd height: data-gradient phases of unequal length
    (16, 32, (3, 3), (2, 2), (1, 1), 1, 9, 11),        # four phases with 4 / 2 / 2 / 1 taps
    (256, 128, (3, 3), (1, 1), (1, 1), 2, 8, 10),
    (128, 40, (3, 3), (1, 1), (1, 1), 2, 8, 10),
    (96, 8, (3, 3), (1, 1), (1, 1), 2, 8, 10),
    (512, 256, (1, 1), (1, 1), (0, 0), 2, 5, 5),
    (256, 16, (1, 1), (1, 1), (0, 0), 4, 1, 1),
    (512, 512, (3, 1), (1, 1), (1, 0), 1, 15, 20),
    # ragged: channel counts off the 64-wide tiles / 64-deep K steps, odd and tiny images
    (40, 72, (1, 3), (1, 1), (0, 1), 1, 5, 7),
    (72, 40, (3, 1), (1, 1), (1, 0), 2, 7, 5),
    (8, 8, (1, 3), (1, 1), (0, 1), 1, 1, 2),
    (24, 136, (3, 3), (1, 1), (1, 1), 1, 5, 3),
    # the bf16-MFMA weight-gradient modes of round 3: 1x1 (stride 1 / 2) and stride-2 3-tap convs,
    # several 64-pixel K steps, odd line lengths (the odd-pixel image runs past the line's end)
    (64, 64, (1, 1), (1, 1), (0, 0), 2, 30, 40),
    (128, 256, (1, 1), (2, 2), (0, 0), 2, 15, 21),
    (128, 256, (3, 1), (2, 1), (1, 0), 2, 30, 40),
    (128, 256, (1, 3), (1, 2), (0, 1), 2, 15, 41),
    (72, 40, (1, 3), (1, 2), (0, 1), 1, 3, 9),
    # enough pixels for the 128-row tiles
    (64, 64, (1, 3), (1, 1), (0, 1), 4, 120, 160),
    (128, 128, (3, 1), (1, 1), (1, 0), 4, 60, 80),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('tile,staging', [(-1, 0), (0, 0), (1, 0), (2, 0), (-1, 1), (1, 1), (2, 2)])
@pytest.mark.parametrize('cfg', CONVS)
def test_conv16_fwd(cfg, tile, staging, dtype, monkeypatch):
    """staging: how the K steps reach LDS -- 0 = LDS-DMA into two buffers (default), 1 = register
    prefetch + one padded buffer, 2 = two register sets (EMSA_CONVH_PF)"""
    Fn = _fn()
    if tile >= 0:
        monkeypatch.setenv('EMSA_CONVH_TILE', str(tile))
    monkeypatch.setenv('EMSA_CONVH_PF', str(staging))
    cin, cout, k, s, p, n, h, w = cfg
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, *k, seed=2, scale=0.1)
    b = rnd(cout, seed=3)
    ref = F.conv2d(q(x, dtype), q(wt, dtype), b.double(), stride=s, padding=p)
    spec = Fn.ConvSpec(cin, cout, k, s, p)
    wp, _ = Fn.pack_weight_t(wt.to(DEV), dtype, fwd=True)
    y, stats = Fn.conv_fwd(act16(x, dtype), wp, spec, bias=b.to(DEV), want_stats=True)
    torch.cuda.synchronize()
    assert y.dtype == dtype
    close(y, ref, tol=TOL[dtype], what='conv16')
    # BatchNorm statistics come from the fp32 accumulators (before the rounding of the output)
    cnt = ref.numel() / cout
    assert float(stats[2][:, 0].sum()) == cnt
    mean = stats[0].sum(0) / cnt
    close(mean, ref.mean((0, 2, 3)), tol=2e-4, what='stats mean')
    tile_mean = stats[0] / stats[2]
    m2 = stats[1].sum(0) + (stats[2] * (tile_mean - mean[None]) ** 2).sum(0)
    close(m2 / cnt, ref.var((0, 2, 3), unbiased=False), tol=4e-4, what='stats var')
    # fused epilogue: folded BN + residual + relu
    sc, sh = rnd(cout, seed=4), rnd(cout, seed=5)
    res = rnd(*ref.shape, seed=6)
    y2 = Fn.conv_fwd(act16(x, dtype), wp, spec, bias=b.to(DEV), scale=sc.to(DEV), shift=sh.to(DEV),
                     residual=act16(res, dtype), act=Fn.ACT_RELU)
    ref2 = F.relu(ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
                  + q(res, dtype))
    close(y2, ref2, tol=TOL[dtype], what='conv16 epilogue')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('cfg', [(512, 512, (3, 3), 1, 15, 20), (512, 256, (3, 3), 1, 30, 40),
                                 (256, 128, (3, 3), 1, 60, 80), (128, 96, (3, 3), 2, 9, 11),
                                 (512, 512, (3, 1), 1, 5, 7)])
def test_conv16_tap_split(cfg, dtype, monkeypatch):
    """few output tiles, long K (the decoders' 3x3 convs at batch 1, BASELINE configs[4]): the
    forward conv runs tap-split (emsa_conv_igemm_splitk_t: ksplit workgroups per tile + a finish
    pass) -- same result as the plain launch within one rounding, epilogue included"""
    Fn = _fn()
    cin, cout, k, n, h, w = cfg
    p_ = (k[0] // 2, k[1] // 2)
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, *k, seed=2, scale=0.05)
    b = rnd(cout, seed=3)
    sc, sh = rnd(cout, seed=4), rnd(cout, seed=5)
    spec = Fn.ConvSpec(cin, cout, k, 1, p_)
    g = spec.geom_fwd(n, h, w, cin, cout)
    assert Fn._splitk_ws_bytes(Fn.DT[dtype], g) > 0, "this case must take the tap-split path"
    ref = F.conv2d(q(x, dtype), q(wt, dtype), b.double(), padding=p_)
    res = rnd(*ref.shape, seed=6)
    wp, _ = Fn.pack_weight_t(wt.to(DEV), dtype, fwd=True)
    xa = act16(x, dtype)
    y = Fn.conv_fwd(xa, wp, spec, bias=b.to(DEV))
    close(y, ref, tol=TOL[dtype], what='tap-split conv')
    y2 = Fn.conv_fwd(xa, wp, spec, bias=b.to(DEV), scale=sc.to(DEV), shift=sh.to(DEV),
                     residual=act16(res, dtype), act=Fn.ACT_RELU)
    ref2 = F.relu(ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None] + q(res, dtype))
    close(y2, ref2, tol=TOL[dtype], what='tap-split conv epilogue')
    # the plain launch on the same operands (EMSA_CONVH_SPLITK is read once per process: compare
    # through the statistics-carrying call, which never splits)
    y0, _ = Fn.conv_fwd(xa, wp, spec, bias=b.to(DEV), want_stats=True)
    close(y, y0.double().cpu(), tol=TOL[dtype], what='tap-split vs plain')


@pytest.mark.parametrize('cfg', [CONVS[0], CONVS[-2], CONVS[-1]])
def test_conv16_k_step_choice(cfg, monkeypatch):
    """short-K convs (<= 6 steps of 64 channels over all taps) run with 32-channel K steps under
    LDS-DMA staging: same accumulation order, so the result equals the 64-channel-step kernel's
    bit for bit (EMSA_CONVH_SHORTK=0)"""
    Fn = _fn()
    dtype = torch.bfloat16
    cin, cout, k, s_, p_, n, h, w = cfg
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, *k, seed=2, scale=0.1)
    b = rnd(cout, seed=3)
    spec = Fn.ConvSpec(cin, cout, k, s_, p_)
    wp, _ = Fn.pack_weight_t(wt.to(DEV), dtype, fwd=True)
    monkeypatch.setenv('EMSA_CONVH_PF', '0')
    y32, st32 = Fn.conv_fwd(act16(x, dtype), wp, spec, bias=b.to(DEV), want_stats=True)
    monkeypatch.setenv('EMSA_CONVH_SHORTK', '0')
    y64, st64 = Fn.conv_fwd(act16(x, dtype), wp, spec, bias=b.to(DEV), want_stats=True)
    torch.cuda.synchronize()
    assert torch.equal(y32, y64) and torch.equal(st32, st64)
    ref = F.conv2d(q(x, dtype), q(wt, dtype), b.double(), stride=s_, padding=p_)
    close(y32, ref, tol=TOL[dtype], what='conv16 (k-step choice)')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('tile,staging', [(-1, 0), (1, 0), (-1, 1)])
@pytest.mark.parametrize('cfg', CONVS)
def test_conv16_dgrad(cfg, tile, staging, dtype, monkeypatch):
    Fn = _fn()
    if tile >= 0:
        monkeypatch.setenv('EMSA_CONVH_TILE', str(tile))
    monkeypatch.setenv('EMSA_CONVH_PF', str(staging))
    # strided convs: one launch over all taps (the 16-bit default) and, with the default tile and
    # staging, one launch per output phase through the output pixel map
    monkeypatch.setattr(Fn, 'DGRAD_PHASES', tile == -1 and staging == 0)
    cin, cout, k, s, p, n, h, w = cfg
    x = rnd(n, cin, h, w, seed=1).double().requires_grad_(True)
    wt = rnd(cout, cin, *k, seed=2, scale=0.1)
    y = F.conv2d(x, q(wt, dtype), None, stride=s, padding=p)
    dy = rnd(*y.shape, seed=7)
    y.backward(q(dy, dtype))
    spec = Fn.ConvSpec(cin, cout, k, s, p)
    _, wpd = Fn.pack_weight_t(wt.to(DEV), dtype, fwd=False, dgrad=True)
    dx = Fn.conv_dgrad(act16(dy, dtype), wpd, spec, (h, w))
    torch.cuda.synchronize()
    close(dx, x.grad, tol=TOL[dtype], what='dgrad16')
    mask = rnd(n, cin, h, w, seed=8)
    res = rnd(n, cin, h, w, seed=9)
    dx2 = Fn.conv_dgrad(act16(dy, dtype), wpd, spec, (h, w), mask_src=act16(mask, dtype))
    close(dx2, x.grad * (q(mask, dtype) > 0), tol=TOL[dtype], what='dgrad16 mask')
    dx3 = Fn.conv_dgrad(act16(dy, dtype), wpd, spec, (h, w), residual=act16(res, dtype))
    close(dx3, x.grad + q(res, dtype), tol=TOL[dtype], what='dgrad16 residual')


@pytest.mark.parametrize('cfg', CONVS)
def test_conv16_wgrad(cfg):
    """fp32 weight / bias gradients from bf16 activations"""
    Fn = _fn()
    dtype = torch.bfloat16
    cin, cout, k, s, p, n, h, w = cfg
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, *k, seed=2, scale=0.1).double().requires_grad_(True)
    b = rnd(cout, seed=3).double().requires_grad_(True)
    y = F.conv2d(q(x, dtype), wt, b, stride=s, padding=p)
    dy = rnd(*y.shape, seed=7)
    y.backward(q(dy, dtype))
    spec = Fn.ConvSpec(cin, cout, k, s, p)
    like = torch.empty(cout, cin, *k, device=DEV)
    for two_pass in (True, False):
        dw, db, packed = Fn.conv_wgrad(act16(x, dtype), act16(dy, dtype), spec, True, like=like,
                                       two_pass=two_pass)
        if packed:
            dw = Fn.unpack_wgrad(dw, like)
        torch.cuda.synchronize()
        # direct products of the stored bf16 values, fp32 accumulation: fp32-kernel accuracy
        close(dw, wt.grad, tol=2e-4, what=f'wgrad16 (two_pass={two_pass})')
        close(db, b.grad, tol=2e-4, what='dbias16')


@pytest.mark.parametrize('c,n,h,w,n_jobs', [(64, 2, 24, 40, 4), (128, 2, 13, 21, 4), (256, 3, 15, 20, 4),
                                            (512, 2, 15, 20, 4), (512, 1, 8, 9, 3), (64, 1, 3, 5, 2),
                                            (256, 32, 30, 40, 4)])
def test_conv16_wgrad_multi(c, n, h, w, n_jobs):
    """the weight gradients of an NBt1D block's convs in ONE launch (emsa_conv_wgrad_multi_t: grid.y =
    job, one split-K budget; ref emsanet/model.py:47-58 -- conv3x1, conv1x3, conv3x1, conv1x3 of the
    same channel count) against the fp64 reference of every job, written straight into `dw_out` /
    `db_out` views like the flat gradient buckets hand out; odd lines, a job without a bias"""
    Fn = _fn()
    dtype = torch.bfloat16
    kinds = [((3, 1), (1, 0)), ((1, 3), (0, 1)), ((3, 1), (1, 0)), ((1, 3), (0, 1))][:n_jobs]
    jobs, refs = [], []
    for j, (k, p) in enumerate(kinds):
        x = rnd(n, c, h, w, seed=10 + j)
        dy = rnd(n, c, h, w, seed=20 + j)
        wt = torch.zeros(c, c, *k, dtype=torch.float64, requires_grad=True)
        b = torch.zeros(c, dtype=torch.float64, requires_grad=True)
        F.conv2d(q(x, dtype), wt, b, padding=p).backward(q(dy, dtype))
        spec = Fn.ConvSpec(c, c, k, (1, 1), p)
        like = torch.empty(c, c, *k, device=DEV)
        want_bias = j != 2
        flat = torch.full((like.numel() + c + 8,), float('nan'), device=DEV)
        dw_out = flat[:like.numel()].view_as(like) if j != 1 else None       # (job 1: own buffers)
        db_out = flat[like.numel() + 4:like.numel() + 4 + c] if (want_bias and j != 1) else None
        jobs.append((act16(x, dtype), act16(dy, dtype), spec, like, dw_out, db_out, want_bias))
        refs.append((wt.grad, b.grad if want_bias else None, flat, dw_out))
    out = Fn.conv_wgrad_multi(jobs)
    torch.cuda.synchronize()
    assert out is not None and len(out) == n_jobs
    for j, ((dw, db), (rw, rb, flat, dw_out)) in enumerate(zip(out, refs)):
        close(dw, rw, tol=2e-4, what=f'multi wgrad job {j}')
        if rb is not None:
            close(db, rb, tol=2e-4, what=f'multi dbias job {j}')
        else:
            assert db is None
        if dw_out is not None:
            assert dw.data_ptr() == dw_out.data_ptr()
            assert bool(torch.isnan(flat[-4:]).all())                       # nothing written past the views
        # and what one launch per job gives (different split counts: fp32 summation order only)
        x, dy, spec, like, _, _, wb = jobs[j]
        dw1, db1, packed = Fn.conv_wgrad(x, dy, spec, wb, like=like, two_pass=True)
        assert not packed
        close(dw, dw1.double().cpu(), tol=2e-5, what=f'multi vs single job {j}')
    # a set the library has no multi-job form for (channel counts differ): the caller falls back
    a = jobs[0]
    spec2 = Fn.ConvSpec(c, c * 2, (1, 3), (1, 1), (0, 1))
    assert Fn.conv_wgrad_multi([a, (a[0], act16(rnd(n, c * 2, h, w, seed=5), dtype), spec2,
                                    torch.empty(c * 2, c, 1, 3, device=DEV), None, None, True)]) is None
    # unequal pixel counts: the shorter job leaves its last split(s) empty -- all-zero
    # partial tiles for the reduction pass; same result as its own single-job launch
    if h > 3:
        hs = max(2, h // 2)
        xs, dys = act16(rnd(n, c, hs, w, seed=6), dtype), act16(rnd(n, c, hs, w, seed=7), dtype)
        out2 = Fn.conv_wgrad_multi([a[:4] + (None, None, True), (xs, dys, a[2], a[3], None, None, True)])
        if out2 is not None:
            torch.cuda.synchronize()
            for (xx, dd), (dw, db) in zip(((a[0], a[1]), (xs, dys)), out2):
                dw1, db1, packed = Fn.conv_wgrad(xx, dd, a[2], True, like=a[3], two_pass=True)
                assert not packed
                close(dw, dw1.double().cpu(), tol=2e-5, what='multi (unequal pixel counts) vs single')
                close(db, db1.double().cpu(), tol=2e-5, what='multi dbias (unequal pixel counts) vs single')


@pytest.mark.parametrize('dtype', DTYPES)
def test_stem16(dtype):
    """7x7/2 stem on the fp32 NCHW network input: packed to 16-bit NHWC4, conv_h kernel, BN stats"""
    Fn = _fn()
    for cin in (3, 1):
        n, h, w = 2, 32, 48
        x = rnd(n, cin, h, w, seed=1)
        wt = rnd(64, cin, 7, 7, seed=2, scale=0.1)
        ref = F.conv2d(q(x, dtype), q(wt, dtype), None, stride=2, padding=3)
        spec = Fn.StemSpec(cin, 64)
        xp = Fn.stem_pack_input(x.to(DEV), dtype)
        wp = Fn.stem_pack_weight(wt.to(DEV), dtype)
        y, stats = Fn.stem_fwd(xp, wp, spec, n, h, w, want_stats=True)
        torch.cuda.synchronize()
        close(y, ref, tol=TOL[dtype], what='stem16')
        close(stats[0].sum(0) / (ref.numel() / 64), ref.mean((0, 2, 3)), tol=2e-4, what='stem mean')
        if dtype == torch.bfloat16:
            wd = wt.double().requires_grad_(True)
            yy = F.conv2d(q(x, dtype), wd, None, stride=2, padding=3)
            dy = rnd(*yy.shape, seed=5)
            yy.backward(q(dy, dtype))
            dw = Fn.stem_wgrad(xp, act16(dy, dtype), spec, n, h, w, wt.to(DEV))
            close(dw, wd.grad, tol=2e-4, what='stem16 wgrad')


@pytest.mark.parametrize('c', [64, 2048])            # (2048: the last stage of the bottleneck ResNets)
@pytest.mark.parametrize('dtype', DTYPES)
def test_bn_act_and_backward16(dtype, c):
    Fn = _fn()
    n, h, w = 3, 9, 7
    x = rnd(n, c, h, w, seed=1)
    res = rnd(n, c, h, w, seed=2)
    sc, sh = rnd(c, seed=3), rnd(c, seed=4)
    drop = (rnd(n, c, seed=5) > 0).float() * 1.25
    xq = q(x, dtype).requires_grad_(True)
    rq = q(res, dtype).requires_grad_(True)
    ref = F.relu((xq * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
                 * drop.double()[:, :, None, None] + rq)
    y, bits = Fn.bn_act(act16(x, dtype), sc.to(DEV), sh.to(DEV), drop.to(DEV), act16(res, dtype),
                        Fn.ACT_RELU, want_mask=True)
    assert y.dtype == dtype
    close(y, ref, tol=TOL[dtype], what='bn_act16')
    # backward through batch-statistics BatchNorm + dropout + residual + ReLU (eval-style check of
    # the fused formula against autograd on the same rounded tensors)
    gamma = rnd(c, seed=6).abs() + 0.5
    xq2 = q(x, dtype).requires_grad_(True)
    mean = xq2.mean((0, 2, 3))
    var = xq2.var((0, 2, 3), unbiased=False)
    invstd = 1 / torch.sqrt(var + 1e-3)
    xhat = (xq2 - mean[None, :, None, None]) * invstd[None, :, None, None]
    beta = rnd(c, seed=7).double()
    rq2 = q(res, dtype).requires_grad_(True)
    out = F.relu((xhat * gamma.double()[None, :, None, None] + beta[None, :, None, None])
                 * drop.double()[:, :, None, None] + rq2)
    dy = rnd(n, c, h, w, seed=8)
    out.backward(q(dy, dtype))
    scale = (gamma.double() * invstd).float()
    shift = (beta - mean * gamma.double() * invstd).float()
    y2, bits2 = Fn.bn_act(act16(x, dtype), scale.to(DEV), shift.to(DEV), drop.to(DEV),
                          act16(res, dtype), Fn.ACT_RELU, want_mask=True)
    dx, dres, dg, db = Fn.bn_bwd(act16(dy, dtype), bits2, act16(x, dtype), gamma.to(DEV),
                                 mean.float().to(DEV), invstd.float().to(DEV), drop.to(DEV),
                                 Fn.ACT_RELU, True, want_dres=True)
    # the engine's ReLU mask is taken on ITS forward values: compare where both agree on the sign
    same = ((y2.float().cpu() > 0) == (out.detach() > 0))
    assert same.float().mean() > 0.995
    close(dres.float().cpu() * same, rq2.grad * same, tol=TOL[dtype], what='bn16 dres')
    close(dx.float().cpu() * same, xq2.grad * same, tol=3 * TOL[dtype], what='bn16 dx')
    assert dg.dtype == torch.float32 and db.dtype == torch.float32


@pytest.mark.parametrize('dtype', DTYPES)
def test_pool_se_upsample_ppm16(dtype):
    Fn = _fn()
    n, c, h, w = 2, 64, 10, 12
    x = rnd(n, c, h, w, seed=1)
    xq = q(x, dtype)
    # max pool 3x3/2 + backward
    xr = xq.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 3, 2, 1)
    y, idx = Fn.maxpool_fwd(act16(x, dtype))
    close(y, ref, tol=0, what='maxpool16')               # a max of stored values: exact
    dy = rnd(*ref.shape, seed=2)
    ref.backward(q(dy, dtype))
    close(Fn.maxpool_bwd(act16(dy, dtype), idx, (h, w)), xr.grad, tol=TOL[dtype], what='maxpool16 bwd')
    # SE pieces
    gap = Fn.channel_mean(act16(x, dtype))
    close(gap, xq.mean((2, 3)), tol=2e-4, what='channel mean16')
    sa, sb = torch.rand(n, c) + 0.1, torch.rand(n, c) + 0.1
    x2 = rnd(n, c, h, w, seed=3)
    out = Fn.se_scale_add(act16(x, dtype), sa.to(DEV), act16(x2, dtype), sb.to(DEV))
    close(out, xq * sa.double()[:, :, None, None] + q(x2, dtype) * sb.double()[:, :, None, None],
          tol=TOL[dtype], what='se_scale_add16')
    # both inputs of an SE-add fusion in two launches: bit-identical to the per-input launches
    cr = 16
    wa = [t.to(DEV) for t in (rnd(cr, c, seed=11, scale=0.2), rnd(cr, seed=12), rnd(c, cr, seed=13, scale=0.2), rnd(c, seed=14))]
    wb = [t.to(DEV) for t in (rnd(cr, c, seed=15, scale=0.2), rnd(cr, seed=16), rnd(c, cr, seed=17, scale=0.2), rnd(c, seed=18))]
    ga, gb, ha, hb, s_a, s_b = Fn.se_pair_fwd(act16(x, dtype), act16(x2, dtype), wa, wb)
    for xin, wts, g_, h_, s_ in ((x, wa, ga, ha, s_a), (x2, wb, gb, hb, s_b)):
        g1 = Fn.channel_mean(act16(xin, dtype))
        h1, s1 = Fn.se_mlp_fwd(g1, *wts)
        assert torch.equal(g_, g1) and torch.equal(h_, h1) and torch.equal(s_, s1)
    ds = Fn.se_scale_bwd_reduce(act16(x2, dtype), act16(x, dtype))
    close(ds, (q(x2, dtype) * xq).sum((2, 3)), tol=2e-4, what='se reduce16')
    dgap = rnd(n, c, seed=4)
    dxs = Fn.se_scale_bwd_apply(act16(x2, dtype), sa.to(DEV), dgap.to(DEV))
    close(dxs, q(x2, dtype) * sa.double()[:, :, None, None] + dgap.double()[:, :, None, None] / (h * w),
          tol=TOL[dtype], what='se apply16')
    # learned x2 up-sampling: 16-bit -> 16-bit with skip, and 16-bit -> fp32 (model boundary)
    wdw = rnd(c, 1, 3, 3, seed=5, scale=0.3)
    bias = rnd(c, seed=6)
    skip = rnd(n, c, 2 * h, 2 * w, seed=7)
    xr = xq.clone().requires_grad_(True)
    wr = wdw.double().requires_grad_(True)
    up = F.conv2d(F.interpolate(xr, scale_factor=2, mode='nearest'), wr, bias.double(), padding=1,
                  groups=c)
    y16 = Fn.up2x_dw_fwd(act16(x, dtype), wdw.to(DEV), bias.to(DEV), act16(skip, dtype))
    close(y16, up + q(skip, dtype), tol=TOL[dtype], what='up2x16')
    y32 = Fn.up2x_dw_fwd(act16(x, dtype), wdw.to(DEV), bias.to(DEV), None, out_f32=True)
    assert y32.dtype == torch.float32
    close(y32, up, tol=2e-4, what='up2x16 -> fp32')
    dyu = rnd(*up.shape, seed=8)
    up.backward(dyu.double())                              # fp32 cotangent (model boundary)
    from util import to_act
    dx, dw, db = Fn.up2x_dw_bwd(to_act(dyu), act16(x, dtype), wdw.to(DEV).contiguous())
    assert dx.dtype == dtype
    close(dx, xr.grad, tol=TOL[dtype], what='up2x16 dx (fp32 dy)')
    close(dw.view(c, 1, 3, 3), wr.grad, tol=2e-4, what='up2x16 dw')
    # pyramid pooling pieces
    close(Fn.adaptive_avgpool_fwd(act16(x, dtype), 5), F.adaptive_avg_pool2d(xq, 5), tol=TOL[dtype],
          what='avgpool16')
    small = rnd(n, c, 5, 5, seed=9)
    buf = Fn.act_empty(n, c, h, w, DEV, dtype=dtype)
    Fn.bilinear_fwd(act16(small, dtype), buf)
    close(buf, F.interpolate(q(small, dtype), size=(h, w), mode='bilinear', align_corners=False),
          tol=TOL[dtype], what='bilinear16')
    sr = q(small, dtype).requires_grad_(True)
    F.interpolate(sr, size=(h, w), mode='bilinear', align_corners=False).backward(q(x, dtype))
    close(Fn.bilinear_bwd(act16(x, dtype), (5, 5)), sr.grad, tol=TOL[dtype], what='bilinear16 bwd')


@pytest.mark.parametrize('dtype', DTYPES)
def test_head_act_and_casts16(dtype):
    from emsanet_amd import ops
    Fn = _fn()
    x = rnd(2, 8, 6, 7, seed=1)
    xq = q(x, dtype).requires_grad_(True)
    y = torch.cat([torch.sigmoid(xq[:, :1]), torch.tanh(xq[:, 1:3]), xq[:, 3:]], 1)
    dy = rnd(*y.shape, seed=2)
    y.backward(dy.double())
    xg = act16(x, dtype).requires_grad_(True)
    c, o, r = ops.HeadActFunction.apply(xg, 1, 2, (1, 2, 2))
    assert c.dtype == torch.float32                       # model outputs are fp32
    close(torch.cat([c, o, r], 1), y[:, :5], tol=2e-4, what='head act16')
    torch.autograd.backward([c, o, r], [dy[:, :1].contiguous().to(DEV), dy[:, 1:3].contiguous().to(DEV),
                                        dy[:, 3:5].contiguous().to(DEV)])
    ref = xq.grad.clone()
    ref[:, 5:] = 0
    assert xg.grad.dtype == dtype
    close(xg.grad, ref, tol=TOL[dtype], what='head act16 bwd')
    # casts (side outputs / scene logits leave the engine as fp32)
    t = act16(x, dtype).requires_grad_(True)
    f = ops.to_float(t)
    assert f.dtype == torch.float32 and torch.equal(f.cpu().double(), q(x, dtype))
    f.backward(torch.ones_like(f))
    assert t.grad.dtype == dtype and float(t.grad.float().min()) == 1.0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('n_norm', [0, 2])
def test_head_act_bwd_gather_is_copy_plus_bwd(dtype, n_norm):
    """emsa_head_act_bwd_gather_t (the three task gradients read inside the backward kernel) == the
    padded copy + emsa_head_act_bwd_t it replaces, bit for bit: channels-last and NCHW-contiguous
    gradients, a missing task gradient, the L2-normalised orientation pair, odd sizes"""
    Fn = _fn()
    n, h, w = 2, 7, 9
    x = rnd(n, 8, h, w, seed=1)
    xg = act16(x, dtype) if dtype != torch.float32 else x.to(DEV).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    y = Fn.head_act_fwd(xg, 1, 2, n_norm)
    sizes = (1, 2, 2)
    gs = [rnd(n, s, h, w, seed=5 + k).to(DEV) for k, s in enumerate(sizes)]
    gs[1] = gs[1].contiguous(memory_format=torch.channels_last)
    for drop in (None, 2):
        grads = [g if k != drop else None for k, g in enumerate(gs)]
        dy = Fn.act_zeros(n, 8, h, w, DEV)
        o = 0
        for s, g in zip(sizes, grads):
            if g is not None:
                Fn.copy_channels(Fn.as_act(g), dy[:, o:o + s])
            o += s
        ref = Fn.head_act_bwd(dy, y, 1, 2, n_norm, x=xg if n_norm else None, dtype=dtype)
        got = Fn.head_act_bwd_gather(grads, sizes, y, 1, 2, n_norm, x=xg if n_norm else None, dtype=dtype)
        torch.cuda.synchronize()
        assert got is not None and got.dtype == dtype and torch.equal(got, ref), (dtype, n_norm, drop)


_BN_FAST_CODE = '''
import sys, torch
sys.path.insert(0, "tests")
from util import DEV, rnd
from emsanet_amd import functional as Fn

def run(dtype, n, c, h, w):
    def act(seed):
        g = torch.Generator().manual_seed(seed)
        t = torch.randn(n, h, w, c, generator=g).to(dtype).to(DEV)
        return t.permute(0, 3, 1, 2)
    x, res, dy = act(1), act(2), act(3)
    sc, sh = (rnd(c, seed=4).abs() + 0.5).to(DEV), rnd(c, seed=5).to(DEV)
    mean, inv = rnd(c, seed=6).to(DEV), (rnd(c, seed=7).abs() + 0.5).to(DEV)
    drop = ((rnd(n, c, seed=8) > -0.5).float() * 1.25).to(DEV)
    out = []
    for d, r in ((None, None), (drop, res), (None, res), (drop, None)):
        y, bits = Fn.bn_act(x, sc, sh, d, r, Fn.ACT_RELU, want_mask=True)
        out += [y, bits[:(y.numel() // 64)]]
        for mask in (bits, y):
            for train in (True, False):
                out += [t for t in Fn.bn_bwd(dy, mask, x, sc, mean, inv, d, Fn.ACT_RELU, train, r is not None)
                        if t is not None]
    y = Fn.bn_act(x, sc, sh, None, None, Fn.ACT_NONE)
    out += [y] + [t for t in Fn.bn_bwd(dy, None, x, sc, mean, inv, None, Fn.ACT_NONE, True, False)
                  if t is not None]
    torch.cuda.synchronize()
    return [t.cpu() for t in out]

res = {}
for name, dtype in (("bf16", torch.bfloat16), ("f32", torch.float32)):
    # several grid strides per thread (pair loop + tail), a ragged end, c = 512 (64 / 128 vectors)
    for shp in ((8, 64, 120, 160), (3, 128, 37, 41), (2, 512, 15, 20), (5, 64, 9, 7)):
        res[name + str(shp)] = run(dtype, *shp)
torch.save(res, sys.argv[1])
print("BN_DUMP_OK")
'''


def test_bn_fast_kernels_are_bit_identical_to_the_general_ones(tmp_path):
    """round 5: the BatchNorm passes run on `*_fast_kernel` forms (fixed channel vector per thread,
    all loads of an iteration issued first, csrc/pointwise.hip); EMSA_BN_FAST=0 selects the general
    loops they replace -- same arithmetic in the same order, so every output must be bit-identical"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for flag in ('1', '0'):
        path = str(tmp_path / f'bn_{flag}.pt')
        env = dict(os.environ, EMSA_BN_FAST=flag)
        r = subprocess.run([sys.executable, '-c', _BN_FAST_CODE, path], cwd=root, env=env,
                           capture_output=True, text=True, timeout=600)
        assert 'BN_DUMP_OK' in r.stdout, r.stderr[-3000:]
        outs[flag] = torch.load(path)
    assert outs['1'].keys() == outs['0'].keys()
    for k in outs['1']:
        a, b = outs['1'][k], outs['0'][k]
        assert len(a) == len(b) and len(a) > 20
        for i, (ta, tb) in enumerate(zip(a, b)):
            assert ta.dtype == tb.dtype and ta.shape == tb.shape
            assert torch.equal(ta, tb), (k, i, (ta.float() - tb.float()).abs().max().item())


_IDX_CODE = '''
import sys, torch
sys.path.insert(0, "tests")
from util import DEV, rnd
from emsanet_amd import functional as Fn

def run(dtype, n, c, h, w):
    def act(seed, hh=h, ww=w):
        g = torch.Generator().manual_seed(seed)
        return torch.randn(n, hh, ww, c, generator=g).to(dtype).to(DEV).permute(0, 3, 1, 2)
    x, b = act(1), act(2)
    out = []
    y, idx = Fn.maxpool_fwd(x)
    out += [y, idx, Fn.maxpool_bwd(act(3, (h + 1) // 2, (w + 1) // 2), idx, (h, w))]
    sa, sb = (rnd(n, c, seed=4).abs() + 0.1).to(DEV), (rnd(n, c, seed=5).abs() + 0.1).to(DEV)
    out += [Fn.se_scale_add(x, sa), Fn.se_scale_add(x, sa, b, sb)]
    out += [Fn.se_scale_bwd_apply(x, sa, sb), Fn.se_scale_bwd_apply(x, sa, sb, b)]
    wdw, bias = rnd(c, 1, 3, 3, seed=6).to(DEV), rnd(c, seed=7).to(DEV)
    skip = act(8, 2 * h, 2 * w)
    out += [Fn.up2x_dw_fwd(x, wdw, bias), Fn.up2x_dw_fwd(x, wdw, bias, skip)]
    if dtype != torch.float32:
        out += [Fn.up2x_dw_fwd(x, wdw, bias, None, out_f32=True)]
    dx, dw, db = Fn.up2x_dw_bwd(skip, x, wdw)
    out += [dx]                                 # (dw / db: fp32 atomics, compared by tolerance elsewhere)
    torch.cuda.synchronize()
    return [t.cpu() for t in out]

res = {}
for name, dtype in (("bf16", torch.bfloat16), ("f32", torch.float32)):
    for shp in ((2, 64, 23, 30), (3, 40, 9, 7), (1, 128, 16, 20)):
        if dtype != torch.float32 and shp[1] % 8:
            continue
        res[name + str(shp)] = run(dtype, *shp)
torch.save(res, sys.argv[1])
print("IDX_DUMP_OK")
'''


def test_index_width_variants_are_bit_identical(tmp_path):
    """round 5: max-pool, SE scale and up-sampling kernels run on 32-bit element indices whenever the
    tensors have < 2^31 elements (every shape of every test); EMSA_IDX32=0 forces the 64-bit
    instantiations that larger tensors take -- they must produce the same bits"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for flag in ('1', '0'):
        path = str(tmp_path / f'idx_{flag}.pt')
        r = subprocess.run([sys.executable, '-c', _IDX_CODE, path], cwd=root,
                           env=dict(os.environ, EMSA_IDX32=flag), capture_output=True, text=True,
                           timeout=600)
        assert 'IDX_DUMP_OK' in r.stdout, r.stderr[-3000:]
        outs[flag] = torch.load(path)
    assert outs['1'].keys() == outs['0'].keys() and len(outs['1']) >= 5
    for k in outs['1']:
        for i, (ta, tb) in enumerate(zip(outs['1'][k], outs['0'][k])):
            assert ta.dtype == tb.dtype and ta.shape == tb.shape
            assert torch.equal(ta, tb), (k, i)
