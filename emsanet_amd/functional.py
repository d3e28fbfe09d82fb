# -*- coding: utf-8 -*-
"""
Raw (no autograd) host wrappers around the C-ABI of libemsanet_hip.so.

Tensor convention: an activation is a torch tensor of logical shape (N, C, H, W) whose memory is
NHWC (torch "channels_last"), possibly a channel slice of a wider NHWC buffer (pixel stride
`ld` >= C).  torch is used here for device memory and the current stream only; every byte of
arithmetic happens in the HIP kernels.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import EmsaConvGeom, check

ACT_NONE, ACT_RELU = 0, 1


def _stream():
    return torch.cuda.current_stream().cuda_stream


_POISON = bool(os.environ.get('EMSA_POISON'))


def _empty(shape, device, dtype=torch.float32):
    """scratch/output allocation; EMSA_POISON=1 fills it with NaN (debug: finds reads of
    memory a kernel was supposed to write)"""
    t = torch.empty(shape, device=device, dtype=dtype)
    if _POISON:
        t.fill_(float('nan') if dtype.is_floating_point else 111)
    return t


# storage types of the activation tensors (EMSA_DT_* of include/emsanet_hip.h): fp32 is the
# reference's arithmetic (BASELINE configs[1]), bf16 the mixed-precision training path (configs[2]),
# bf16 / fp16 the 16-bit inference path (configs[4])
DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def dt(t):
    """EMSA_DT_* code of an activation tensor"""
    try:
        return DT[t.dtype]
    except KeyError:
        raise _lib.EmsaError(f"activation dtype {t.dtype} (the engine stores fp32, bf16 or fp16)")


def call_t(name, code, *args):
    """emsa_<name>(...) for fp32 activations, emsa_<name>_t(dtype, ...) for the 16-bit ones"""
    L = _lib.lib()
    if code == 0:
        return getattr(L, name)(*args)
    return getattr(L, name + '_t')(code, *args)


def act_empty(n, c, h, w, device, ld=None, dtype=torch.float32):
    """(N,C,H,W) view over fresh NHWC memory (optionally a slice of an ld-wide buffer)."""
    if ld is None or ld == c:
        return _empty((n, h, w, c), device, dtype).permute(0, 3, 1, 2)
    return _empty((n, h, w, ld), device, dtype)[..., :c].permute(0, 3, 1, 2)


def act_zeros(n, c, h, w, device, dtype=torch.float32):
    return torch.zeros((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


def ld_of(t):
    """pixel stride of an NHWC activation (validates the layout)."""
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    ld = sw if w > 1 else (sh // w if h > 1 else (sn // (h * w) if n > 1 else c))
    if c > 1 and sc != 1:
        raise _lib.EmsaError(f"activation is not NHWC: shape {tuple(t.shape)} stride {t.stride()}")
    if (w > 1 and sw != ld) or (h > 1 and sh != w * ld) or (n > 1 and sn != h * w * ld) or ld < c:
        raise _lib.EmsaError(f"activation is not NHWC: shape {tuple(t.shape)} stride {t.stride()}")
    return ld


def to_nhwc(t):
    """Boundary helper: accept any layout of a logical NCHW tensor, return dense NHWC memory.
    A foreign layout (the reference's contiguous NCHW decoder inputs and loss cotangents, expanded
    scalars, sliced views) costs ONE pass of `emsa_to_nhwc_t` -- never a torch
    permute().contiguous()."""
    if t.dtype not in DT:
        t = t.float()
    n, c, h, w = t.shape
    try:
        if ld_of(t) == c:
            return t
    except _lib.EmsaError:
        pass
    if not t.is_cuda:
        # host-side dry runs (tests with a stand-in library): layout bookkeeping only
        return t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    out = act_empty(n, c, h, w, t.device, dtype=t.dtype)
    sn, sc, sh, sw = t.stride()
    check(_lib.lib().emsa_to_nhwc_t(DT[t.dtype], _p(t), _p(out), n, c, h, w, sn, sc, sh, sw,
                                    _stream()), 'emsa_to_nhwc_t')
    return out


def as_act(t, dense=False):
    """t itself when it already is an NHWC activation (dense if requested), else a dense copy."""
    if t.dtype in DT:
        try:
            ld = ld_of(t)
            if not dense or ld == t.shape[1]:
                return t
        except _lib.EmsaError:
            pass
    return to_nhwc(t)


# bench.py: a one-element list while it counts the ALGORITHMIC bytes of one eager step -- every tensor
# operand of every launch once (sum over fused kernels of input + output + weight
# bytes); None otherwise
BYTES = None


def _p(t):
    if t is None:
        return None
    if BYTES is not None:
        BYTES[0] += t.numel() * t.element_size()
    return t.data_ptr()


# bench.py sets this while it samples kernel times: convs that run on zero-padded channel counts
# (stem, merged / padded heads) then declare their REAL direct-convolution FLOPs per launch
PROF_REAL_FLOPS = False


def prof_flops(flops):
    if PROF_REAL_FLOPS:
        _lib.lib().emsa_prof_next_flops(float(flops))


class ConvSpec:
    """Geometry of one nn.Conv2d (OIHW parameter of shape [cout, cin, kh, kw])."""

    __slots__ = ('cin', 'cout', 'kh', 'kw', 'sh', 'sw', 'ph', 'pw', '_geoms')

    def __init__(self, cin, cout, kernel, stride=1, padding=0):
        self.cin, self.cout = cin, cout
        self.kh, self.kw = (kernel, kernel) if isinstance(kernel, int) else kernel
        self.sh, self.sw = (stride, stride) if isinstance(stride, int) else stride
        self.ph, self.pw = (padding, padding) if isinstance(padding, int) else padding
        self._geoms = {}

    def out_hw(self, h, w):
        return ((h + 2 * self.ph - self.kh) // self.sh + 1,
                (w + 2 * self.pw - self.kw) // self.sw + 1)

    def geom_fwd(self, n, h, w, ld_in, ld_out):
        key = ('f', n, h, w, ld_in, ld_out)
        g = self._geoms.get(key)
        if g is None:
            oh, ow = self.out_hw(h, w)
            g = EmsaConvGeom(n, h, w, oh, ow, self.cin, self.cout, self.kh, self.kw,
                             self.sh, -self.ph, 1, 1, self.sw, -self.pw, 1, 1,
                             h * w * ld_in, w * ld_in, ld_in, ld_out)
            self._geoms[key] = g
        return g

    def geom_dgrad(self, n, h, w, ld_dy, ld_dx):
        """h, w = input (dx) size; gathers from dy (out_hw) with the transposed index map."""
        key = ('d', n, h, w, ld_dy, ld_dx)
        g = self._geoms.get(key)
        if g is None:
            oh, ow = self.out_hw(h, w)
            g = EmsaConvGeom(n, oh, ow, h, w, self.cout, self.cin, self.kh, self.kw,
                             1, self.ph, -1, self.sh, 1, self.pw, -1, self.sw,
                             oh * ow * ld_dy, ow * ld_dy, ld_dy, ld_dx)
            self._geoms[key] = g
        return g


def pad4(c):
    return (c + 3) // 4 * 4


def pad8(c):
    """channel padding of the narrow heads: multiples of 8 so that the 16-bit kernels' 16-byte
    accesses (8 channels) apply to them too"""
    return (c + 7) // 8 * 8


def pack_weight_t(w, dtype, fwd=True, dgrad=False, cout_total=None, cout_off=0, cin_total=None,
                  cin_off=0, out_fwd=None, out_dgrad=None):
    """fp32 OIHW parameter -> 16-bit packed operands of emsa_conv_igemm_t: forward
    [tap][cout_total][cin_total] and / or data gradient [tap][cin_total][cout_total]"""
    cout, cin, kh, kw = w.shape if w.dim() == 4 else (w.shape[0], w.shape[1], 1, 1)
    cout_total = cout_total or cout
    cin_total = cin_total or cin
    n = kh * kw * cout_total * cin_total
    padded = cout_total != cout or cin_total != cin
    mk = (lambda: torch.zeros(n, device=w.device, dtype=dtype)) if padded \
        else (lambda: _empty((n,), w.device, dtype))
    if fwd and out_fwd is None:
        out_fwd = mk()
    if dgrad and out_dgrad is None:
        out_dgrad = mk()
    check(_lib.lib().emsa_pack_weight_t(DT[dtype], _p(w.contiguous()), _p(out_fwd) if fwd else None,
                                        _p(out_dgrad) if dgrad else None, cout, cin, kh, kw,
                                        cout_total, cout_off, cin_total, cin_off, _stream()),
          'emsa_pack_weight_t')
    return out_fwd if fwd else None, out_dgrad if dgrad else None


# The register-stationary streaming kernel (csrc/conv_rs.hip) for the stride-1 3-tap 1-D convs of the
# NBt1D blocks in 16-bit storage (/root/reference/emsanet/model.py:47-58): C_in = C_out in
# {64, 128, 256, 512}.  EMSA_CONV_RS=0 keeps every conv on the generic implicit GEMM.
CONV_RS = os.environ.get('EMSA_CONV_RS', '1') != '0'     # tests: set False / True


def rs_eligible(spec):
    return ((spec.kh, spec.kw, spec.ph, spec.pw) in ((3, 1, 1, 0), (1, 3, 0, 1)) and spec.sh == 1
            and spec.sw == 1 and spec.cin == spec.cout and spec.cin in (64, 128, 256, 512))


def _resolve(w):
    """packed weights, or a callable that packs them on demand (ops.ConvRT hands the [tap][n][k] form of
    a conv_rs-capable conv over lazily: where conv_rs takes the geometry nobody reads it)"""
    return w() if callable(w) else w


def rs_supported(code, g):
    """the kernel takes this geometry (cached on the geometry object: a plan per shape)"""
    if not CONV_RS or code == 0:
        return False
    r = getattr(g, '_rs', None)
    if r is None or r[0] != code or r[3] != RS_EPOCH:
        ok = _lib.lib().emsa_conv1d_rs_supported(code, g) == 1
        r = (code, ok, _lib.lib().emsa_conv1d_rs_stats_rows(code, g) if ok else 0, RS_EPOCH)
        g._rs = r
    return r[1]


# The persistent grid of conv_rs is sized to the CUs of the device; a CU budget below that leaves
# room for kernels that must be co-resident for the whole step -- RCCL's all-reduce workgroups when
# several ranks train (a collective's channels must not queue behind a
# 2-workgroups-per-CU persistent grid whose static tile partition waits for its last workgroup).
# EMSA_RS_CUS=n fixes it by hand; parallel.GradientBuckets sets it when the world has > 1 rank.  The
# plans (statistics rows per launch) are cached per geometry: RS_EPOCH invalidates them.
RS_EPOCH = 0


def set_rs_cu_budget(cus):
    """cus <= 0: all CUs.  -> the number of CUs conv_rs now plans with"""
    global RS_EPOCH
    got = _lib.lib().emsa_conv_rs_set_cu_budget(int(cus))
    RS_EPOCH += 1
    return got


def _splitk_ws_bytes(code, g):
    """workspace bytes of the tap-split form of this launch (0: plain launch), cached per geometry"""
    r = getattr(g, '_sk', None)
    if r is None or r[0] != code:
        r = (code, int(_lib.lib().emsa_conv_igemm_splitk_ws_bytes_t(code, g)))
        g._sk = r
    return r[1]


def pack_weight_frag_t(w, dtype, fwd=True, dgrad=False, out_fwd=None, out_dgrad=None):
    """fp32 OIHW [cout][cin][3 taps] -> fragment-ordered 16-bit operands of emsa_conv1d_rs_t"""
    cout, cin = w.shape[0], w.shape[1]
    n = 3 * cout * cin
    if fwd and out_fwd is None:
        out_fwd = _empty((n,), w.device, dtype)
    if dgrad and out_dgrad is None:
        out_dgrad = _empty((n,), w.device, dtype)
    check(_lib.lib().emsa_pack_weight_frag_t(DT[dtype], _p(w.contiguous()),
                                             _p(out_fwd) if fwd else None,
                                             _p(out_dgrad) if dgrad else None, cout, cin,
                                             _stream()), 'emsa_pack_weight_frag_t')
    return out_fwd if fwd else None, out_dgrad if dgrad else None


# ---------------------------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------------------------
def pack_weight(w, mode, cout_total=None, cout_off=0, cin_total=None, cin_off=0, out=None):
    """mode 'fwd' -> [tap][cout_total][cin_total]; 'dgrad' -> [tap][cin_total][cout_total]."""
    cout, cin, kh, kw = w.shape if w.dim() == 4 else (w.shape[0], w.shape[1], 1, 1)
    cout_total = cout_total or cout
    cin_total = cin_total or cin
    if out is None:
        if cout_total != cout or cin_total != cin:
            out = torch.zeros(kh * kw * cout_total * cin_total, device=w.device, dtype=torch.float32)
        else:
            out = _empty(kh * kw * cout_total * cin_total, w.device)
    fn = _lib.lib().emsa_pack_weight_fwd if mode == 'fwd' else _lib.lib().emsa_pack_weight_dgrad
    check(fn(_p(w), _p(out), cout, cin, kh, kw, cout_total, cout_off, cin_total, cin_off,
             _stream()), 'emsa_pack_weight_' + mode)
    return out


def pack_weight_pair(w):
    """both packed layouts ([tap][cout][cin], [tap][cin][cout]) in one launch"""
    cout, cin, kh, kw = w.shape
    buf = _empty((2, kh * kw * cout * cin), w.device)
    check(_lib.lib().emsa_pack_weight_pair(_p(w), _p(buf[0]), _p(buf[1]), cout, cin, kh, kw,
                                           _stream()), 'emsa_pack_weight_pair')
    return buf[0], buf[1]


def unpack_wgrad(dwp, like, cout_total=None, cout_off=0, cin_total=None, cin_off=0, out=None):
    """`out`: contiguous destination with the parameter's shape (e.g. its flat-bucket view)"""
    cout, cin, kh, kw = like.shape if like.dim() == 4 else (like.shape[0], like.shape[1], 1, 1)
    dw = out if out is not None else _empty(tuple(like.shape), like.device)
    check(_lib.lib().emsa_unpack_wgrad(_p(dwp), _p(dw), cout, cin, kh, kw, cout_total or cout,
                                       cout_off, cin_total or cin, cin_off, _stream()),
          'emsa_unpack_wgrad')
    return dw


# ---------------------------------------------------------------------------------------------
# convolution
# ---------------------------------------------------------------------------------------------
def wino_eligible(spec):
    """stride-1 "same" convolution with a 3-tap row (NBt1D 3x1 / 1x3, decoder / head 3x3) and
    channel counts the Winograd kernel accepts"""
    shape = (spec.kh, spec.kw, spec.ph, spec.pw) in ((3, 1, 1, 0), (1, 3, 0, 1), (3, 3, 1, 1))
    return shape and spec.sh == 1 and spec.sw == 1 and spec.cin % 4 == 0 and spec.cout % 4 == 0


def wino_rows(spec):
    return 3 if (spec.kh, spec.kw) == (3, 3) else 1


def pack_wino(w, fwd=True, dgrad=False):
    """OIHW [cout][cin][3x1|1x3] -> Winograd F(2,3) weights U [4][n][k]: forward (n,k = cout,cin)
    and/or data gradient (n,k = cin,cout, flipped taps), one launch.  -> (u or None, ud or None)"""
    cout, cin, kh, kw = w.shape
    rows = 3 if (kh, kw) == (3, 3) else 1
    buf = _empty((int(fwd) + int(dgrad), 4 * rows * cout * cin), w.device)
    u = buf[0] if fwd else None
    ud = buf[-1] if dgrad else None
    check(_lib.lib().emsa_pack_wino(_p(w.contiguous()), _p(u), _p(ud), cout, cin, rows,
                                    _stream()), 'emsa_pack_wino')
    return u, ud


def pack_wino_packed(wp, n_ch, k_ch, rows, flip):
    """Winograd weights from a PACKED [tap][n_ch][k_ch] weight (merged / channel-padded convs);
    flip=True for a data-gradient pack [tap][cin][cout]"""
    u = _empty((4 * rows * n_ch * k_ch,), wp.device)
    check(_lib.lib().emsa_pack_wino_packed(_p(wp), _p(u), n_ch, k_ch, rows, 1 if flip else 0,
                                           _stream()), 'emsa_pack_wino_packed')
    return u


def conv_fwd(x, wp, spec, bias=None, want_stats=False, scale=None, shift=None, residual=None,
             act=ACT_NONE, out=None, wino_u=None, want_relu_bits=False, in_affine=None, wfrag=None):
    """wp = packed [tap][cout][cin] weights in the dtype of `x` (MFMA implicit GEMM) -- or, fp32
    only, wino_u = Winograd weights of an eligible conv (emsa_conv1d_wino).  want_relu_bits
    (Winograd kernel with act = ReLU): additionally returns (out > 0) as a bit mask for
    `conv_dgrad(mask_bits=...)`, else None.  in_affine = (scale, shift) per INPUT channel: the
    conv runs on relu(x * scale + shift) formed in its loader (emsa_conv1d_wino_inbn: the
    BatchNorm + ReLU in front of a 1-D Winograd conv without a pass of its own).  wfrag = the
    fragment-ordered 16-bit weights of a 3-tap 1-D conv: where emsa_conv1d_rs_t takes the geometry
    it runs instead of the implicit GEMM (same epilogue, its own statistics rows)."""
    n, c, h, w = x.shape
    oh, ow = spec.out_hw(h, w)
    code = dt(x)
    if out is None:
        out = act_empty(n, spec.cout, oh, ow, x.device, dtype=x.dtype)
    g = spec.geom_fwd(n, h, w, ld_of(x), ld_of(out))
    L = _lib.lib()
    stats = None
    if code != 0:
        wino_u = None                       # the 16-bit path is conv_rs.hip / conv_h.hip
    use_rs = wfrag is not None and wfrag.dtype == x.dtype and rs_supported(code, g) and \
        not (want_stats and residual is not None)
    if want_stats:
        if wino_u is not None:
            rows = L.emsa_conv1d_wino_stats_rows(g)
        elif use_rs:
            rows = g._rs[2]
        else:
            rows = call_t('emsa_conv_stats_rows', code, g)
        if rows <= 0:
            check(rows or -1, 'emsa_conv_stats_rows')
        stats = _empty((3, rows, spec.cout), x.device)
    lr = ld_of(residual) if residual is not None else 0
    bits = None
    if in_affine is not None and code != 0:
        # 16-bit twin of the fold: conv_rs.hip takes the BatchNorm + ReLU of its input into the loader
        # (3x1 convs; emsa_conv1d_rs_inbn_t).  The caller asked `bn1_fold16_ok` first.
        if not use_rs or scale is not None or residual is not None:
            raise _lib.EmsaError("in_affine (16-bit): only the conv_rs 3x1 forward folds its input's "
                                 "BatchNorm (no output affine / residual)")
        check(L.emsa_conv1d_rs_inbn_t(code, g, _p(x), _p(wfrag), _p(out), _p(bias), _p(stats),
                                      _p(in_affine[0]), _p(in_affine[1]), act, _stream()),
              'emsa_conv1d_rs_inbn_t')
        res = (out, stats) if want_stats else out
        return (res, None) if want_relu_bits else res
    if in_affine is not None and (wino_u is None or scale is not None or residual is not None):
        raise _lib.EmsaError("in_affine: only the fp32 1-D Winograd forward folds its input's "
                             "BatchNorm (no output affine / residual)")
    if wino_u is not None:
        if want_relu_bits and act == ACT_RELU:
            bits = torch.empty(L.emsa_conv_relu_bits_words(n * oh * ow, spec.cout),
                               device=x.device, dtype=torch.int64)
        if in_affine is not None:
            check(L.emsa_conv1d_wino_inbn(g, _p(x), _p(wino_u), _p(out), _p(bias), _p(stats),
                                          _p(in_affine[0]), _p(in_affine[1]), act, _p(bits),
                                          _stream()), 'emsa_conv1d_wino_inbn')
            res = (out, stats) if want_stats else out
            return (res, bits) if want_relu_bits else res
        check(L.emsa_conv1d_wino(g, _p(x), _p(wino_u), _p(out), _p(bias), _p(stats), _p(scale),
                                 _p(shift), _p(residual), lr, None, 0, act, None, _p(bits),
                                 _stream()), 'emsa_conv1d_wino')
    elif use_rs:
        check(L.emsa_conv1d_rs_t(code, g, _p(x), _p(wfrag), _p(out), _p(bias), _p(stats), _p(scale),
                                 _p(shift), _p(residual), lr, None, 0, act, _stream()),
              'emsa_conv1d_rs_t')
    elif code != 0 and not want_stats and _splitk_ws_bytes(code, g) > 0:
        # few output tiles, long K (the decoders' 3x3 convs at batch 1): tap-split + finish pass
        wp = _resolve(wp)
        if wp is None or wp.dtype != x.dtype:
            raise _lib.EmsaError("conv weights are not packed in the activations' dtype")
        ws = _empty((_splitk_ws_bytes(code, g) // 4,), x.device)
        check(L.emsa_conv_igemm_splitk_t(code, g, _p(x), _p(wp), _p(out), _p(bias), _p(scale),
                                         _p(shift), _p(residual), lr, act, _p(ws), _stream()),
              'emsa_conv_igemm_splitk_t')
    else:
        wp = _resolve(wp)
        if wp is None or wp.dtype != x.dtype:
            raise _lib.EmsaError(f"conv weights packed as {None if wp is None else wp.dtype} for "
                                 f"{x.dtype} activations")
        check(call_t('emsa_conv_igemm', code, g, _p(x), _p(wp), _p(out), _p(bias), _p(stats),
                     _p(scale), _p(shift), _p(residual), lr, None, 0, act, _stream()),
              'emsa_conv_igemm')
    res = (out, stats) if want_stats else out
    return (res, bits) if want_relu_bits else res


# Fused NBt1D half-block (csrc/conv_hb.hip): conv3x1 + ReLU -> conv1x3 + folded BatchNorm (+ residual)
# + ReLU as ONE launch, for the small-batch 16-bit eval fast path where a launch is its fixed cost.
# OPT-IN (EMSA_HALF_BLOCK=1): bit-identical to the two launches it replaces and 19 graph nodes fewer at
# batch 1 (155 -> 136), but not faster -- one fused launch takes 11-13 us at C = 64 and 14-16 us at
# C = 128 where the two twin conv_rs launches take 2 x 5.5-6.5 (profiles/r05_e_*: 1.16 vs 1.14-1.15 ms
# per forward).  With it on, maps above HALF_BLOCK_MAX_PIXELS pixels (bs * h * w at the
# block's resolution) still take the persistent conv_rs kernel.
_HALF_BLOCK_ENV = os.environ.get('EMSA_HALF_BLOCK')
HALF_BLOCK = None                    # tests: True / False overrides
HALF_BLOCK_MAX_PIXELS = 3 * 120 * 160


def half_block_ok(x, c):
    """this map (N, c, H, W) takes the fused half-block kernel"""
    if x.dtype == torch.float32 or c not in (64, 128) or ld_of(x) != c:
        return False
    on = HALF_BLOCK if HALF_BLOCK is not None else (_HALF_BLOCK_ENV == '1')
    if not on:
        return False
    n, _, h, w = x.shape
    if n * h * w > HALF_BLOCK_MAX_PIXELS:
        return False
    return _lib.lib().emsa_nbt_half_block_supported(dt(x), c, w) == 1


def nbt_half_block(xs, wfa, bias_a, wfb, bias_b, scales, shifts, residuals, act=ACT_RELU):
    """xs: 1 or 2 dense maps of the same shape (the twin modules); every other argument a sequence of
    the same length (entries of bias / scale+shift / residual may be None) -> tuple of outputs"""
    k = len(xs)
    x0 = xs[0]
    n, c, h, w = x0.shape
    code = dt(x0)
    for x in xs:
        if x.shape != x0.shape or x.dtype != x0.dtype or ld_of(x) != c:
            raise _lib.EmsaError("nbt_half_block: the tensor sets differ in shape / dtype / layout")
    rs = [r for r in residuals if r is not None]
    if rs and (len(rs) != k or any(ld_of(r) != ld_of(rs[0]) or r.dtype != x0.dtype for r in rs)):
        raise _lib.EmsaError("nbt_half_block: residual operands differ between the tensor sets")
    outs = [act_empty(n, c, h, w, x0.device, dtype=x0.dtype) for _ in range(k)]

    def arr(ts):
        a = (ctypes.c_void_p * k)()
        for j, t in enumerate(ts):
            a[j] = _p(t)
        return a
    for wf in list(wfa) + list(wfb):
        if wf is None or wf.dtype != x0.dtype:
            raise _lib.EmsaError("nbt_half_block: fragment-ordered weights missing for this dtype")
    check(_lib.lib().emsa_nbt_half_block_t(
        code, k, n, h, w, c, arr(xs), c, arr(wfa), arr(bias_a), arr(wfb), arr(bias_b), arr(scales),
        arr(shifts), arr(residuals), ld_of(rs[0]) if rs else 0, arr(outs), c, act, _stream()),
        'emsa_nbt_half_block_t')
    return tuple(outs)


def conv_fwd_pair(xs, wfrags, spec, biases=(None, None), scales=(None, None), shifts=(None, None),
                  residuals=(None, None), act=ACT_NONE):
    """two forward convs of ONE geometry (the rgb | depth encoders, the semantic | instance decoders)
    in one launch of the register-stationary kernel (emsa_conv1d_rs_pair_t, grid.y = 2); returns
    (out0, out1), or None where that kernel does not take the pair (other dtype / geometry / strides)
    -- the caller then launches the two convs one by one.  Each result is bit-identical to its own
    conv_fwd(..., wfrag=...) launch."""
    x0, x1 = xs
    if wfrags[0] is None or wfrags[1] is None or x0.dtype != x1.dtype or x0.shape != x1.shape:
        return None
    code = dt(x0)
    if code == 0 or wfrags[0].dtype != x0.dtype or wfrags[1].dtype != x0.dtype:
        return None
    n, c, h, w = x0.shape
    if ld_of(x0) != ld_of(x1):
        return None
    r0, r1 = residuals
    if (r0 is None) != (r1 is None) or (r0 is not None and ld_of(r0) != ld_of(r1)):
        return None
    for pr in (biases, scales, shifts):
        if (pr[0] is None) != (pr[1] is None):
            return None
    oh, ow = spec.out_hw(h, w)
    g = spec.geom_fwd(n, h, w, ld_of(x0), spec.cout)
    if not rs_supported(code, g):
        return None
    out0 = act_empty(n, spec.cout, oh, ow, x0.device, dtype=x0.dtype)
    out1 = act_empty(n, spec.cout, oh, ow, x0.device, dtype=x0.dtype)
    lr = ld_of(r0) if r0 is not None else 0
    check(_lib.lib().emsa_conv1d_rs_pair_t(
        code, g, _p(x0), _p(x1), _p(wfrags[0]), _p(wfrags[1]), _p(out0), _p(out1), _p(biases[0]),
        _p(biases[1]), _p(scales[0]), _p(scales[1]), _p(shifts[0]), _p(shifts[1]), _p(r0), _p(r1),
        lr, act, _stream()), 'emsa_conv1d_rs_pair_t')
    return out0, out1


def conv_igemm_pair(xs, wps, spec, biases=(None, None), scales=(None, None), shifts=(None, None),
                    residuals=(None, None), act=ACT_NONE):
    """conv_fwd_pair for the convs the implicit GEMM runs (strided, 1x1, 3x3; 16-bit storage):
    emsa_conv_igemm_pair_t, tap-split where the single launch would be.  (out0, out1) or None"""
    x0, x1 = xs
    if wps[0] is None or wps[1] is None or x0.dtype != x1.dtype or x0.shape != x1.shape:
        return None
    code = dt(x0)
    if code == 0 or wps[0].dtype != x0.dtype or wps[1].dtype != x0.dtype or ld_of(x0) != ld_of(x1):
        return None
    r0, r1 = residuals
    if (r0 is None) != (r1 is None) or (r0 is not None and ld_of(r0) != ld_of(r1)):
        return None
    for pr in (biases, scales, shifts):
        if (pr[0] is None) != (pr[1] is None):
            return None
    if os.environ.get('EMSA_CONVH_PF', '0') not in ('', '0'):
        return None
    n, c, h, w = x0.shape
    oh, ow = spec.out_hw(h, w)
    g = spec.geom_fwd(n, h, w, ld_of(x0), spec.cout)
    out0 = act_empty(n, spec.cout, oh, ow, x0.device, dtype=x0.dtype)
    out1 = act_empty(n, spec.cout, oh, ow, x0.device, dtype=x0.dtype)
    wsb = _splitk_ws_bytes(code, g)
    ws0 = _empty((wsb // 4,), x0.device) if wsb > 0 else None
    ws1 = _empty((wsb // 4,), x0.device) if wsb > 0 else None
    lr = ld_of(r0) if r0 is not None else 0
    check(_lib.lib().emsa_conv_igemm_pair_t(
        code, g, _p(x0), _p(x1), _p(wps[0]), _p(wps[1]), _p(out0), _p(out1), _p(biases[0]),
        _p(biases[1]), _p(scales[0]), _p(scales[1]), _p(shifts[0]), _p(shifts[1]), _p(r0), _p(r1),
        lr, act, _p(ws0), _p(ws1), _stream()), 'emsa_conv_igemm_pair_t')
    return out0, out1


# Strided data gradients by output phase (one dense stride-1 launch per parity class instead of ONE
# launch that visits every tap for every output pixel, half of them structurally zero at stride 2).
# Measured per training step: fp32 +1.2 % (the matrix-bound implicit GEMM does half the MACs: 12
# launches of ~230 us -> 24 of ~60); bf16 +-0 (latency-bound launches: two cost what one did).
# Default: fp32 only; EMSA_DGRAD_PHASES=0 / 1 forces it off / on for every storage type.
_DGRAD_PHASES_ENV = os.environ.get('EMSA_DGRAD_PHASES')
DGRAD_PHASES = None          # tests: True / False overrides the default rule


def dgrad_phases(dtype):
    if DGRAD_PHASES is not None:
        return DGRAD_PHASES
    if _DGRAD_PHASES_ENV is not None:
        return _DGRAD_PHASES_ENV != '0'
    return dtype == torch.float32


def _dgrad_phases(spec, h, w):
    """phase decomposition of a strided data gradient: output pixels (s_h*j + p_h, s_w*i + p_w) only
    meet the taps kh = kh0 + s_h*t with kh0 = (p_h + pad_h) % s_h -- a DENSE stride-1 convolution of
    dy with those taps.  -> [(p_h, p_w, khs, kws, off_h, off_w, rows, cols)]"""
    def axis(size, k, s_, pad):
        res = []
        for ph in range(s_):
            k0 = (ph + pad) % s_
            taps = list(range(k0, k, s_))
            n_out = (size - ph + s_ - 1) // s_ if size > ph else 0
            res.append((ph, taps, (ph + pad - k0) // s_, n_out))
        return res
    return [(ph, pw, khs, kws, oh_, ow_, nh, nw)
            for ph, khs, oh_, nh in axis(h, spec.kh, spec.sh, spec.ph)
            for pw, kws, ow_, nw in axis(w, spec.kw, spec.sw, spec.pw)]


def _conv_dgrad_phased(dy, wpd, spec, in_hw, mask_src, residual, out):
    n = dy.shape[0]
    h, w = in_hw
    phases = _dgrad_phases(spec, h, w)
    empty = [p_ for p_ in phases if not p_[2] or not p_[3]]
    if empty and (mask_src is not None or residual is not None):
        return None                       # (a phase without taps still owes the fused epilogue)
    if out is None:
        out = act_empty(n, spec.cin, h, w, dy.device, dtype=dy.dtype)
    if empty:
        out.zero_()
    L = _lib.lib()
    oh, ow = dy.shape[2], dy.shape[3]
    ld_dy, ld_dx = ld_of(dy), ld_of(out)
    lr = ld_of(residual) if residual is not None else 0
    lm = ld_of(mask_src) if mask_src is not None else 0
    taps = spec.kh * spec.kw
    wv = wpd.view(taps, spec.cin * spec.cout)
    for ph, pw, khs, kws, off_h, off_w, rows, cols in phases:
        if not khs or not kws or rows == 0 or cols == 0:
            continue
        idx = [kh * spec.kw + kw for kh in khs for kw in kws]
        if idx == list(range(idx[0], idx[0] + len(idx))):
            wp = wv[idx[0]:idx[0] + len(idx)]                     # contiguous taps: a view
        else:
            wp = torch.cat([wv[i:i + 1] for i in idx])       # (no host table: graph-capture safe)
        g = EmsaConvGeom(n, oh, ow, rows, cols, spec.cout, spec.cin, len(khs), len(kws),
                         1, off_h, -1, 1, 1, off_w, -1, 1,
                         oh * ow * ld_dy, ow * ld_dy, ld_dy, ld_dx,
                         h * w, spec.sh * w, spec.sw, ph * w + pw)
        check(call_t('emsa_conv_igemm', dt(dy), g, _p(dy), _p(wp), _p(out), None, None, None, None,
                     _p(residual), lr, _p(mask_src), lm, ACT_NONE, _stream()),
              'emsa_conv_igemm(dgrad phase)')
    return out


def conv_dgrad(dy, wpd, spec, in_hw, mask_src=None, residual=None, out=None, wino_u=None,
               mask_bits=None, wfrag=None):
    """dx = conv_transpose(dy); optional fused `* (mask_src > 0)` -- or the same mask as bits
    (`mask_bits`, Winograd kernel only) -- and `+ residual`."""
    n = dy.shape[0]
    h, w = in_hw
    code = dt(dy)
    if out is None:
        out = act_empty(n, spec.cin, h, w, dy.device, dtype=dy.dtype)
    g = spec.geom_dgrad(n, h, w, ld_of(dy), ld_of(out))
    L = _lib.lib()
    lr = ld_of(residual) if residual is not None else 0
    if code != 0:
        wino_u = None
    if wino_u is None and (spec.sh > 1 or spec.sw > 1):
        wpd = _resolve(wpd)
    if wino_u is None and (spec.sh > 1 or spec.sw > 1) and dgrad_phases(dy.dtype) \
            and wpd is not None and wpd.dtype == dy.dtype:
        r = _conv_dgrad_phased(dy, wpd, spec, in_hw, mask_src, residual, out)
        if r is not None:
            return r
    if wino_u is not None:
        if mask_bits is not None:
            mask_src = None
        lm = ld_of(mask_src) if mask_src is not None else 0
        check(L.emsa_conv1d_wino(g, _p(dy), _p(wino_u), _p(out), None, None, None, None,
                                 _p(residual), lr, _p(mask_src), lm, ACT_NONE, _p(mask_bits), None,
                                 _stream()), 'emsa_conv1d_wino(dgrad)')
    elif wfrag is not None and wfrag.dtype == dy.dtype and rs_supported(code, g):
        lm = ld_of(mask_src) if mask_src is not None else 0
        check(L.emsa_conv1d_rs_t(code, g, _p(dy), _p(wfrag), _p(out), None, None, None, None,
                                 _p(residual), lr, _p(mask_src), lm, ACT_NONE, _stream()),
              'emsa_conv1d_rs_t(dgrad)')
    else:
        wpd = _resolve(wpd)
        if wpd is None or wpd.dtype != dy.dtype:
            raise _lib.EmsaError("data-gradient weights are not packed in the gradient's dtype")
        lm = ld_of(mask_src) if mask_src is not None else 0
        check(call_t('emsa_conv_igemm', code, g, _p(dy), _p(wpd), _p(out), None, None, None, None,
                     _p(residual), lr, _p(mask_src), lm, ACT_NONE, _stream()),
              'emsa_conv_igemm(dgrad)')
    return out


# BatchNorm-backward reduction inside the data gradient in front of it (conv_dgrad_bnb): OFF by
# default in both storage types since round 5.  History: with the implicit GEMM (round 3) it gained
# 0.4 ms per bf16 step and was the 16-bit default; fp32 measured +-0 (the Winograd data gradient of
# the 64 / 128-channel stages is not purely matrix-bound: 124.3 vs 121.7 us per launch).  Since the
# 16-bit 1-D convs run on conv_rs.hip (round 4) the fused epilogue costs +48 / +25 / +10 / +8 us per
# launch at 64 / 128 / 256 / 512 channels (profiles/r05_z_bf16_one_stream_kernel_stats.md: 84.8 vs
# 36.3, 50.8 vs 26.6, 35.8 vs 26.8, 32.9 vs 25.6 us) where the separate reduction pass it replaces
# costs ~36 / 21 / 13 / 9 us.  A/B on one box, bf16 step replayed from a hipGraph, two runs each
# (profiles/r05_f_*): fused everywhere 855.1 / 852.7, at >= 128 channels 856.0 / 856.2, at >= 256
# 861.0 / 861.8, nowhere **863.3 / 864.3** images/s.
# EMSA_BN_FUSE=0 / 1 forces it off / on for every storage type; EMSA_BN_FUSE_MIN_C=n switches it on
# for 16-bit convs of >= n channels.
_BN_FUSE_ENV = os.environ.get('EMSA_BN_FUSE')
_BN_FUSE_MIN_C = int(os.environ.get('EMSA_BN_FUSE_MIN_C', '0'))


def bn_fused_reduce(dtype, c=None):
    if _BN_FUSE_ENV is not None:
        return _BN_FUSE_ENV != '0'
    if dtype == torch.float32 or _BN_FUSE_MIN_C <= 0:
        return False
    return c is None or c >= _BN_FUSE_MIN_C


# The NBt1D block's bn1 without a forward pass of its own (fp32 training): conv3x1_2 and its weight
# gradient form relu(bn1(y2)) in their loaders, the data gradient's epilogue recomputes the ReLU
# decisions and emits the backward sums (conv_dgrad_bnb).  Measured per block at bs=32
# (tools/conv_bench.py inbn, profiles/r03_b_conv_bench_inbn.txt): the fold removes the normalise pass
# (56 / 30 / 15 / 13 us at the /4 /8 /16 /32 stages) and the separate reduction pass (about the
# same) and costs +2..7 us on the forward conv, +4..7 us on the weight gradient (12 / 32 vector
# instructions per K step next to 16 / 32 MFMAs) and +12..21 us on the data gradient: a clear gain
# on the large tensors, a small loss at 512 channels / 15x20.  Default: fold when the BatchNorm's
# tensor is at least EMSA_BN1_FOLD_MIN_MB (24) MiB; EMSA_BN1_FOLD=0 / 1 forces never / always.
# tests: BN1_FOLD = True / False overrides (always / never).
_BN1_FOLD_ENV = os.environ.get('EMSA_BN1_FOLD')
_BN1_FOLD_MIN_BYTES = int(float(os.environ.get('EMSA_BN1_FOLD_MIN_MB', '24')) * (1 << 20))
BN1_FOLD = None


def bn1_fold(t):
    """fold the BatchNorm + ReLU applied to activation `t` into the loaders of the conv behind it?"""
    if t.dtype != torch.float32:
        return False
    # the weight gradient's loader fold exists for the exact-fp32 Winograd F(3,2) form only: not for
    # the direct form (EMSA_WGRAD_WINO=0) nor beside the opt-in bf16-MFMA fp32 mode
    if os.environ.get('EMSA_WGRAD_WINO') == '0':
        return False
    if BN1_FOLD is not None:
        return BN1_FOLD
    if _BN1_FOLD_ENV is not None:
        return _BN1_FOLD_ENV != '0'
    return t.numel() * 4 >= _BN1_FOLD_MIN_BYTES


# The same fold in 16-bit storage (round 6): conv_rs.hip's loader normalises the staged tile
# (emsa_conv1d_rs_inbn_t), the transposed-read weight gradient recomputes it (emsa_conv_wgrad_inbn_t /
# emsa_conv_wgrad_multi_inbn_t), and bn1's backward passes recompute the ReLU decisions from their
# input (emsa_bn_bwd_*_aff_t) -- the data gradient stays the plain conv_rs launch (its fused
# BatchNorm-backward epilogue costs more than the separate reduction, section 4.3 of DESIGN.md).
# What leaves the step: the bn_act_fwd pass of bn1 (52 launches) and its ReLU bit mask.
# Measured per block at bs 32 (tools/bn1_fold16_bench.py, profiles/r06_bn1_fold16_stages.txt; us at
# the /4 /8 /16 /32 stages): saved normalise pass 33.0 / 18.0 / 11.0 / 7.3; cost forward conv +3.0 /
# +2.4 / +5.9 / +5.5, weight gradient +1.3 / +3.5 / +5.6 / +6.0, bn1 backward -1.7 / -1.4 / -0.2 / -0.9
# (no bit mask to read) -> net +30.4 / +13.5 / -0.3 / -3.2: like the fp32 fold it pays on the large
# tensors only.  Default: fold when the BatchNorm's tensor is at least EMSA_BN1_FOLD16_MIN_MB (24) MiB
# (the /4 and /8 stages at bs 32); step: 983.6 vs 977.9 images/s folded everywhere (+0.6 %, 1,444 vs
# 1,494 graph nodes; profiles/r06_k_*).  EMSA_BN1_FOLD16=0 / 1 forces never / always; tests:
# BN1_FOLD16 = True / False.
_BN1_FOLD16_ENV = os.environ.get('EMSA_BN1_FOLD16')
_BN1_FOLD16_MIN_BYTES = int(float(os.environ.get('EMSA_BN1_FOLD16_MIN_MB', '24')) * (1 << 20))
BN1_FOLD16 = None


def bn1_fold16(t, spec):
    """fold the BatchNorm + ReLU applied to the bf16 activation `t` into the loader of the 3x1 conv
    `spec` behind it?"""
    if t.dtype != torch.bfloat16 or not CONV_RS or not rs_eligible(spec) or (spec.kh, spec.kw) != (3, 1):
        return False
    if BN1_FOLD16 is not None:
        on = BN1_FOLD16
    elif _BN1_FOLD16_ENV is not None:
        on = _BN1_FOLD16_ENV != '0'
    else:
        on = t.numel() * 2 >= _BN1_FOLD16_MIN_BYTES
    if not on or not deterministic_wgrad():
        return False
    n, c, h, w = t.shape
    g = spec.geom_fwd(n, h, w, ld_of(t), spec.cout)
    # (the weight gradient's fold lives in the transposed-read kernel: 16-byte rows, channels % 8)
    return rs_supported(dt(t), g) and c % 8 == 0 and ld_of(t) % 8 == 0


def bn_bwd_aff(dy, x, gamma, mean, invstd, affine, dg_out=None, db_out=None):
    """BatchNorm (batch statistics) + ReLU backward with the ReLU decisions recomputed from the
    BatchNorm's input x and the forward's folded (scale, shift): returns dx, dgamma, dbeta"""
    n, c, h, w = x.shape
    assert ld_of(x) == c and ld_of(dy) == c and dy.dtype == x.dtype
    L = _lib.lib()
    code = dt(x)
    rows = L.emsa_bn_bwd_rows(n * h * w, c)
    partial = _empty((2, rows, c), x.device)
    check(L.emsa_bn_bwd_reduce_aff_t(code, _p(dy), _p(x), _p(mean), _p(invstd), _p(affine[0]),
                                     _p(affine[1]), n, h * w, c, _p(partial), _stream()),
          'emsa_bn_bwd_reduce_aff_t')
    dx = act_empty(n, c, h, w, x.device, dtype=x.dtype)
    if dg_out is None or db_out is None:
        dgb = _empty((2, c), x.device)
        dg_out, db_out = dgb[0], dgb[1]
    check(L.emsa_bn_bwd_apply_aff_t(code, _p(dy), _p(x), _p(gamma), _p(mean), _p(invstd), _p(affine[0]),
                                    _p(affine[1]), _p(partial), rows, n, h * w, c, _p(dx), _p(dg_out),
                                    _p(db_out), _stream()), 'emsa_bn_bwd_apply_aff_t')
    return dx, dg_out, db_out


def conv_dgrad_bnb(dy, wpd, spec, in_hw, t, bn_scale, bn_shift, bn_mean, bn_invstd, residual=None,
                   wino_u=None, wfrag=None):
    """data gradient of the conv behind a BatchNorm+ReLU with that BatchNorm's backward reduction
    in the epilogue: returns (g, partial, rows) with g = dz * (relu(bn(t)) > 0) and `partial` the
    (sum g, sum g * xhat) rows for `bn_bwd_from_rows`; fp32 needs the Winograd weights `wino_u`"""
    n = dy.shape[0]
    h, w = in_hw
    code = dt(dy)
    assert t.dtype == dy.dtype and tuple(t.shape[2:]) == (h, w) and t.shape[1] == spec.cin
    out = act_empty(n, spec.cin, h, w, dy.device, dtype=dy.dtype)
    g = spec.geom_dgrad(n, h, w, ld_of(dy), ld_of(out))
    L = _lib.lib()
    lr = ld_of(residual) if residual is not None else 0
    use_rs = wfrag is not None and wfrag.dtype == dy.dtype and rs_supported(code, g)
    if code == 0:
        rows = L.emsa_conv1d_wino_stats_rows(g)
    elif use_rs:
        rows = g._rs[2]
    else:
        rows = L.emsa_conv_stats_rows_t(code, g)
    if rows <= 0:
        raise _lib.EmsaError(f"fused BatchNorm reduction: unsupported geometry (status {rows})")
    partial = _empty((2, rows + 16, spec.cin), dy.device)
    if code == 0:
        check(L.emsa_conv1d_wino_bnb(g, _p(dy), _p(wino_u), _p(out), _p(residual), lr, _p(t),
                                     ld_of(t), _p(bn_scale), _p(bn_shift), _p(bn_mean),
                                     _p(bn_invstd), _p(partial), rows + 16, _stream()),
              'emsa_conv1d_wino_bnb')
    elif use_rs:
        check(L.emsa_conv1d_rs_bnb_t(code, g, _p(dy), _p(wfrag), _p(out), _p(residual), lr, _p(t),
                                     ld_of(t), _p(bn_scale), _p(bn_shift), _p(bn_mean),
                                     _p(bn_invstd), _p(partial), rows + 16, _stream()),
              'emsa_conv1d_rs_bnb_t')
    else:
        wpd = _resolve(wpd)
        if wpd is None or wpd.dtype != dy.dtype:
            raise _lib.EmsaError("data-gradient weights are not packed in the gradient's dtype")
        check(L.emsa_conv_igemm_bnb_t(code, g, _p(dy), _p(wpd), _p(out), _p(residual), lr, _p(t),
                                      ld_of(t), _p(bn_scale), _p(bn_shift), _p(bn_mean),
                                      _p(bn_invstd), _p(partial), rows + 16, _stream()),
              'emsa_conv_igemm_bnb_t')
    return out, partial, rows


def bn_bwd_from_rows(g, x, gamma, mean, invstd, partial, rows, train, dg_out=None, db_out=None):
    """BatchNorm backward from the per-tile sums of `conv_dgrad_bnb` (g already carries the ReLU
    mask): returns dx, dgamma, dbeta"""
    n, c, h, w = x.shape
    assert ld_of(x) == c and ld_of(g) == c and g.dtype == x.dtype
    dx = act_empty(n, c, h, w, x.device, dtype=x.dtype)
    if dg_out is None or db_out is None:
        dgb = _empty((2, c), x.device)
        dg_out, db_out = dgb[0], dgb[1]
    check(_lib.lib().emsa_bn_bwd_apply_rows_t(dt(x), _p(g), _p(x), _p(gamma), _p(mean), _p(invstd),
                                              _p(partial), rows, n, h * w, c, 1 if train else 0,
                                              _p(dx), _p(dg_out), _p(db_out), _stream()),
          'emsa_bn_bwd_apply_rows_t')
    return dx, dg_out, db_out


def deterministic_wgrad():
    """Weight gradients of the 1-D convs go through the atomics-free two-pass kernel
    (bit-reproducible) by default: since the reduction pass keeps eight loads in flight it is at
    least as fast as fp32 atomics on every layer shape and saves the zero-fill and the unpack
    kernel (119.9 vs 120.6 ms per step).  EMSA_DETERMINISTIC=0 selects the atomics form."""
    return os.environ.get('EMSA_DETERMINISTIC', '1') != '0'


def conv_wgrad(x, dy, spec, want_bias, like=None, two_pass=None, dw_out=None, db_out=None,
               in_affine=None):
    """weight (+bias) gradient.  returns (dw, dbias or None, packed):
    packed=False: dw is already in the parameter layout [cout][cin][kh][kw] (deterministic
    two-pass kernel of the 1-D convs, needs `like` = the weight for the shape); `dw_out` /
    `db_out` (contiguous, parameter-shaped, 16-byte aligned: the flat gradient-bucket views)
    receive the result directly;
    packed=True: dw is the flat packed [tap][cout][cin] accumulator (-> unpack_wgrad).
    in_affine = (scale, shift): the conv's input was relu(x * scale + shift) formed in the forward
    loader (conv_fwd(in_affine=...)); the weight gradient forms it again (emsa_conv_wgrad_inbn)."""
    if two_pass is None:
        two_pass = deterministic_wgrad()
    n, c, h, w = x.shape
    g = spec.geom_fwd(n, h, w, ld_of(x), ld_of(dy))
    L = _lib.lib()
    taps = spec.kh * spec.kw
    nw = taps * spec.cout * spec.cin
    nb = spec.cout if want_bias else 0
    ws_bytes = call_t('emsa_conv_wgrad_ws_bytes', dt(x), g) if (like is not None and two_pass) else 0
    if ws_bytes > 0:
        ws = _empty((ws_bytes // 4,), x.device)
        if dw_out is not None and (db_out is not None or not want_bias):
            dw, db = dw_out.view(-1), db_out
        else:
            buf = _empty((nw + nb,), x.device)
            dw = buf[:nw]
            db = buf[nw:] if want_bias else None
    else:
        buf = torch.zeros(nw + nb, device=x.device, dtype=torch.float32)
        ws = None
        dw = buf[:nw]
        db = buf[nw:] if want_bias else None
    if x.dtype != dy.dtype:
        raise _lib.EmsaError(f"weight gradient of {x.dtype} activations with a {dy.dtype} gradient")
    if in_affine is not None and dt(x) != 0:
        if ws is None:
            raise _lib.EmsaError("in_affine (16-bit): the two-pass weight gradient only")
        check(L.emsa_conv_wgrad_inbn_t(dt(x), g, _p(x), _p(dy), _p(dw), _p(db), _p(ws),
                                       _p(in_affine[0]), _p(in_affine[1]), _stream()),
              'emsa_conv_wgrad_inbn_t')
    elif in_affine is not None:
        check(L.emsa_conv_wgrad_inbn(g, _p(x), _p(dy), _p(dw), _p(db), _p(ws), _p(in_affine[0]),
                                     _p(in_affine[1]), _stream()), 'emsa_conv_wgrad_inbn')
    else:
        check(call_t('emsa_conv_wgrad', dt(x), g, _p(x), _p(dy), _p(dw), _p(db), _p(ws), _stream()),
              'emsa_conv_wgrad')
    if ws is not None:
        return dw.view(like.shape), db, False
    return dw, db, True


# Weight gradients of one NBt1D block in ONE launch (bf16 storage; csrc/conv_mfma.hip
# emsa_conv_wgrad_multi_t): the block's four convs have the same channel count and pixel count, their
# weight gradients are off the critical path of the backward pass (nothing waits for them but the
# optimizer) and each alone is a launch of 64-256 output tiles split ~12x along K.  EMSA_WGRAD_MULTI=0
# keeps one launch per conv.
WGRAD_MULTI = os.environ.get('EMSA_WGRAD_MULTI', '1') != '0'
WGRAD_MULTI_MAX = 4


def wgrad_multi_eligible(x, spec):
    return (WGRAD_MULTI and x.dtype == torch.bfloat16 and deterministic_wgrad() and spec.sh == 1 and
            spec.sw == 1 and (spec.kh, spec.kw, spec.ph, spec.pw) in ((3, 1, 1, 0), (1, 3, 0, 1)) and
            spec.cin % 8 == 0 and spec.cout % 8 == 0)


def conv_wgrad_multi(jobs):
    """jobs: [(x, dy, spec, like, dw_out, db_out, want_bias[, in_affine])] of `wgrad_multi_eligible` convs
    with the same channel counts (in_affine = (scale, shift): that job's x is the input of a folded
    BatchNorm + ReLU, recomputed in the loader) -> [(dw (parameter layout), dbias or None)] per job, or None when the library
    has no multi-job form for this set (caller: one conv_wgrad per job)"""
    n_jobs = len(jobs)
    if n_jobs < 2 or n_jobs > WGRAD_MULTI_MAX:
        return None
    L = _lib.lib()
    geoms = (_lib.EmsaConvGeom * n_jobs)()
    for j, (x, dy, spec, *_rest) in enumerate(jobs):
        n, c, h, w = x.shape
        if x.dtype != dy.dtype:
            raise _lib.EmsaError(f"weight gradient of {x.dtype} activations with a {dy.dtype} gradient")
        g = spec.geom_fwd(n, h, w, ld_of(x), ld_of(dy))
        ctypes.memmove(ctypes.byref(geoms[j]), ctypes.byref(g), ctypes.sizeof(_lib.EmsaConvGeom))
    code = dt(jobs[0][0])
    ws_bytes = L.emsa_conv_wgrad_multi_ws_bytes(code, n_jobs, geoms)
    if ws_bytes <= 0:
        return None
    dev = jobs[0][0].device
    ws = _empty((ws_bytes // 4,), dev)
    arr = lambda: (ctypes.c_void_p * n_jobs)()          # noqa: E731
    a_in, a_dy, a_dw, a_db, a_sc, a_sh = arr(), arr(), arr(), arr(), arr(), arr()
    outs = []
    any_aff = False
    for j, (x, dy, spec, like, dw_out, db_out, want_bias, *aff) in enumerate(jobs):
        if aff and aff[0] is not None:
            a_sc[j], a_sh[j] = _p(aff[0][0]), _p(aff[0][1])
            any_aff = True
        nw = spec.kh * spec.kw * spec.cout * spec.cin
        if dw_out is not None and (db_out is not None or not want_bias):
            dw, db = dw_out.view(-1), db_out
        else:
            buf = _empty((nw + (spec.cout if want_bias else 0),), dev)
            dw = buf[:nw]
            db = buf[nw:] if want_bias else None
        a_in[j], a_dy[j], a_dw[j], a_db[j] = _p(x), _p(dy), _p(dw), _p(db)
        outs.append((dw.view(like.shape), db))
    if any_aff:
        check(L.emsa_conv_wgrad_multi_inbn_t(code, n_jobs, geoms, a_in, a_dy, a_dw, a_db, _p(ws), a_sc,
                                             a_sh, _stream()), 'emsa_conv_wgrad_multi_inbn_t')
    else:
        check(L.emsa_conv_wgrad_multi_t(code, n_jobs, geoms, a_in, a_dy, a_dw, a_db, _p(ws), _stream()),
              'emsa_conv_wgrad_multi_t')
    return outs


# ---------------------------------------------------------------------------------------------
# stem (7x7 s2 p3 on NCHW input)
# ---------------------------------------------------------------------------------------------
# EMSA_STEM_ROWS=0: the one-channel (depth) stem in the generic NHWC4 layout (seven row taps of 32,
# 28 of them zeros) instead of rows-as-channels (two super-taps) -- A/B switch
STEM_ROWS = os.environ.get('EMSA_STEM_ROWS', '1') != '0'


class StemSpec:
    """7x7/2 conv expressed as a 7x1 conv over the zero-padded NHWC4 image (k_ch = 32); with ONE
    input channel (`rows`): as a 2x1 conv over the rows-as-channels image (emsa_stem_*_rows*),
    2/7 of the matrix work"""

    def __init__(self, cin, cout=64):
        self.cin, self.cout = cin, cout
        self.rows = cin == 1 and STEM_ROWS
        self.taps = 2 if self.rows else 7
        self._geoms = {}

    def out_hw(self, h, w):
        return (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1

    def geom(self, n, h, w, ld_out):
        key = (n, h, w, ld_out)
        g = self._geoms.get(key)
        if g is None:
            oh, ow = self.out_hw(h, w)
            wp = w + 8
            if self.rows:
                hp = h + 4
                g = EmsaConvGeom(n, hp, wp, oh, ow, 32, self.cout, 2, 1,
                                 2, 0, 4, 1, 2, 0, 1, 1,
                                 hp * wp * 4, wp * 4, 4, ld_out)
            else:
                g = EmsaConvGeom(n, h, wp, oh, ow, 32, self.cout, 7, 1,
                                 2, -3, 1, 1, 2, 0, 1, 1,
                                 h * wp * 4, wp * 4, 4, ld_out)
            self._geoms[key] = g
        return g


def stem_pack_input(x_nchw, dtype=torch.float32):
    x = x_nchw.contiguous()
    n, c, h, w = x.shape
    if c == 1 and STEM_ROWS:
        xp = _empty((n, h + 4, w + 8, 4), x.device, dtype)
        check(_lib.lib().emsa_stem_pack_input_rows_t(DT[dtype], _p(x), _p(xp), n, h, w, _stream()),
              'emsa_stem_pack_input_rows_t')
        return xp
    xp = _empty((n, h, w + 8, 4), x.device, dtype)
    check(call_t('emsa_stem_pack_input', DT[dtype], _p(x), _p(xp), n, c, h, w, _stream()),
          'emsa_stem_pack_input')
    return xp


def stem_pack_weight(w, dtype=torch.float32):
    cout, cin = w.shape[:2]
    if cin == 1 and STEM_ROWS:
        wp = _empty(2 * cout * 32, w.device, dtype)
        check(_lib.lib().emsa_stem_pack_weight_rows_t(DT[dtype], _p(w), _p(wp), cout, _stream()),
              'emsa_stem_pack_weight_rows_t')
        return wp
    wp = _empty(7 * cout * 32, w.device, dtype)
    check(call_t('emsa_stem_pack_weight', DT[dtype], _p(w), _p(wp), cout, cin, _stream()),
          'emsa_stem_pack_weight')
    return wp


def stem_fwd(xp, wpk, spec, n, h, w, want_stats=True, bias=None):
    oh, ow = spec.out_hw(h, w)
    out = act_empty(n, spec.cout, oh, ow, xp.device, dtype=xp.dtype)
    g = spec.geom(n, h, w, spec.cout)
    code = dt(xp)
    stats = None
    if want_stats:
        rows = call_t('emsa_conv_stats_rows', code, g)
        stats = _empty((3, rows, spec.cout), xp.device)
    prof_flops(2.0 * n * oh * ow * spec.cout * 49 * spec.cin)
    check(call_t('emsa_conv_igemm', code, g, _p(xp), _p(wpk), _p(out), _p(bias), _p(stats), None,
                 None, None, 0, None, 0, ACT_NONE, _stream()), 'emsa_conv_igemm(stem)')
    return out, stats


def stem_fwd_folded(xp, wpk, spec, n, h, w, scale, shift, bias=None):
    oh, ow = spec.out_hw(h, w)
    out = act_empty(n, spec.cout, oh, ow, xp.device, dtype=xp.dtype)
    g = spec.geom(n, h, w, spec.cout)
    prof_flops(2.0 * n * oh * ow * spec.cout * 49 * spec.cin)
    check(call_t('emsa_conv_igemm', dt(xp), g, _p(xp), _p(wpk), _p(out), _p(bias), None, _p(scale),
                 _p(shift), None, 0, None, 0, ACT_RELU, _stream()),
          'emsa_conv_igemm(stem)')
    return out


def stem_wgrad(xp, dy, spec, n, h, w, like, out=None, want_bias=False):
    """-> dw (OIHW), or (dw, dbias) with want_bias"""
    g = spec.geom(n, h, w, ld_of(dy))
    nw = spec.taps * spec.cout * 32
    buf = torch.zeros(nw + (spec.cout if want_bias else 0), device=dy.device, dtype=torch.float32)
    dwp = buf[:nw]
    db = buf[nw:] if want_bias else None
    oh, ow = spec.out_hw(h, w)
    prof_flops(2.0 * n * oh * ow * spec.cout * 49 * spec.cin)
    check(call_t('emsa_conv_wgrad', dt(dy), g, _p(xp), _p(dy), _p(dwp), _p(db), None, _stream()),
          'emsa_conv_wgrad(stem)')
    dw = out if out is not None else _empty(tuple(like.shape), like.device)
    if spec.rows:
        check(_lib.lib().emsa_stem_unpack_wgrad_rows(_p(dwp), _p(dw), spec.cout, _stream()),
              'emsa_stem_unpack_wgrad_rows')
    else:
        check(_lib.lib().emsa_stem_unpack_wgrad(_p(dwp), _p(dw), spec.cout, spec.cin, _stream()),
              'emsa_stem_unpack_wgrad')
    return (dw, db) if want_bias else dw


# ---------------------------------------------------------------------------------------------
# batch norm
# ---------------------------------------------------------------------------------------------
def bn_finalize(stats, count, gamma, beta, eps, momentum, running_mean, running_var):
    c = gamma.shape[0]
    buf = _empty((4, c), gamma.device)
    ws = _empty((_lib.lib().emsa_bn_finalize_ws_bytes(c) // 8,), gamma.device, torch.float64)
    check(_lib.lib().emsa_bn_finalize(_p(stats), stats.shape[1], c, count, _p(gamma), _p(beta),
                                      eps, momentum, _p(running_mean), _p(running_var),
                                      _p(buf[0]), _p(buf[1]), _p(buf[2]), _p(buf[3]), _p(ws),
                                      _stream()), 'emsa_bn_finalize')
    return buf[0], buf[1], buf[2], buf[3]      # scale, shift, mean, invstd


def bn_fold(gamma, beta, running_mean, running_var, eps, out=None):
    """`out`: a (3, c) fp32 buffer to write into (callers that cache the fold keep its address)"""
    c = gamma.shape[0]
    buf = out if out is not None else _empty((3, c), gamma.device)
    check(_lib.lib().emsa_bn_fold(_p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, c,
                                  _p(buf[0]), _p(buf[1]), _p(buf[2]), _stream()), 'emsa_bn_fold')
    return buf[0], buf[1], buf[2]               # scale, shift, invstd


def bn_act(x, scale, shift, drop=None, residual=None, act=ACT_NONE, want_mask=False):
    """y = act((x*scale + shift)*drop + residual); want_mask: also (y > 0) as a bit mask
    (int64 words) that `bn_bwd` reads instead of y  -> y or (y, mask)"""
    n, c, h, w = x.shape
    assert ld_of(x) == c and (residual is None or ld_of(residual) == c)
    y = act_empty(n, c, h, w, x.device, dtype=x.dtype)
    L = _lib.lib()
    bits = None
    if want_mask and act == ACT_RELU:
        bits = torch.empty(L.emsa_relu_mask_words(n * c * h * w), device=x.device,
                           dtype=torch.int64)
    check(call_t('emsa_bn_act_fwd', dt(x), _p(x), _p(y), _p(scale), _p(shift), _p(drop),
                 _p(residual), n, h * w, c, act, _p(bits), _stream()), 'emsa_bn_act_fwd')
    return (y, bits) if want_mask else y


def bn_bwd(dy, y, x, gamma, mean, invstd, drop, act, train, want_dres, dg_out=None, db_out=None):
    """returns dx, dres (or None), dgamma, dbeta.  `y` = the activation output (ReLU mask y > 0)
    or the int64 bit mask `bn_act(..., want_mask=True)` produced.  dg_out / db_out: destinations
    of dgamma / dbeta (the parameters' flat gradient-bucket views)"""
    n, c, h, w = x.shape
    assert ld_of(x) == c and ld_of(dy) == c and dy.dtype == x.dtype
    L = _lib.lib()
    code = dt(x)
    bits = None
    if y is not None and y.dtype == torch.int64:
        y, bits = None, y
    rows = L.emsa_bn_bwd_rows(n * h * w, c)
    partial = _empty((2, rows, c), x.device)
    check(call_t('emsa_bn_bwd_reduce', code, _p(dy), _p(y), _p(bits), _p(x), _p(mean), _p(invstd),
                 _p(drop), n, h * w, c, act, _p(partial), _stream()), 'emsa_bn_bwd_reduce')
    dx = act_empty(n, c, h, w, x.device, dtype=x.dtype)
    dres = act_empty(n, c, h, w, x.device, dtype=x.dtype) if want_dres else None
    if dg_out is None or db_out is None:
        dgb = _empty((2, c), x.device)
        dg_out, db_out = dgb[0], dgb[1]
    check(call_t('emsa_bn_bwd_apply', code, _p(dy), _p(y), _p(bits), _p(x), _p(gamma), _p(mean),
                 _p(invstd), _p(drop), _p(partial), rows, n, h * w, c, act, 1 if train else 0,
                 _p(dx), _p(dres), _p(dg_out), _p(db_out), _stream()), 'emsa_bn_bwd_apply')
    return dx, dres, dg_out, db_out


def dropout2d_mask(n, c, p, seed, layer_id, device):
    """seed: int (host value) or an int32 device tensor {base seed, training step} -- the seed of the
    step is then formed on the device (hipGraph-captured training steps)"""
    m = _empty((n, c), device)
    if torch.is_tensor(seed):
        check(_lib.lib().emsa_dropout2d_mask_dev(_p(m), n, c, p, _p(seed), layer_id, _stream()),
              'emsa_dropout2d_mask_dev')
    else:
        check(_lib.lib().emsa_dropout2d_mask(_p(m), n, c, p, seed & 0xFFFFFFFF, layer_id,
                                             _stream()), 'emsa_dropout2d_mask')
    return m


def dropout2d_mask_batch(jobs_dev, n_jobs, total_floats, n, max_c, seed, device):
    """all Dropout2d masks of a training step in one launch -> flat fp32 buffer (layer j at
    jobs[j].offset); `seed` as in dropout2d_mask"""
    m = _empty((total_floats,), device)
    if torch.is_tensor(seed):
        check(_lib.lib().emsa_dropout2d_mask_batch(_p(m), _p(jobs_dev), n_jobs, n, max_c, 0, _p(seed),
                                                   _stream()), 'emsa_dropout2d_mask_batch')
    else:
        check(_lib.lib().emsa_dropout2d_mask_batch(_p(m), _p(jobs_dev), n_jobs, n, max_c,
                                                   seed & 0xFFFFFFFF, None, _stream()),
              'emsa_dropout2d_mask_batch')
    return m


# ---------------------------------------------------------------------------------------------
# pooling / SE / upsampling / PPM / heads
# ---------------------------------------------------------------------------------------------
def maxpool_fwd(x):
    n, c, h, w = x.shape
    assert ld_of(x) == c
    oh, ow = (h + 1) // 2, (w + 1) // 2
    y = act_empty(n, c, oh, ow, x.device, dtype=x.dtype)
    idx = _empty((n, oh, ow, c), x.device, torch.int8)
    check(call_t('emsa_maxpool3x3s2_fwd', dt(x), _p(x), _p(y), _p(idx), n, h, w, c, _stream()),
          'emsa_maxpool3x3s2_fwd')
    return y, idx


def maxpool_bwd(dy, idx, in_hw):
    n, c = dy.shape[:2]
    h, w = in_hw
    assert ld_of(dy) == c
    dx = act_empty(n, c, h, w, dy.device, dtype=dy.dtype)
    check(call_t('emsa_maxpool3x3s2_bwd', dt(dy), _p(dy), _p(idx), _p(dx), n, h, w, c, _stream()),
          'emsa_maxpool3x3s2_bwd')
    return dx


def channel_mean(x):
    n, c, h, w = x.shape
    assert ld_of(x) == c
    gap = _empty((n, c), x.device)
    ws = _empty((_lib.lib().emsa_channel_ws_floats(n, h * w, c),), x.device)
    check(call_t('emsa_channel_mean', dt(x), _p(x), _p(gap), _p(ws), n, h * w, c, _stream()),
          'emsa_channel_mean')
    return gap


def se_mlp_fwd(gap, w1, b1, w2, b2):
    n, c = gap.shape
    cr = w1.shape[0]
    hid = _empty((n, cr), gap.device)
    s = _empty((n, c), gap.device)
    check(_lib.lib().emsa_se_mlp_fwd(_p(gap), _p(w1), _p(b1), _p(w2), _p(b2), _p(hid), _p(s), n, c,
                                     cr, _stream()), 'emsa_se_mlp_fwd')
    return hid, s


def se_pair_fwd(xa, xb, wa, wb):
    """squeeze + excitation of both inputs of an SE-add fusion in two launches; wa / wb =
    (w1 [cr][c], b1, w2 [c][cr], b2) of each.  -> (gap_a, gap_b, hid_a, hid_b, s_a, s_b), bit-identical
    to channel_mean + se_mlp_fwd per input"""
    n, c, h, w = xa.shape
    assert xb.shape == xa.shape and xb.dtype == xa.dtype and ld_of(xa) == c and ld_of(xb) == c
    cr = wa[0].shape[0]
    dev = xa.device
    gap, hid, s = _empty((2, n, c), dev), _empty((2, n, cr), dev), _empty((2, n, c), dev)
    ws = _empty((2 * _lib.lib().emsa_channel_ws_floats(n, h * w, c),), dev)
    check(_lib.lib().emsa_se_pair_fwd_t(dt(xa), _p(xa), _p(xb), _p(ws), _p(wa[0]), _p(wa[1]),
                                        _p(wa[2]), _p(wa[3]), _p(wb[0]), _p(wb[1]), _p(wb[2]),
                                        _p(wb[3]), _p(gap), _p(hid), _p(s), n, h * w, c, cr,
                                        _stream()), 'emsa_se_pair_fwd_t')
    return gap[0], gap[1], hid[0], hid[1], s[0], s[1]


def se_mlp_bwd(gap, w1, w2, hid, s, ds):
    n, c = gap.shape
    cr = w1.shape[0]
    dev = gap.device
    dgap = _empty((n, c), dev)
    dw1 = _empty((cr, c), dev)
    db1 = _empty((cr,), dev)
    dw2 = _empty((c, cr), dev)
    db2 = _empty((c,), dev)
    check(_lib.lib().emsa_se_mlp_bwd(_p(gap), _p(w1), _p(w2), _p(hid), _p(s), _p(ds), _p(dgap),
                                     _p(dw1), _p(db1), _p(dw2), _p(db2), n, c, cr, _stream()),
          'emsa_se_mlp_bwd')
    return dgap, dw1, db1, dw2, db2


def se_scale_add(a, sa, b=None, sb=None):
    n, c, h, w = a.shape
    out = act_empty(n, c, h, w, a.device, dtype=a.dtype)
    check(call_t('emsa_se_scale_add_fwd', dt(a), _p(a), _p(sa), _p(b), _p(sb), _p(out), n, h * w,
                 c, _stream()), 'emsa_se_scale_add_fwd')
    return out


def se_scale_bwd_reduce(dout, x):
    n, c, h, w = x.shape
    assert dout.dtype == x.dtype
    ds = _empty((n, c), x.device)
    ws = _empty((_lib.lib().emsa_channel_ws_floats(n, h * w, c),), x.device)
    check(call_t('emsa_se_scale_bwd_reduce', dt(x), _p(dout), _p(x), _p(ds), _p(ws), n, h * w, c,
                 _stream()), 'emsa_se_scale_bwd_reduce')
    return ds


def se_scale_bwd_apply(dout, s, dgap, extra=None):
    n, c, h, w = dout.shape
    dx = act_empty(n, c, h, w, dout.device, dtype=dout.dtype)
    check(call_t('emsa_se_scale_bwd_apply', dt(dout), _p(dout), _p(s), _p(dgap), _p(extra), _p(dx),
                 n, h * w, c, _stream()), 'emsa_se_scale_bwd_apply')
    return dx


def _two(name, feat, out_f32, *args):
    """kernels at the model boundary: features in the dtype of `feat`, output-side tensors fp32
    when `out_f32` (always the case for an fp32 engine)"""
    code = dt(feat)
    L = _lib.lib()
    if code == 0:
        return getattr(L, name)(*args)
    return getattr(L, name + '_t')(code, 1 if out_f32 else 0, *args)


def up2x_dw_fwd(x, wdw, bias, skip=None, out_f32=False):
    """out_f32: write the result as fp32 from 16-bit features (last up-sampling of a head)"""
    n, c, h, w = x.shape
    assert ld_of(x) == c and (skip is None or (ld_of(skip) == c and skip.dtype == x.dtype))
    y = act_empty(n, c, 2 * h, 2 * w, x.device,
                  dtype=torch.float32 if out_f32 else x.dtype)
    check(_two('emsa_up2x_dw3x3_fwd', x, out_f32, _p(x), _p(wdw), _p(bias), _p(skip), _p(y), n, h,
               w, c, _stream()), 'emsa_up2x_dw3x3_fwd')
    return y


# EMSA_UP2X_FUSED=0: the two separate passes (data / weight gradient) for A/B measurements
UP2X_FUSED_BWD = os.environ.get('EMSA_UP2X_FUSED', '1') != '0'


def up2x_dw_fwd_pair(xs, wdws, biases, skips):
    """up2x_dw_fwd of two twin modules in one launch (emsa_up2x_dw3x3_fwd_pair_t: 16-bit features,
    same shapes); (y0, y1) or None where there is no twin form"""
    x0, x1 = xs
    if x0.dtype == torch.float32 or x0.dtype != x1.dtype or x0.shape != x1.shape:
        return None
    if (biases[0] is None) != (biases[1] is None) or (skips[0] is None) != (skips[1] is None):
        return None
    n, c, h, w = x0.shape
    for t in (x0, x1) + tuple(k for k in skips if k is not None):
        if ld_of(t) != c or t.dtype != x0.dtype:
            return None
    y0 = act_empty(n, c, 2 * h, 2 * w, x0.device, dtype=x0.dtype)
    y1 = act_empty(n, c, 2 * h, 2 * w, x0.device, dtype=x0.dtype)
    check(_lib.lib().emsa_up2x_dw3x3_fwd_pair_t(
        dt(x0), _p(x0), _p(x1), _p(wdws[0]), _p(wdws[1]), _p(biases[0]), _p(biases[1]), _p(skips[0]),
        _p(skips[1]), _p(y0), _p(y1), n, h, w, c, _stream()), 'emsa_up2x_dw3x3_fwd_pair_t')
    return y0, y1


def up2x_dw_bwd(dy, x, wdw, need_dx=True):
    n, c, h, w = x.shape
    assert ld_of(dy) == c
    out_f32 = dy.dtype == torch.float32
    if not out_f32 and dy.dtype != x.dtype:
        raise _lib.EmsaError(f"up-sampling gradient {dy.dtype} for {x.dtype} features")
    dx = act_empty(n, c, h, w, x.device, dtype=x.dtype) if need_dx else None
    dwb = torch.zeros(c * 10, device=x.device, dtype=torch.float32)
    dw, db = dwb[:c * 9], dwb[c * 9:]
    if UP2X_FUSED_BWD and _lib.lib().emsa_up2x_dw3x3_bwd_supported(c, x.element_size()):
        # one LDS-tiled pass over dy for dx, dw and db
        check(_two('emsa_up2x_dw3x3_bwd', x, out_f32, _p(dy), _p(x), _p(wdw), _p(dx), _p(dw), _p(db),
                   n, h, w, c, _stream()), 'emsa_up2x_dw3x3_bwd')
        return dx, dw, db
    if need_dx:
        check(_two('emsa_up2x_dw3x3_bwd_data', x, out_f32, _p(dy), _p(wdw), _p(dx), n, h, w, c,
                   _stream()), 'emsa_up2x_dw3x3_bwd_data')
    check(_two('emsa_up2x_dw3x3_bwd_weight', x, out_f32, _p(dy), _p(x), _p(dw), _p(db), n, h, w, c,
               _stream()), 'emsa_up2x_dw3x3_bwd_weight')
    return dx, dw, db


def adaptive_avgpool_fwd(x, bins):
    n, c, h, w = x.shape
    assert ld_of(x) == c
    y = act_empty(n, c, bins, bins, x.device, dtype=x.dtype)
    check(call_t('emsa_adaptive_avgpool_fwd', dt(x), _p(x), _p(y), n, h, w, c, bins, _stream()),
          'emsa_adaptive_avgpool_fwd')
    return y


def adaptive_avgpool_bwd(dy, dx, bins, accumulate):
    n, c, h, w = dx.shape
    assert dy.dtype == dx.dtype
    check(call_t('emsa_adaptive_avgpool_bwd', dt(dx), _p(dy), _p(dx), n, h, w, c, bins,
                 1 if accumulate else 0, _stream()), 'emsa_adaptive_avgpool_bwd')
    return dx


def bilinear_fwd(x, out):
    """x (N,C,ih,iw) dense -> out (N,C,oh,ow) possibly a channel slice"""
    n, c, ih, iw = x.shape
    oh, ow = out.shape[2:]
    assert ld_of(x) == c and out.dtype == x.dtype
    check(call_t('emsa_bilinear_fwd', dt(x), _p(x), _p(out), n, ih, iw, oh, ow, c, ld_of(out),
                 _stream()), 'emsa_bilinear_fwd')
    return out


def bilinear_bwd(dy, in_hw):
    """gather-form backward (one thread per dx element, fixed summation order: bit-reproducible) into
    an fp32 dx, returned in the dtype of dy"""
    n, c, oh, ow = dy.shape
    ih, iw = in_hw
    dx = act_empty(n, c, ih, iw, dy.device)
    check(call_t('emsa_bilinear_bwd', dt(dy), _p(dy), _p(dx), n, ih, iw, oh, ow, c, ld_of(dy),
                 _stream()), 'emsa_bilinear_bwd')
    if dy.dtype != torch.float32:
        dx = cast(dx, dy.dtype)
    return dx


def nearest_fwd(x, out):
    """x (N,C,ih,iw) dense -> out (N,C,oh,ow) possibly a channel slice: F.interpolate(mode='nearest')"""
    n, c, ih, iw = x.shape
    oh, ow = out.shape[2:]
    assert ld_of(x) == c and out.dtype == x.dtype
    check(_lib.lib().emsa_nearest_fwd_t(dt(x), _p(x), _p(out), n, ih, iw, oh, ow, c, ld_of(out),
                                        _stream()), 'emsa_nearest_fwd_t')
    return out


def nearest_bwd(dy, in_hw):
    """gather-form backward of `nearest_fwd` into an fp32 dx, returned in the dtype of dy"""
    n, c, oh, ow = dy.shape
    ih, iw = in_hw
    dx = act_empty(n, c, ih, iw, dy.device)
    check(_lib.lib().emsa_nearest_bwd_t(dt(dy), _p(dy), _p(dx), n, ih, iw, oh, ow, c, ld_of(dy),
                                        _stream()), 'emsa_nearest_bwd_t')
    if dy.dtype != torch.float32:
        dx = cast(dx, dy.dtype)
    return dx


def add(a, b):
    """a + b of two dense activations of one storage type (fp32 sum, rounded once)"""
    assert a.shape == b.shape and a.dtype == b.dtype and ld_of(a) == a.shape[1] == ld_of(b)
    n, c, h, w = a.shape
    out = act_empty(n, c, h, w, a.device, dtype=a.dtype)
    check(_lib.lib().emsa_add_t(dt(a), _p(a), _p(b), _p(out), a.numel(), _stream()), 'emsa_add_t')
    return out


def head_act_fwd(x, n_sig, n_tanh, n_norm=0, norm_off=3):
    """16-bit features -> fp32 outputs (the model's instance outputs), fp32 -> fp32"""
    n, c, h, w = x.shape
    assert ld_of(x) == c
    y = act_empty(n, c, h, w, x.device)
    check(_two('emsa_head_act_fwd', x, True, _p(x), _p(y), n * h * w, c, n_sig, n_tanh, norm_off,
               n_norm, _stream()), 'emsa_head_act_fwd')
    return y


def head_act_bwd(dy, y, n_sig, n_tanh, n_norm=0, x=None, norm_off=3, dtype=torch.float32):
    """dy, y fp32 (output side) -> dx in `dtype` (the features' storage type)"""
    n, c, h, w = y.shape
    assert ld_of(dy) == c and dy.dtype == torch.float32 and y.dtype == torch.float32
    dx = act_empty(n, c, h, w, y.device, dtype=dtype)
    check(_two('emsa_head_act_bwd', dx, True, _p(dy), _p(y), _p(x), _p(dx), n * h * w, c, n_sig,
               n_tanh, norm_off, n_norm, _stream()), 'emsa_head_act_bwd')
    return dx


# EMSA_HEAD_GATHER=0: the instance head's backward as padded copy + emsa_head_act_bwd_t (A/B)
HEAD_GATHER = os.environ.get('EMSA_HEAD_GATHER', '1') != '0'


def head_act_bwd_gather(grads, sizes, y, n_sig, n_tanh, n_norm=0, x=None, norm_off=3,
                        dtype=torch.float32):
    """head_act_bwd with the task gradients gathered inside the kernel: grads[k] (or None) is the fp32
    gradient of the sizes[k] channels behind the previous tasks' (<= 3 tasks, 8-channel head) -> dx in
    `dtype`, or None when the kernel does not take the case (caller: copy + head_act_bwd)"""
    n, c, h, w = y.shape
    if c != 8 or len(sizes) > 3 or y.dtype != torch.float32 or ld_of(y) != c or not HEAD_GATHER:
        return None
    gs = []
    for g_ in grads:
        if g_ is None:
            gs.append(None)
            continue
        g_ = as_act(g_ if g_.dtype == torch.float32 else g_.float())
        if tuple(g_.shape[2:]) != (h, w):
            return None
        gs.append(g_)
    gs += [None] * (3 - len(gs))
    cs = list(sizes) + [0] * (3 - len(sizes))
    dx = act_empty(n, c, h, w, y.device, dtype=dtype)
    args = []
    for g_, ck in zip(gs, cs):
        args += [_p(g_), ld_of(g_) if g_ is not None else 0, ck]
    check(_lib.lib().emsa_head_act_bwd_gather_t(DT[dtype], *args, _p(y), _p(x), _p(dx), n * h * w, c,
                                                n_sig, n_tanh, norm_off, n_norm, _stream()),
          'emsa_head_act_bwd_gather_t')
    return dx


def copy_channels(src, dst):
    """copy an activation (possibly a channel slice) into another (possibly a slice); the storage
    types may differ (conversion on the fly)."""
    n, c, h, w = src.shape
    if src.dtype == dst.dtype == torch.float32:
        check(_lib.lib().emsa_copy_channels(_p(src), ld_of(src), _p(dst), ld_of(dst), n * h * w, c,
                                            _stream()), 'emsa_copy_channels')
    else:
        check(_lib.lib().emsa_cast_channels(dt(src), _p(src), ld_of(src), dt(dst), _p(dst),
                                            ld_of(dst), n * h * w, c, _stream()),
              'emsa_cast_channels')
    return dst


def cast(x, dtype):
    """dense copy of an activation in another storage type"""
    x = as_act(x)
    n, c, h, w = x.shape
    return copy_channels(x, act_empty(n, c, h, w, x.device, dtype=dtype))


def axpy_(y, x, alpha=1.0):
    check(_lib.lib().emsa_axpy(_p(x), _p(y), y.numel(), alpha, _stream()), 'emsa_axpy')
    return y
