/*
 * emsanet_hip.h -- C-ABI of libemsanet_hip.so (gfx950 / MI355X).
 *
 * The reference (TUI-NICR/EMSANet) has no FFI or plugin interface: its hot path is a Python
 * nn.Module (`emsanet/model.py:27-233`, `emsanet/decoder.py:32-201`) whose layers come from the
 * un-vendored `nicr_mt_scene_analysis.model.*` and run as cuDNN/ATen eager kernels
 * (`main.py:23-24`).  This header is therefore the boundary the BUILD defines underneath that
 * module surface (last bullet): one entry point per fused op, each citing the
 * reference-side module call it stands in for.  `emsanet_amd/_lib.py` binds exactly these
 * symbols with ctypes; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - all tensors are fp32, device memory, NHWC ("pixel rows" of C contiguous channels);
 *     `ld*` = distance in elements between consecutive pixels (>= C; lets a kernel read or
 *     write a channel slice of a wider tensor, i.e. concat/split without a copy)
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous on it, allocates
 *     nothing, and is re-entrant per stream
 *   - return value: 0 = launched, negative = rejected (EMSA_E_*)
 */
#ifndef EMSANET_HIP_H
#define EMSANET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMSA_OK 0
#define EMSA_E_SHAPE (-1)   /* unsupported geometry (e.g. channel count not a multiple of 4) */
#define EMSA_E_ARG (-2)     /* null pointer / inconsistent arguments */
#define EMSA_E_LAUNCH (-3)  /* hipGetLastError() != hipSuccess after the launch */

#define EMSA_ACT_NONE 0
#define EMSA_ACT_RELU 1

/* storage type of the activation tensors of the `_t` ("typed") entry points: BASELINE configs[2]
 * (bf16 mixed-precision training) and configs[4] (16-bit inference).  Arithmetic is fp32 in
 * registers / fp32 MFMA accumulation everywhere; parameters, BatchNorm statistics, SE vectors and
 * weight gradients stay fp32.  Every `emsa_X_t(dtype, ...)` has the argument list of `emsa_X(...)`
 * with the activation pointers typed `void*`; EMSA_DT_F32 selects the fp32 kernels. */
#define EMSA_DT_F32 0
#define EMSA_DT_BF16 1
#define EMSA_DT_F16 2

/* library identity: returns the gfx arch string the kernels were compiled for ("gfx950") */
const char* emsa_arch(void);
int emsa_version(void);
/* batch-invariant reductions (default off): with 1 the per-image pixel partition of the SE / channel
 * reductions depends on the map size only, so a sample's forward does not depend on the batch it sits
 * in (the default rule re-partitions small launches for latency).  Returns the previous setting.
 * No reference counterpart: ATen makes no such promise (emsanet/model.py:192-233 runs eager ATen). */
int emsa_set_batch_invariant(int on);

/* ------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA (v_mfma_f32_32x32x2_f32).
 * Stands in for every nn.Conv2d of the model: NBt1D 3x1/1x3 (`get_block_class('nonbottleneck1d')`,
 * emsanet/model.py:49-52, decoder.py:69-72), decoder 3x3, 1x1 (downsample, skip fusion, PPM, side
 * heads), the 7x7 stem (through emsa_stem_pack_input) and nn.Linear of the scene head
 * (decoder.py:191-199).
 *
 * Gather geometry (shared by forward, data-gradient and weight-gradient):
 *   GEMM row m  <-> pixel (img, oh, ow) of the "out" grid  [n_img x out_h x out_w]
 *   tap (kh,kw) reads the "in" grid at
 *        ih = (oh*mul_h + off_h + kh*step_h) / div_h   (tap skipped when not divisible or OOB)
 *        iw = (ow*mul_w + off_w + kw*step_w) / div_w
 *   forward conv:   mul=stride, off=-pad, step=dilation, div=1
 *   data gradient:  mul=1, off=+pad, step=-dilation, div=stride  (in/out grids swapped)
 *   element address of channel c of that pixel:
 *        in + img*in_img_stride + ih*in_row_stride + iw*in_px_stride + c
 * ------------------------------------------------------------------------------------------ */
typedef struct EmsaConvGeom {
  int32_t n_img;
  int32_t in_h, in_w;          /* grid that is gathered from */
  int32_t out_h, out_w;        /* grid that is produced (GEMM rows) */
  int32_t k_ch;                /* channels gathered per tap (GEMM K per tap), multiple of 4 */
  int32_t n_ch;                /* channels produced (GEMM N) */
  int32_t kh, kw;
  int32_t mul_h, off_h, step_h, div_h;
  int32_t mul_w, off_w, step_w, div_w;
  int64_t in_img_stride, in_row_stride;   /* elements */
  int32_t in_px_stride;                   /* elements */
  int32_t ld_out;                         /* pixel stride of the produced tensor */
  /* Optional output pixel map (all four 0 = dense: row m is pixel m of the produced tensor).
   * Otherwise GEMM row (img, oh, ow) is stored at pixel
   *     img*out_pix_img + oh*out_pix_row + ow*out_pix_px + out_pix_off
   * of the produced tensor (and read there from `residual` / `mask_src`): a launch that produces
   * every s-th row / column of a larger tensor -- one PHASE of a strided data gradient, which is a
   * dense stride-1 convolution over the taps of its parity (emsa_conv_igemm and emsa_conv_igemm_t;
   * the Winograd and weight-gradient entry points return EMSA_E_SHAPE for a mapped geometry). */
  int32_t out_pix_img, out_pix_row, out_pix_px, out_pix_off;
} EmsaConvGeom;

/* out[m][n] = epilogue( sum_{tap,c} in[gather(m,tap)][c] * w[tap][n][c] )
 *   w        packed [kh*kw][n_ch][k_ch]  (emsa_pack_weight_*)
 *   bias     [n_ch] or NULL: v = acc + bias[n]
 *   stats    NULL, or float[3][gridM][n_ch]: per-M-tile sum of v, M2 = sum (v - tile mean)^2 and
 *            row count (BatchNorm batch statistics; gridM = emsa_conv_stats_rows(geom));
 *            merged by emsa_bn_finalize (Chan et al., fp64) -- no E[x^2]-mean^2 cancellation
 *   scale/shift [n_ch] or NULL: v = v*scale[n] + shift[n]     (folded eval-mode BatchNorm)
 *   residual NULL or tensor with pixel stride ld_res: v += residual[m][n]
 *   mask_src NULL or tensor with pixel stride ld_mask: v = mask_src[m][n] > 0 ? v : 0
 *            (ReLU backward fused into the data-gradient epilogue)
 *   act      EMSA_ACT_*                                                                     */
int emsa_conv_igemm(const EmsaConvGeom* g, const float* in, const float* w, float* out,
                    const float* bias, float* stats, const float* scale, const float* shift,
                    const float* residual, int32_t ld_res, const float* mask_src,
                    int32_t ld_mask, int32_t act, void* stream);
/* number of M tiles (rows of the stats partial buffer) emsa_conv_igemm will use for `g` */
int emsa_conv_stats_rows(const EmsaConvGeom* g);

/* weight gradient  dw[tap][n][c] = sum_m dout[m][n] * in[gather(m,tap)][c],  dbias[n] = sum_m dout
 *   dout   gradient of the produced tensor, pixel stride g->ld_out
 *   dbias  NULL or [n_ch]
 *   ws     NULL: split-K partial sums are accumulated with fp32 atomics into dw in the packed
 *          layout [tap][n][c] (emsa_unpack_wgrad converts) and into dbias; BOTH MUST BE ZEROED by
 *          the caller.
 *          non-NULL (only when emsa_conv_wgrad_ws_bytes(g) > 0: the stride-1 3-tap 1-D and 3x3 convs;
 *          at least that many bytes): deterministic two-pass form -- the partial tiles are stored
 *          to ws and a reduce kernel writes dw directly in the reference's OIHW parameter layout
 *          [n][c][tap] and dbias; no zeroing, no unpack pass, bit-reproducible.                */
int64_t emsa_conv_wgrad_ws_bytes(const EmsaConvGeom* g);
int emsa_conv_wgrad(const EmsaConvGeom* g, const float* in, const float* dout, float* dw,
                    float* dbias, float* ws, void* stream);

/* Winograd F(2,3) variant of emsa_conv_igemm for the stride-1 "same" convolutions with a 3-tap
 * row: the 3x1 / 1x3 convs of the NBt1D blocks and the 3x3 convs of the decoders / heads (forward
 * and data gradient).  Along the 3-tap direction two outputs cost 4 products instead of 6: 4 MFMA
 * GEMMs over half the pixels instead of 3 over all (1.5x fewer matrix instructions, fp32-exact
 * coefficients 1 and 1/2); a 3x3 conv is transformed along W and its three kernel rows are part of
 * the GEMM K dimension (K = 3 * k_ch).
 * Same arguments and fused epilogue as emsa_conv_igemm, but `u` = transformed weights
 * [4][n_ch][rows * k_ch] (rows = 1 or 3 kernel rows) from emsa_pack_wino (from the OIHW parameter;
 * u_dgrad: data-gradient weights, taps flipped, channels transposed) or emsa_pack_wino_packed
 * (from the packed [tap][n][k] layout, flip = 1 for a data-gradient pack);
 * stats partial rows = emsa_conv1d_wino_stats_rows(g).                                          */
int emsa_conv1d_wino_supported(const EmsaConvGeom* g);
int emsa_conv1d_wino_stats_rows(const EmsaConvGeom* g);
/* ReLU masks as bits (1/32 of the traffic of a float mask tensor): relu_bits (out, may be NULL)
 * receives (result > 0) as uint64[pixels][ceil(n_ch/64)] -- one word per pixel and 64-channel
 * tile, bit = (channel % 4) * 16 + (channel % 64) / 4; mask_bits (in, may be NULL; excludes
 * mask_src) applies such a mask of the producing layer like mask_src does.
 * emsa_conv_relu_bits_words(pixels, n_ch) = number of words of such a mask.                     */
int64_t emsa_conv_relu_bits_words(int64_t pixels, int32_t n_ch);
int emsa_conv1d_wino(const EmsaConvGeom* g, const float* in, const float* u, float* out,
                     const float* bias, float* stats, const float* scale, const float* shift,
                     const float* residual, int32_t ld_res, const float* mask_src,
                     int32_t ld_mask, int32_t act, const uint64_t* mask_bits,
                     uint64_t* relu_bits, void* stream);
/* Data gradient of the conv BEHIND a BatchNorm+ReLU with that BatchNorm's backward reduction fused
 * into the epilogue (the NBt1D block's conv3x1_2 -> bn1: one pass over dz and t less per block).
 * dz = conv_transpose(dy) is the gradient w.r.t. a = relu(t * bn_scale + bn_shift); the kernel
 * stores g = dz * (a > 0) (+ residual before the mask) to `out` and writes, per pixel tile,
 *   partial[0][tile][c] = sum g,   partial[1][tile][c] = sum g * (t - bn_mean) * bn_invstd
 * i.e. the rows emsa_bn_bwd_reduce would produce.  partial = float[2][rows_alloc][n_ch] with
 * rows_alloc = emsa_conv1d_wino_stats_rows(g) + 16; emsa_bn_bwd_apply_rows_t turns (g, partial)
 * into the BatchNorm's dx, dgamma, dbeta.  16-bit twin: emsa_conv_igemm_bnb_t.                     */
int emsa_conv1d_wino_bnb(const EmsaConvGeom* g, const float* dy, const float* u, float* out,
                         const float* residual, int32_t ld_res, const float* t, int32_t ld_t,
                         const float* bn_scale, const float* bn_shift, const float* bn_mean,
                         const float* bn_invstd, float* partial, int32_t rows_alloc, void* stream);
/* The NBt1D block's bn1 folded into its consumers (/root/reference/emsanet/model.py:47-58: conv1x3_1
 * -> BatchNorm -> ReLU -> conv3x1_2): the forward conv and its weight gradient read the BatchNorm's
 * INPUT `in` and form a = relu(in * in_scale[c] + in_shift[c]) (c = input channel, k_ch floats each;
 * the affine form emsa_bn_finalize produces) where the K step enters LDS, with zero padding applied
 * to `a`.  The normalised tensor is never written: one read + one write of the activation less per
 * block forward; emsa_conv1d_wino_bnb recomputes the same ReLU decisions for the backward pass.
 * 1-D stride-1 3-tap convs only (EMSA_E_SHAPE otherwise); other arguments as emsa_conv1d_wino /
 * emsa_conv_wgrad.                                                                                */
int emsa_conv1d_wino_inbn(const EmsaConvGeom* g, const float* in, const float* u, float* out,
                          const float* bias, float* stats, const float* in_scale,
                          const float* in_shift, int32_t act, uint64_t* relu_bits, void* stream);
int emsa_conv_wgrad_inbn(const EmsaConvGeom* g, const float* in, const float* dout, float* dw,
                         float* dbias, float* ws, const float* in_scale, const float* in_shift,
                         void* stream);
/* u (forward weights [4][cout][rows*cin]) and/or u_dgrad (data-gradient weights
 * [4][cin][rows*cout]) from the OIHW taps in one launch; either output may be NULL              */
int emsa_pack_wino(const float* w_oihw, float* u, float* u_dgrad, int32_t cout, int32_t cin,
                   int32_t rows, void* stream);
int emsa_pack_wino_packed(const float* w_packed, float* u, int32_t n_ch, int32_t k_ch,
                          int32_t rows, int32_t flip, void* stream);
/* All weight transforms of a model in ONE launch: a table of jobs in DEVICE memory, job j owns the
 * workgroups [first_block_j, first_block_(j+1)) of a grid of total_blocks (first_block ascending
 * from 0).  kind 0: OIHW -> the packed layouts of emsa_pack_weight_pair (dst0 forward, dst1 data
 * gradient); kind 1: OIHW -> the Winograd weights of emsa_pack_wino (dst0 = u, dst1 = u_dgrad).
 * NULL destinations are skipped.                                                                */
typedef struct EmsaPackJob {
  const float* src;
  float* dst0;
  float* dst1;
  int32_t cout, cin, kh, kw;
  int32_t kind;
  int32_t first_block;
  /* placement inside a wider destination (channel padding, block-diagonal merged heads; the rest
   * of the destination is never written: zero it once): the parameter's (co, ci) lands at
   * (co + cout_off, ci + cin_off) of a cout_total x cin_total operand.  kind 4: copy the `cout`
   * floats of `src` (a bias vector) to dst0 + cout_off.                                          */
  int32_t cout_total, cout_off, cin_total, cin_off;
} EmsaPackJob;
int emsa_pack_batch(const EmsaPackJob* jobs_device, int32_t n_jobs, int32_t total_blocks,
                    void* stream);

/* weight layout transforms between the reference's OIHW parameters and the packed layouts.
 * The packed buffer may be wider than the parameter (cout_total/cin_total >= cout/cin) and the
 * parameter is placed at (cout_off, cin_off): used to zero-pad channel counts to a multiple of 4
 * (instance head 5 -> 8, scene head 10 -> 12) and to lay the three `task_convs` of the instance
 * head (emsanet/weights.py:49-52) out as one block-diagonal 96 -> 8 convolution.              */
int emsa_pack_weight_fwd(const float* w_oihw, float* w_packed, int32_t cout, int32_t cin,
                         int32_t kh, int32_t kw, int32_t cout_total, int32_t cout_off,
                         int32_t cin_total, int32_t cin_off,
                         void* stream);   /* -> [tap][cout_total][cin_total] */
int emsa_pack_weight_dgrad(const float* w_oihw, float* w_packed, int32_t cout, int32_t cin,
                           int32_t kh, int32_t kw, int32_t cout_total, int32_t cout_off,
                           int32_t cin_total, int32_t cin_off,
                           void* stream); /* -> [tap][cin_total][cout_total] */
/* both packed layouts of one parameter in a single launch (training: forward + data gradient) */
int emsa_pack_weight_pair(const float* w_oihw, float* w_packed_fwd, float* w_packed_dgrad,
                          int32_t cout, int32_t cin, int32_t kh, int32_t kw, void* stream);
int emsa_unpack_wgrad(const float* dw_packed, float* dw_oihw, int32_t cout, int32_t cin,
                      int32_t kh, int32_t kw, int32_t cout_total, int32_t cout_off,
                      int32_t cin_total, int32_t cin_off,
                      void* stream);      /* [tap][cout_total][cin_total] -> OIHW */

/* 7x7 stride-2 stem (`get_backbone(... n_input_channels=3|1)`, emsanet/model.py:47-74):
 * NCHW input -> zero-padded NHWC4 [n][h][w+8][4] (3 left, 5 right columns) so that the 7 kw taps
 * x 4 channels of one kh row are 32 contiguous floats: the stem runs on emsa_conv_igemm as a
 * 7x1 conv with k_ch = 32.  Weights: OIHW [64][c][7][7] <-> packed [7][64][32].               */
int emsa_stem_pack_input(const float* x_nchw, float* x_pad, int32_t n, int32_t c, int32_t h,
                         int32_t w, void* stream);
int emsa_stem_pack_weight(const float* w_oihw, float* w_packed, int32_t cout, int32_t cin,
                          void* stream);
int emsa_stem_unpack_wgrad(const float* dw_packed, float* dw_oihw, int32_t cout, int32_t cin,
                           void* stream);
/* One-channel stem (depth encoder) in the rows-as-channels layout: the four slots of a packed
 * pixel hold four consecutive ROWS of the input plane, xp [n][h+4][w+8][4] with
 * xp[n][r'][col][r] = x[n][0][r'-3+r][col-3]; the 7x7 kernel becomes two super-taps (kh = 4*tap + r),
 * packed weights [2][cout][32 = 8 kw x 4 r].  Geometry for emsa_conv_igemm(_t) / emsa_conv_wgrad(_t):
 * in (h+4) x (w+8), k_ch 32, kh 2, kw 1, mul_h 2, off_h 0, step_h 4, mul_w 2, off_w 0: 2/7 of the
 * matrix work of the generic stem for the same result. */
int emsa_stem_pack_input_rows_t(int32_t dtype, const float* x, void* xp, int32_t n, int32_t h,
                                int32_t w, void* stream);
int emsa_stem_pack_weight_rows_t(int32_t dtype, const float* w, void* wp, int32_t cout, void* stream);
int emsa_stem_unpack_wgrad_rows(const float* dwp, float* dw, int32_t cout, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm (+ReLU, +Dropout2d, +residual add)  -- nn.BatchNorm2d / activation / nn.Dropout2d /
 * `out + identity` of the NBt1D block and of every ConvNormAct.
 * ------------------------------------------------------------------------------------------ */
/* merge the conv's stats partials [3][rows][c] -> batch mean/var; writes scale = gamma*invstd,
 * shift = beta - mean*scale, save_mean, save_invstd; updates running stats in place
 * (momentum, unbiased variance) when running_mean != NULL.  count = pixels per channel.      */
int emsa_bn_finalize(const float* stats, int32_t rows, int32_t c, int64_t count,
                     const float* gamma, const float* beta, float eps, float momentum,
                     float* running_mean, float* running_var, float* scale, float* shift,
                     float* save_mean, float* save_invstd, void* ws, void* stream);
/* bytes of the fp64 scratch `ws` emsa_bn_finalize needs for c channels */
int emsa_bn_finalize_ws_bytes(int32_t c);
/* eval mode: scale/shift (and optionally invstd, may be NULL) from running statistics */
int emsa_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                 const float* running_var, float eps, int32_t c, float* scale, float* shift,
                 float* save_invstd, void* stream);
/* y = act( (x*scale[c] + shift[c]) * drop[n][c] + residual ), drop/residual may be NULL.
 * mask_bits (may be NULL): uint64[emsa_relu_mask_words(n_img*hw*c)], receives (y > 0) as 1 bit per
 * element -- the backward passes below read it INSTEAD of y (1/32 of the traffic)               */
int64_t emsa_relu_mask_words(int64_t elements);
int emsa_bn_act_fwd(const float* x, float* y, const float* scale, const float* shift,
                    const float* drop, const float* residual, int32_t n_img, int64_t hw,
                    int32_t c, int32_t act, uint64_t* mask_bits, void* stream);
/* backward, pass 1: g = dy * (y>0 if act; from mask_bits when given, then y may be NULL);
 * partial[2][rows][c] of  sum g*drop  and
 * sum g*drop*xhat  with xhat = (x-mean)*invstd.  rows = emsa_bn_bwd_rows(n_img*hw, c) is the row
 * count to ALLOCATE (per-workgroup partial rows + the slice-sum rows pass 2 uses as scratch)   */
int emsa_bn_bwd_reduce(const float* dy, const float* y, const uint64_t* mask_bits, const float* x,
                       const float* save_mean, const float* save_invstd, const float* drop,
                       int32_t n_img, int64_t hw, int32_t c, int32_t act, float* partial,
                       void* stream);
int emsa_bn_bwd_rows(int64_t pixels, int32_t c);
/* backward, pass 2: reduces `partial` in two levels (its trailing rows are scratch, so it is
 * not const; rows = the emsa_bn_bwd_rows value it was allocated with; dgamma, dbeta written), then
 *   train: dx = gamma*invstd*(g*drop - dbeta/M - xhat*dgamma/M);  eval (train=0): dx = g*drop*scale
 *   dres (may be NULL) = g                                                                  */
int emsa_bn_bwd_apply(const float* dy, const float* y, const uint64_t* mask_bits, const float* x,
                      const float* gamma, const float* save_mean, const float* save_invstd,
                      const float* drop,
                      float* partial, int32_t rows, int32_t n_img, int64_t hw, int32_t c,
                      int32_t act, int32_t train, float* dx, float* dres, float* dgamma,
                      float* dbeta, void* stream);
/* one Dropout2d layer of emsa_dropout2d_mask_batch: masks of layer `layer_id` (probability p, c
 * channels) go to masks[offset ...]                                                               */
typedef struct EmsaDropoutJob {
  int64_t offset;
  int32_t c;
  uint32_t layer_id;
  float p;
  int32_t pad_;
} EmsaDropoutJob;
/* Dropout2d channel mask [n][c]: 0 or 1/(1-p); counter-based hash shared with the oracle */
int emsa_dropout2d_mask(float* mask, int32_t n, int32_t c, float p, uint32_t seed,
                        uint32_t layer_id, void* stream);
/* the same with the seed in DEVICE memory: state = {base seed, training step}, seed of the step =
 * base + 0x632BE5AB * step (what EMSANet._dropout_seed computes on the host); emsa_u32_add bumps the
 * step counter on the stream.  A training step captured in a hipGraph draws fresh masks per replay. */
int emsa_dropout2d_mask_dev(float* mask, int32_t n, int32_t c, float p, const uint32_t* state,
                            uint32_t layer_id, void* stream);
/* all Dropout2d masks of a step in one launch (jobs in device memory); values identical to the
 * per-layer entry points                                                                          */
int emsa_dropout2d_mask_batch(float* masks, const EmsaDropoutJob* jobs_device, int32_t n_jobs,
                              int32_t n, int32_t max_c, uint32_t seed, const uint32_t* state,
                              void* stream);
int emsa_u32_add(uint32_t* counter, uint32_t value, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pooling / squeeze-and-excitation fusion / upsampling
 * ------------------------------------------------------------------------------------------ */
/* 3x3 stride-2 pad-1 max pool of the ResNet stem; idx = int8 argmax tap per output element */
int emsa_maxpool3x3s2_fwd(const float* x, float* y, int8_t* idx, int32_t n, int32_t h, int32_t w,
                          int32_t c, void* stream);
int emsa_maxpool3x3s2_bwd(const float* dy, const int8_t* idx, float* dx, int32_t n, int32_t h,
                          int32_t w, int32_t c, void* stream);

/* SE-add fusion ('se-add-uni-rgb', emsanet/args.py:143-147):
 *   emsa_channel_mean: gap[n][c] = mean_hw x                      (F.adaptive_avg_pool2d(x,1))
 *   emsa_se_mlp_fwd:   hid = relu(W1 gap + b1); s = sigmoid(W2 hid + b2)   (W1 [cr][c], W2 [c][cr])
 *   emsa_se_scale_add_fwd: out = a*sa[n][c] + b*sb[n][c]          (b/sb NULL: plain SE)        */
/* `ws` = caller-provided scratch of emsa_channel_ws_floats(n, hw, c) floats (two-stage,
 * atomics-free, bit-reproducible reduction) */
int emsa_channel_ws_floats(int32_t n, int64_t hw, int32_t c);
int emsa_channel_mean(const float* x, float* gap, float* ws, int32_t n, int64_t hw, int32_t c,
                      void* stream);
int emsa_se_mlp_fwd(const float* gap, const float* w1, const float* b1, const float* w2,
                    const float* b2, float* hid, float* s, int32_t n, int32_t c, int32_t cr,
                    void* stream);
int emsa_se_mlp_bwd(const float* gap, const float* w1, const float* w2, const float* hid,
                    const float* s, const float* ds, float* dgap, float* dw1, float* db1,
                    float* dw2, float* db2, int32_t n, int32_t c, int32_t cr, void* stream);
int emsa_se_scale_add_fwd(const float* a, const float* sa, const float* b, const float* sb,
                          float* out, int32_t n, int64_t hw, int32_t c, void* stream);
/* ds[n][c] = sum_hw dout*x   (gradient w.r.t. the SE weighting) */
int emsa_se_scale_bwd_reduce(const float* dout, const float* x, float* ds, float* ws, int32_t n,
                             int64_t hw, int32_t c, void* stream);
/* dx = dout*s[n][c] + dgap[n][c]/hw  (+ dx_extra if not NULL: gradient arriving from another
 * consumer of x, e.g. the depth stream that continues un-fused)                             */
int emsa_se_scale_bwd_apply(const float* dout, const float* s, const float* dgap,
                            const float* dx_extra, float* dx, int32_t n, int64_t hw, int32_t c,
                            void* stream);

/* 'learned-3x3-zeropad' upsampling (emsanet/args.py:290-298): nearest x2 then depth-wise 3x3,
 * zero padding, + bias, + optional skip tensor (encoder-decoder fusion 'add-rgb').
 *   x [n][h][w][c] -> y [n][2h][2w][c];  wdw [c][3][3] (OIHW of a depth-wise conv), bias [c]  */
int emsa_up2x_dw3x3_fwd(const float* x, const float* wdw, const float* bias, const float* skip,
                        float* y, int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
int emsa_up2x_dw3x3_bwd_data(const float* dy, const float* wdw, float* dx, int32_t n, int32_t h,
                             int32_t w, int32_t c, void* stream);
/* dw [c][9] and db [c] accumulated with atomics: zero them first */
int emsa_up2x_dw3x3_bwd_weight(const float* dy, const float* x, float* dw, float* db, int32_t n,
                               int32_t h, int32_t w, int32_t c, void* stream);
/* The same backward in ONE pass over dy (LDS-tiled; dx may be NULL): the path the engine takes
 * when emsa_up2x_dw3x3_bwd_supported(c, element size of the features) says so -- c in {8, 16, 32,
 * 40} or a multiple of 64.  dw / db are accumulated with atomics: zero them first. */
int emsa_up2x_dw3x3_bwd(const float* dy, const float* x, const float* wdw, float* dx, float* dw,
                        float* db, int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
int emsa_up2x_dw3x3_bwd_supported(int32_t c, int32_t esize);

/* pyramid pooling ('ppm', emsanet/args.py:243-256): adaptive average pool to bins x bins and
 * bilinear (align_corners=False) upsampling back, written into a channel slice (ld_y).
 * emsa_bilinear_bwd WRITES every element of dx (gather form, fixed summation order: reproducible;
 * no zero-fill needed).                                                                       */
int emsa_adaptive_avgpool_fwd(const float* x, float* y, int32_t n, int32_t h, int32_t w,
                              int32_t c, int32_t bins, void* stream);
int emsa_adaptive_avgpool_bwd(const float* dy, float* dx, int32_t n, int32_t h, int32_t w,
                              int32_t c, int32_t bins, int32_t accumulate, void* stream);
int emsa_bilinear_fwd(const float* x, float* y, int32_t n, int32_t ih, int32_t iw, int32_t oh,
                      int32_t ow, int32_t c, int32_t ld_y, void* stream);
int emsa_bilinear_bwd(const float* dy, float* dx, int32_t n, int32_t ih, int32_t iw, int32_t oh,
                      int32_t ow, int32_t c, int32_t ld_dy, void* stream);

/* instance head activations (sigmoid centre, tanh offset; emsanet/model.py:122-137):
 * channels [0,n_sig) sigmoid, [n_sig, n_sig+n_tanh) tanh, channels [norm_off, norm_off+n_norm)
 * (n_norm 0 or 2, behind the sigmoid/tanh channels) L2-normalised over the pair (orientation biternion; [U] switch ORIENTATION_L2_NORMALIZE of the
 * oracle's Spec, F.normalize(dim=1, eps=1e-12) semantics), rest identity.  bwd needs the forward
 * input x only when n_norm > 0.                                                                */
int emsa_head_act_fwd(const float* x, float* y, int64_t pixels, int32_t c, int32_t n_sig,
                      int32_t n_tanh, int32_t norm_off, int32_t n_norm, void* stream);
int emsa_head_act_bwd(const float* dy, const float* y, const float* x, float* dx, int64_t pixels,
                      int32_t c, int32_t n_sig, int32_t n_tanh, int32_t norm_off, int32_t n_norm,
                      void* stream);

/* plain copies with strides (channel slice <-> dense), and y += x */
int emsa_copy_channels(const float* x, int32_t ld_x, float* y, int32_t ld_y, int64_t pixels,
                       int32_t c, void* stream);
int emsa_axpy(const float* x, float* y, int64_t n, float alpha, void* stream);

/* ------------------------------------------------------------------------------------------
 * Class-weighted cross-entropy with optional label smoothing (SURVEY.md 8f-1;
 * `task_helper.training_step`, main.py:131-141): the semantic head and its side outputs
 * (numerics pinned by emsanet/tests/test_semantic_loss.py:15-48) and -- with 1x1 "images" --
 * the scene head (label smoothing 0.1, args.py:790-796).  target 0 = void (ignored):
 *   l_p  = (1-eps) * w[t] * -log p_p[t]  +  eps/C * sum_c w[c] * -log p_p[c]      (t = t_p - 1)
 *   loss = sum_p l_p / sum_p w[t_p-1]         (torch.nn.CrossEntropyLoss semantics for eps, w)
 * logits NHWC with pixel stride ld (>= n_classes rounded up to 4), target int64 [pixels].
 *   weights_sum  sum_c w[c] (only read when label_smoothing != 0)
 *   partial  float[2 * emsa_ce_semantic_blocks(pixels)] scratch
 *   out      float[2]: out[0] = loss, out[1] = divisor (kept for backward)
 *   backward: dlogits = grad_out[0]/divisor * d l_p / d x, 0 for void pixels and for padding
 *             channels; grad_out is a DEVICE scalar (no host sync)
 * ------------------------------------------------------------------------------------------ */
int emsa_ce_semantic_blocks(int64_t pixels);
int emsa_ce_semantic_fwd(const float* logits, int32_t ld, const int64_t* target,
                         const float* weights, int32_t n_classes, int64_t pixels,
                         float label_smoothing, float weights_sum, float* partial, float* out,
                         void* stream);
int emsa_ce_semantic_bwd(const float* logits, int32_t ld, const int64_t* target,
                         const float* weights, int32_t n_classes, int64_t pixels,
                         float label_smoothing, float weights_sum, const float* sums,
                         const float* grad_out, float* dlogits, int32_t ld_d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Instance-decoder losses in one fused pass (defaults args.py:739-770): MSE on
 * the centre heatmap, L1 on the offsets inside the instance foreground, von-Mises loss
 * 1 - exp(kappa (cos(dtheta) - 1)) on the (sin, cos) orientation pair inside the foreground with
 * an orientation label.  The loss classes are in the un-vendored nicr_mt_scene_analysis library:
 * restated from the published definitions (oracle/instance_loss_oracle.py, parity unpinned).
 *   center/offset/orient  predictions, pixel strides ld_* (orient may be NULL)
 *   center_gt [pixels], offset_gt [pixels][2], orient_gt [pixels] (angle, rad)
 *   center_mask (NULL = all pixels), fg, fg_orient: uint8 [pixels]
 *   partial  float[6 * emsa_instance_loss_blocks(pixels)] scratch
 *   out      float[6]: losses (centre, offset, orientation), then their divisors
 *   backward: grad_out = DEVICE float[3]; d_* written with pixel strides ldd_*
 * ------------------------------------------------------------------------------------------ */
int emsa_instance_loss_blocks(int64_t pixels);
int emsa_instance_loss_fwd(const float* center, int32_t ld_c, const float* offset, int32_t ld_o,
                           const float* orient, int32_t ld_r, const float* center_gt,
                           const float* offset_gt, const float* orient_gt,
                           const uint8_t* center_mask, const uint8_t* fg,
                           const uint8_t* fg_orient, int64_t pixels, float kappa, float* partial,
                           float* out, void* stream);
int emsa_instance_loss_bwd(const float* center, int32_t ld_c, const float* offset, int32_t ld_o,
                           const float* orient, int32_t ld_r, const float* center_gt,
                           const float* offset_gt, const float* orient_gt,
                           const uint8_t* center_mask, const uint8_t* fg,
                           const uint8_t* fg_orient, int64_t pixels, float kappa,
                           const float* sums, const float* grad_out, float* d_center,
                           int32_t ldd_c, float* d_offset, int32_t ldd_o, float* d_orient,
                           int32_t ldd_r, void* stream);

/* ------------------------------------------------------------------------------------------
 * Eval post-processing and input normalisation (csrc/postproc.hip).
 *   emsa_softmax_argmax     per pixel: idx = argmax_c logits, score = softmax(logits)[idx]
 *                           (`semantic_segmentation_idx|score`, `scene_class_idx|score`)
 *   emsa_instance_centers   heat >= threshold  ->  k x k max-pool NMS  ->  top_k by score
 *                           (args.py:468-504); centers float[n][top_k][2] = (y, x) or -1,
 *                           scores float[n][top_k], n_centers int[n].  The top-k is EXACT over all
 *                           NMS survivors (ref decoder.py:95-104), however many a saturated
 *                           heat-map has: per-chunk top-k, then a merge in rounds; top_k <=
 *                           emsa_center_candidates_max() / 2.  ws_* = scratch: ws_count int[2n]
 *                           ([n, 2n) returns the number of NMS survivors per image), ws_score
 *                           float[n * emsa_center_ws_entries(h, w, top_k)], ws_pos int32 alike;
 *                           fg (may be NULL): uint8 foreground applied to the heatmap first
 *   emsa_instance_assign    ids[p] = 1 + argmin_k |(y + off_y*scale_y, x + off_x*scale_x) - c_k|,
 *                           0 outside fg / without centres / beyond max_distance (<= 0: off)
 *   emsa_normalize_rgb      uint8 [n][h][w][3] -> float [n][3][h][w]: (v*scale - mean)/std
 *   emsa_normalize_depth    uint16 -> float: (v - mean)/std, 0 kept 0 when keep_zero
 * ------------------------------------------------------------------------------------------ */
int emsa_softmax_argmax(const float* logits, int32_t ld, int32_t n_classes, int64_t pixels,
                        float* score, int64_t* idx, void* stream);
int emsa_center_candidates_max(void);
int64_t emsa_center_ws_entries(int32_t h, int32_t w, int32_t top_k);
int emsa_instance_centers(const float* heat, int32_t ld, int32_t n, int32_t h, int32_t w,
                          int32_t nms_kernel, float threshold, int32_t top_k, const uint8_t* fg,
                          int32_t* ws_count, float* ws_score, int32_t* ws_pos, float* centers,
                          float* scores, int32_t* n_centers, void* stream);
int emsa_instance_assign(const float* offset, int32_t ld, int32_t n, int32_t h, int32_t w,
                         float scale_y, float scale_x, const float* centers,
                         const int32_t* n_centers, int32_t top_k, const uint8_t* fg,
                         float max_distance, int32_t* ids, void* stream);
/* panoptic merge: every instance (ids from emsa_instance_assign on the thing pixels) takes the
 * majority semantic class of its pixels; stuff pixels keep their class; thing pixels without an
 * instance become void.  pan_semantic is in the label list WITH void (0 = void, class c -> c + 1:
 * /root/reference/inference_dataset.py:298-304, emsanet/visualization.py:726-727; until round 6
 * it was c with -1 = void), pan_instance, pan_id = pan_semantic * label_divisor + instance
 * (0 = void; emsanet/tests/test_metrics_with_model.py:113-131).  class_is_thing uint8[n_classes]
 * (DEVICE); ws_votes int32[n*(top_k+1)*n_classes] scratch; ws_class int32[n*(top_k+1)]: on return
 * the class (WITHOUT void, -1 = no pixel) of every instance slot.                               */
int emsa_panoptic_merge(const int64_t* semantic_idx, const int32_t* instance_ids,
                        const uint8_t* class_is_thing, int32_t n, int64_t hw, int32_t n_classes,
                        int32_t top_k, int32_t label_divisor, int32_t* ws_votes,
                        int32_t* ws_class, int64_t* pan_semantic, int32_t* pan_instance,
                        int64_t* pan_id, void* stream);
/* per-instance pixel count and (value != NULL) fixed-point sum of a [0, 1] score map:
 * area[n][slots] = #pixels with that id (0 < id < slots, mask NULL or != 0), sum[n][slots] =
 * sum of floor(value * 2^30 + 0.5) -- integer atomics: independent of the pixel order.           */
int emsa_instance_stats(const int32_t* ids, const float* value, const uint8_t* mask, int32_t n,
                        int64_t hw, int32_t slots, int64_t* sum, int32_t* area, void* stream);
/* the score maps / per-instance scores the reference's panoptic post-processing is built with
 * (`compute_scores=True`, /root/reference/emsanet/decoder.py:152; consumers:
 * inference_dataset.py:420-437,486-523): per instance the mean semantic score of its pixels and
 * "score_instance_center * (mean_semantic_score_of_instance)" (inference_dataset.py:506-507); per
 * pixel (semantic, instance, panoptic) score = those of its instance; stuff pixels: (own semantic
 * score, 0, own semantic score); void: 0.  pan_instance / pan_semantic from emsa_panoptic_merge,
 * center_scores float[n][top_k] from emsa_instance_centers; ws_sum int64[n*(top_k+1)] scratch;
 * inst_* [n][top_k+1].                                                                         */
int emsa_panoptic_scores(const float* semantic_score, const int32_t* pan_instance,
                         const int64_t* pan_semantic, const float* center_scores, int32_t n,
                         int64_t hw, int32_t top_k, int64_t* ws_sum, int32_t* inst_area,
                         float* inst_semantic_score, float* inst_panoptic_score,
                         float* semantic_score_out, float* instance_score_out,
                         float* panoptic_score_out, void* stream);
/* per-instance sum of the two orientation channels (pixel-major, row stride ld) in fixed point
 * (round(v * 2^24), |v| clamped to 2^14): vec_sum int64[n][slots][2], count int32[n][slots]; the
 * instance's angle is atan2 of the two sums (the `orientations_*` dictionaries of
 * /root/reference/emsanet/visualization.py:752-813).                                            */
int emsa_instance_orientation(const float* orientation, int32_t ld, const int32_t* ids,
                              const uint8_t* mask, int32_t n, int64_t hw, int32_t slots,
                              int64_t* vec_sum, int32_t* count, void* stream);
int emsa_normalize_rgb(const uint8_t* rgb_hwc, float* out_chw, int32_t n, int32_t h, int32_t w,
                       float scale, const float* mean3, const float* std3, void* stream);
int emsa_normalize_depth(const uint16_t* depth, float* out, int64_t total, float mean, float std,
                         int32_t keep_zero, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer step over one flat bucket: torch.optim.SGD(momentum, weight_decay,
 * nesterov=True) as the reference configures it (emsanet/optimizer.py:29-36), one pass:
 *   d = grad * grad_scale + weight_decay * p;  buf = first_step ? d : momentum * buf + d;
 *   p -= lr * (d + momentum * buf)
 * grad_scale folds the 1/world_size averaging of the all-reduced bucket into the update.
 * ------------------------------------------------------------------------------------------ */
int emsa_sgd_nesterov(float* param, const float* grad, float* momentum_buf, int64_t n, float lr,
                      float momentum, float weight_decay, float grad_scale, int32_t first_step,
                      void* stream);
/* the same with {lr, momentum, weight_decay, grad_scale, first_step} read from DEVICE memory at run
 * time (hipGraph-captured training step: the one-cycle schedule changes them between replays) */
int emsa_sgd_nesterov_dev(float* param, const float* grad, float* momentum_buf, int64_t n,
                          const float* hyper, void* stream);
/* Adam / AdamW / RAdam step over one flat bucket -- the other optimizers of the reference's factory
 * (/root/reference/emsanet/optimizer.py:37-57), arithmetic of torch.optim's single-tensor paths.
 * hyper double[8] (DEVICE, written by the host): {lr, beta1, beta2, eps, weight_decay, grad_scale,
 * mode: 0 adam (L2 decay), 1 adamw (decoupled), 2 radam (L2)}; state double[8] (DEVICE, owned by the
 * kernels): {step t, lr/bc1, bc1, sqrt(bc2), RAdam rectification, rho_t > 5}.  emsa_adam_advance:
 * t += 1 and the step-dependent scalars, once per optimizer step in front of the bucket launches
 * (a step captured in a hipGraph counts its replays).                                          */
int emsa_adam_advance(const double* hyper, double* state, void* stream);
int emsa_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                   const double* hyper, const double* state, void* stream);

/* ------------------------------------------------------------------------------------------
 * 16-bit convolution family (BASELINE configs[2] bf16 mixed-precision training, configs[4] 16-bit
 * inference; the reference trains / times fp32 and fp16-TensorRT, README.md:176-181).
 *
 * emsa_conv_igemm_t: forward conv / data gradient on v_mfma_f32_32x32x16_{bf16,f16} with fp32
 *   accumulation; `in`, `w`, `out`, `residual`, `mask_src` are tensors of `dtype`, `w` is the packed
 *   [tap][n_ch][k_ch] operand in `dtype` (emsa_pack_weight_t / emsa_pack_batch kinds 2, 3),
 *   bias / scale / shift and the BatchNorm statistics partials (taken from the fp32 accumulators)
 *   are fp32.  k_ch, n_ch, ld_* must be multiples of 8 (16-byte accesses).  EMSA_DT_F32 forwards
 *   to emsa_conv_igemm.  emsa_conv_stats_rows_t: statistics rows of that launch.
 * emsa_conv_wgrad_t: weight (+bias) gradient from `dtype` activations (bf16) into fp32 gradients;
 *   same workspace contract as emsa_conv_wgrad, sized by emsa_conv_wgrad_ws_bytes_t (the
 *   split-K plan of the 16-bit kernel differs: 64-pixel K steps of the bf16 MFMA).
 * emsa_pack_weight_t / emsa_stem_pack_weight_t: fp32 OIHW parameter -> 16-bit packed operands.
 * ------------------------------------------------------------------------------------------ */
int emsa_conv_stats_rows_t(int32_t dtype, const EmsaConvGeom* g);
int emsa_conv_igemm_t(int32_t dtype, const EmsaConvGeom* g, const void* in, const void* w,
                      void* out, const float* bias, float* stats, const float* scale,
                      const float* shift, const void* residual, int32_t ld_res,
                      const void* mask_src, int32_t ld_mask, int32_t act, void* stream);
/* Tap-split forward conv for maps with few output tiles (batch-1 inference, BASELINE configs[4]: the
 * decoders' 3x3 convs at /32 and /16 are 40-76 tiles of 72 K steps on 256 CUs): `ksplit` workgroups
 * per tile, each over a range of taps, raw fp32 partial sums in `ws`, then one pass that sums them in
 * a fixed order and applies the epilogue of emsa_conv_igemm_t (bias, folded BatchNorm, residual,
 * ReLU; no statistics, no mask).  emsa_conv_igemm_splitk_ws_bytes_t: bytes of `ws`, 0 = use
 * emsa_conv_igemm_t (enough tiles, short K, fp32, mapped output).  EMSA_CONVH_SPLITK=0 switches it off. */
int64_t emsa_conv_igemm_splitk_ws_bytes_t(int32_t dtype, const EmsaConvGeom* g);
int emsa_conv_igemm_splitk_t(int32_t dtype, const EmsaConvGeom* g, const void* in, const void* w,
                             void* out, const float* bias, const float* scale, const float* shift,
                             const void* residual, int32_t ld_res, int32_t act, float* ws,
                             void* stream);
/* Twin launch of emsa_conv_igemm_t / emsa_conv_igemm_splitk_t (16-bit storage): ONE launch (grid.y = 2)
 * runs the same geometry on two independent sets of tensors -- the strided / 1x1 convs of the
 * rgb | depth encoder blocks and the 3x3 / skip-fusion convs of the semantic | instance decoder modules
 * (/root/reference/emsanet/model.py:95-160, decoder.py:63-139); in0 may equal in1.  Forward epilogue
 * only; each half == its own launch bit for bit.  ws0 / ws1: emsa_conv_igemm_splitk_ws_bytes_t(dtype, g)
 * bytes each where that is > 0 (tap-split form, one finish launch for both halves), else NULL.
 * EMSA_E_SHAPE: not a 16-bit launch the twin form exists for (the caller launches twice). */
int emsa_conv_igemm_pair_t(int32_t dtype, const EmsaConvGeom* g, const void* in0, const void* in1,
                           const void* w0, const void* w1, void* out0, void* out1,
                           const float* bias0, const float* bias1, const float* scale0,
                           const float* scale1, const float* shift0, const float* shift1,
                           const void* residual0, const void* residual1, int32_t ld_res,
                           int32_t act, float* ws0, float* ws1, void* stream);
/* 16-bit twin of emsa_conv1d_wino_bnb (tiles: emsa_conv_stats_rows_t(dtype, g)) and the BatchNorm
 * backward that consumes its output: g = the masked gradient stored by the conv, partial =
 * float[2][rows + 16][c] with rows [0, rows) filled; dtype EMSA_DT_F32 pairs with the Winograd
 * entry point above.                                                                             */
int emsa_conv_igemm_bnb_t(int32_t dtype, const EmsaConvGeom* g, const void* dy, const void* w,
                          void* out, const void* residual, int32_t ld_res, const void* t,
                          int32_t ld_t, const float* bn_scale, const float* bn_shift,
                          const float* bn_mean, const float* bn_invstd, float* partial,
                          int32_t rows_alloc, void* stream);
int emsa_bn_bwd_apply_rows_t(int32_t dtype, const void* g, const void* x, const float* gamma,
                             const float* save_mean, const float* save_invstd, float* partial,
                             int32_t rows, int32_t n_img, int64_t hw, int32_t c, int32_t train,
                             void* dx, float* dgamma, float* dbeta, void* stream);
int64_t emsa_conv_wgrad_ws_bytes_t(int32_t dtype, const EmsaConvGeom* g);
int emsa_conv_wgrad_t(int32_t dtype, const EmsaConvGeom* g, const void* in, const void* dout,
                      float* dw, float* dbias, float* ws, void* stream);
/* Several independent weight gradients in ONE launch + one reduction launch (bf16 activations):
 * n_jobs = 2..4 stride-1 3-tap 1-D (or 3x3) convs with the same channel counts -- the four convs of
 * an NBt1D block, /root/reference/emsanet/model.py:47-58 -- share one split-K budget (grid.y = job).
 * geoms: array of n_jobs geometries; in / dout / dw / dbias: arrays of n_jobs pointers (dbias[j]
 * may be NULL); ws: emsa_conv_wgrad_multi_ws_bytes bytes (0 = no multi-job form for these jobs: run
 * them through emsa_conv_wgrad_t).  Every job's result is what emsa_conv_wgrad_t's deterministic
 * two-pass form gives up to the fp32 summation order of the splits. */
int64_t emsa_conv_wgrad_multi_ws_bytes(int32_t dtype, int32_t n_jobs, const EmsaConvGeom* geoms);
int emsa_conv_wgrad_multi_t(int32_t dtype, int32_t n_jobs, const EmsaConvGeom* geoms,
                            const void* const* in, const void* const* dout, float* const* dw,
                            float* const* dbias, float* ws, void* stream);
int emsa_pack_weight_t(int32_t dtype, const float* w, void* wp_fwd, void* wp_dgrad, int32_t cout,
                       int32_t cin, int32_t kh, int32_t kw, int32_t cout_total, int32_t cout_off,
                       int32_t cin_total, int32_t cin_off, void* stream);
int emsa_stem_pack_weight_t(int32_t dtype, const float* w, void* wp, int32_t cout, int32_t cin,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * Register-stationary streaming kernel for the stride-1 3-tap 1-D convs of the NBt1D blocks in
 * 16-bit storage (csrc/conv_rs.hip; /root/reference/emsanet/model.py:47-58 composes the block,
 * args.py:158-164 selects it): conv3x1 / conv1x3 with C_in = C_out in {64, 128, 256, 512}, forward
 * and data gradient.  Persistent workgroups keep their slice of the weights in registers as MFMA
 * operands for the whole launch and stream pixel tiles (+ halo) through LDS once; HBM traffic =
 * input + output.  Same arguments and fused epilogue as emsa_conv_igemm_t / emsa_conv_igemm_bnb_t
 * except the weight operand:
 *   wfrag  fragment-ordered 16-bit weights from emsa_pack_weight_frag_t (or emsa_pack_batch kinds
 *          5 = bf16, 6 = fp16): [tap][n / 32][k / 16][n % 32 + 32 * ((k % 16) / 8)][k % 8], i.e. one
 *          B operand of v_mfma_f32_32x32x16 per contiguous KB; forward (n, k) = (cout, cin), data
 *          gradient (n, k) = (cin, cout), taps unflipped (the geometry's step = -1 flips them).
 *   stats / partial rows: emsa_conv1d_rs_stats_rows(dtype, g) (one row per persistent workgroup
 *          of a channel slice; the rows carry (sum, M2, count) like emsa_conv_igemm's).
 * emsa_conv1d_rs_supported: 1 when this kernel takes the geometry (and EMSA_CONV_RS != 0), else 0 --
 * callers fall back to emsa_conv_igemm_t with the [tap][n][k] operand.
 * emsa_conv_rs_set_cu_budget(cus): plan the persistent grid for `cus` CUs instead of all of them
 * (<= 0: all; also EMSA_RS_CUS) so that long-lived co-resident kernels -- RCCL's all-reduce
 * channels of a multi-rank step -- find free CU slots; returns the CU count now in use.  Changes
 * emsa_conv1d_rs_stats_rows: query it again afterwards.
 * ------------------------------------------------------------------------------------------ */
/* Fused NBt1D half-block for small-batch 16-bit inference (csrc/conv_hb.hip; the block is composed at
 * /root/reference/emsanet/model.py:47-58, args.py:158-164):
 *   out = act((conv1x3(relu(conv3x1(in) + bias_a)) + bias_b) * scale + shift + residual)
 * as ONE launch for c = 64 / 128 -- the intermediate row stays in LDS.  n_sets = 1 or 2 independent
 * tensor sets of the same shape (the twin modules: rgb | depth, semantic | instance); every pointer
 * argument is an ARRAY of n_sets pointers (bias_*[k], scale[k] + shift[k], residual[k] may be NULL).
 * wfa / wfb: fragment-ordered forward weights (emsa_pack_weight_frag_t).  Bit-identical to
 * emsa_conv1d_rs_t on the 3x1 conv followed by emsa_conv1d_rs_t on the 1x3 conv. */
int emsa_nbt_half_block_supported(int32_t dtype, int32_t c, int32_t w);
int emsa_nbt_half_block_t(int32_t dtype, int32_t n_sets, int32_t n_img, int32_t h, int32_t w,
                          int32_t c, const void* const* in, int32_t ld_in, const void* const* wfa,
                          const float* const* bias_a, const void* const* wfb,
                          const float* const* bias_b, const float* const* scale,
                          const float* const* shift, const void* const* residual, int32_t ld_res,
                          void* const* out, int32_t ld_out, int32_t act, void* stream);
int emsa_conv_rs_set_cu_budget(int32_t cus);
int emsa_conv1d_rs_supported(int32_t dtype, const EmsaConvGeom* g);
int emsa_conv1d_rs_stats_rows(int32_t dtype, const EmsaConvGeom* g);
int emsa_conv1d_rs_t(int32_t dtype, const EmsaConvGeom* g, const void* in, const void* wfrag,
                     void* out, const float* bias, float* stats, const float* scale,
                     const float* shift, const void* residual, int32_t ld_res,
                     const void* mask_src, int32_t ld_mask, int32_t act, void* stream);
/* Twin launch of the kernel above: ONE launch (grid.y = 2) runs the same geometry on two independent
 * sets of tensors -- the rgb | depth encoder blocks and the semantic | instance decoder blocks of
 * /root/reference/emsanet/model.py:95-160 have identical shapes, and at batch 1 (BASELINE configs[4])
 * a launch is its fixed cost.  Forward epilogue only; each half == its own emsa_conv1d_rs_t launch bit
 * for bit.  bias / scale + shift / residual: for both halves or for neither; one ld_res. */
int emsa_conv1d_rs_pair_t(int32_t dtype, const EmsaConvGeom* g, const void* in0, const void* in1,
                          const void* wfrag0, const void* wfrag1, void* out0, void* out1,
                          const float* bias0, const float* bias1, const float* scale0,
                          const float* scale1, const float* shift0, const float* shift1,
                          const void* residual0, const void* residual1, int32_t ld_res,
                          int32_t act, void* stream);
int emsa_conv1d_rs_bnb_t(int32_t dtype, const EmsaConvGeom* g, const void* dy, const void* wfrag,
                         void* out, const void* residual, int32_t ld_res, const void* t,
                         int32_t ld_t, const float* bn_scale, const float* bn_shift,
                         const float* bn_mean, const float* bn_invstd, float* partial,
                         int32_t rows_alloc, void* stream);
int emsa_pack_weight_frag_t(int32_t dtype, const float* w_oihw, void* wf_fwd, void* wf_dgrad,
                            int32_t cout, int32_t cin, void* stream);

/* ------------------------------------------------------------------------------------------
 * The NBt1D block's bn1 (BatchNorm + ReLU between conv1x3_1 and conv3x1_2, emsanet/model.py:47-58,
 * args.py:125-131) FOLDED into its consumers in 16-bit storage (round 6; the fp32 twins are
 * emsa_conv1d_wino_inbn / emsa_conv_wgrad_inbn / emsa_conv1d_wino_bnb): the normalised tensor and
 * its ReLU bit mask are never written.  in_scale / in_shift / mask_scale / mask_shift = the
 * BatchNorm's folded per-channel scale and shift of THIS forward pass (emsa_bn_finalize).
 *   emsa_conv1d_rs_inbn_t        forward conv3x1 on relu(in * in_scale + in_shift), formed where the
 *                                staged tile enters LDS; padding rows stay zero (the padding pads the
 *                                normalised tensor).  bf16, taps along H; epilogue bias / statistics /
 *                                ReLU.  EMSA_E_SHAPE: no such form for this geometry.
 *   emsa_conv_wgrad_inbn_t       weight gradient of that conv, x operand recomputed in the loader.
 *   emsa_conv_wgrad_multi_inbn_t the multi-job launch with per-job in_scale[j] / in_shift[j] (NULL
 *                                entries = plain jobs).
 *   emsa_bn_bwd_reduce_aff_t /   BatchNorm + ReLU backward whose ReLU decisions are recomputed from the
 *   emsa_bn_bwd_apply_aff_t      BatchNorm's input: (x * mask_scale + mask_shift) > 0, the same single
 *                                fma as the loaders.  Batch statistics, no Dropout2d; partial rows as
 *                                emsa_bn_bwd_reduce_t / emsa_bn_bwd_apply_t.
 * ------------------------------------------------------------------------------------------ */
int emsa_conv1d_rs_inbn_t(int32_t dtype, const EmsaConvGeom* g, const void* in, const void* wfrag,
                          void* out, const float* bias, float* stats, const float* in_scale,
                          const float* in_shift, int32_t act, void* stream);
int emsa_conv_wgrad_inbn_t(int32_t dtype, const EmsaConvGeom* g, const void* in, const void* dout,
                           float* dw, float* dbias, float* ws, const float* in_scale,
                           const float* in_shift, void* stream);
int emsa_conv_wgrad_multi_inbn_t(int32_t dtype, int32_t n_jobs, const EmsaConvGeom* geoms,
                                 const void* const* in, const void* const* dout, float* const* dw,
                                 float* const* dbias, float* ws, const float* const* in_scale,
                                 const float* const* in_shift, void* stream);
int emsa_bn_bwd_reduce_aff_t(int32_t dtype, const void* dy, const void* x, const float* save_mean,
                             const float* save_invstd, const float* mask_scale,
                             const float* mask_shift, int32_t n_img, int64_t hw, int32_t c,
                             float* partial, void* stream);
int emsa_bn_bwd_apply_aff_t(int32_t dtype, const void* dy, const void* x, const float* gamma,
                            const float* save_mean, const float* save_invstd,
                            const float* mask_scale, const float* mask_shift, float* partial,
                            int32_t rows_alloc, int32_t n_img, int64_t hw, int32_t c, void* dx,
                            float* dgamma, float* dbeta, void* stream);

/* ------------------------------------------------------------------------------------------
 * Typed ("_t") forms of the HBM-bound kernels: same semantics and argument order as the fp32 entry
 * points above, activation tensors in the storage type `dtype` (EMSA_DT_*).  Kernels at the model
 * boundary take `out_f32`: the tensors on the OUTPUT side of the op (y of a forward kernel, dy / y
 * of its backward kernels) are fp32 while the features are 16-bit -- the last up-sampling of a head
 * and the head activations write the model's fp32 outputs directly.  emsa_bilinear_bwd_t always
 * writes an fp32 dx (gather form: every element written, reproducible).
 * emsa_cast_channels: strided channel-slice copy with conversion between storage types.
 * ------------------------------------------------------------------------------------------ */
int emsa_bn_act_fwd_t(int32_t dtype, const void* x, void* y, const float* scale, const float*
    shift, const float* drop, const void* residual, int32_t n_img, int64_t hw, int32_t c,
    int32_t act, uint64_t* mask_bits, void* stream);
int emsa_bn_bwd_reduce_t(int32_t dtype, const void* dy, const void* y, const uint64_t*
    mask_bits, const void* x, const float* save_mean, const float* save_invstd, const float*
    drop, int32_t n_img, int64_t hw, int32_t c, int32_t act, float* partial, void* stream);
int emsa_bn_bwd_apply_t(int32_t dtype, const void* dy, const void* y, const uint64_t* mask_bits,
    const void* x, const float* gamma, const float* save_mean, const float* save_invstd, const
    float* drop, float* partial, int32_t rows_alloc, int32_t n_img, int64_t hw, int32_t c,
    int32_t act, int32_t train, void* dx, void* dres, float* dgamma, float* dbeta, void*
    stream);
int emsa_maxpool3x3s2_fwd_t(int32_t dtype, const void* x, void* y, int8_t* idx, int32_t n,
    int32_t h, int32_t w, int32_t c, void* stream);
int emsa_maxpool3x3s2_bwd_t(int32_t dtype, const void* dy, const int8_t* idx, void* dx, int32_t
    n, int32_t h, int32_t w, int32_t c, void* stream);
int emsa_se_scale_add_fwd_t(int32_t dtype, const void* a, const float* sa, const void* b, const
    float* sb, void* out, int32_t n, int64_t hw, int32_t c, void* stream);
int emsa_se_scale_bwd_apply_t(int32_t dtype, const void* dout, const float* s, const float*
    dgap, const void* dx_extra, void* dx, int32_t n, int64_t hw, int32_t c, void* stream);
int emsa_up2x_dw3x3_fwd_t(int32_t dtype, int32_t out_f32, const void* x, const float* wdw, const
    float* bias, const void* skip, void* y, int32_t n, int32_t h, int32_t w, int32_t c, void*
    stream);
/* Twin launch of emsa_up2x_dw3x3_fwd_t (16-bit features in, the same type out): the learned x2
 * up-sampling + skip addition of the semantic | instance decoder modules
 * (/root/reference/emsanet/decoder.py:63-139, args.py:290-298) in ONE launch, grid.y = 2; each half ==
 * its own launch bit for bit.  bias / skip: for both halves or for neither. */
int emsa_up2x_dw3x3_fwd_pair_t(int32_t dtype, const void* x0, const void* x1, const float* wdw0,
                               const float* wdw1, const float* bias0, const float* bias1,
                               const void* skip0, const void* skip1, void* y0, void* y1, int32_t n,
                               int32_t h, int32_t w, int32_t c, void* stream);
int emsa_up2x_dw3x3_bwd_data_t(int32_t dtype, int32_t out_f32, const void* dy, const float* wdw,
    void* dx, int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
int emsa_up2x_dw3x3_bwd_weight_t(int32_t dtype, int32_t out_f32, const void* dy, const void* x,
    float* dw, float* db, int32_t n, int32_t h, int32_t w, int32_t c, void* stream);
int emsa_up2x_dw3x3_bwd_t(int32_t dtype, int32_t out_f32, const void* dy, const void* x, const
    float* wdw, void* dx, float* dw, float* db, int32_t n, int32_t h, int32_t w, int32_t c, void*
    stream);
int emsa_adaptive_avgpool_fwd_t(int32_t dtype, const void* x, void* y, int32_t n, int32_t h,
    int32_t w, int32_t c, int32_t bins, void* stream);
int emsa_adaptive_avgpool_bwd_t(int32_t dtype, const void* dy, void* dx, int32_t n, int32_t h,
    int32_t w, int32_t c, int32_t bins, int32_t accumulate, void* stream);
int emsa_bilinear_fwd_t(int32_t dtype, const void* x, void* y, int32_t n, int32_t ih, int32_t
    iw, int32_t oh, int32_t ow, int32_t c, int32_t ld_y, void* stream);
int emsa_bilinear_bwd_t(int32_t dtype, const void* dy, float* dx, int32_t n, int32_t ih, int32_t
    iw, int32_t oh, int32_t ow, int32_t c, int32_t ld_dy, void* stream);
/* out = a + b over n dense elements of one storage type (dtype: EMSA_DT_*): the encoder skip added
 * behind a plain 'nearest' / 'bilinear' decoder up-sampling (`--{semantic,instance}-decoder-
 * upsampling`, /root/reference/emsanet/args.py:280-298,363-372)                                 */
int emsa_add_t(int32_t dtype, const void* a, const void* b, void* out, int64_t n, void* stream);
/* 'nearest' up-sampling of the pyramid-pooling branches (`--upsampling-context-module nearest`,
 * emsanet/args.py:250-256, passed on at emsanet/model.py:109-119): source index floor(dst * in / out) as
 * torch.nn.functional.interpolate(mode='nearest'); y may be a channel slice (ld_y).  The backward pass
 * writes every element of an fp32 dx (gather, fixed order). */
int emsa_nearest_fwd_t(int32_t dtype, const void* x, void* y, int32_t n, int32_t ih, int32_t iw,
    int32_t oh, int32_t ow, int32_t c, int32_t ld_y, void* stream);
int emsa_nearest_bwd_t(int32_t dtype, const void* dy, float* dx, int32_t n, int32_t ih, int32_t iw,
    int32_t oh, int32_t ow, int32_t c, int32_t ld_dy, void* stream);
int emsa_head_act_fwd_t(int32_t dtype, int32_t out_f32, const void* x, void* y, int64_t pixels,
    int32_t c, int32_t n_sig, int32_t n_tanh, int32_t norm_off, int32_t n_norm, void* stream);
/* emsa_head_act_bwd over the 8-channel (padded) instance head with the three task gradients (centre,
 * offset, orientation: fp32, NHWC, own pixel strides; NULL = zero) gathered on the fly instead of
 * copied into a padded tensor first; y fp32, x / dx in `dtype` */
int emsa_head_act_bwd_gather_t(int32_t dtype, const float* g0, int32_t ld0, int32_t c0, const float* g1,
                               int32_t ld1, int32_t c1, const float* g2, int32_t ld2, int32_t c2,
                               const float* y, const void* x, void* dx, int64_t pixels, int32_t c,
                               int32_t n_sig, int32_t n_tanh, int32_t norm_off, int32_t n_norm,
                               void* stream);
int emsa_head_act_bwd_t(int32_t dtype, int32_t out_f32, const void* dy, const void* y, const
    void* x, void* dx, int64_t pixels, int32_t c, int32_t n_sig, int32_t n_tanh, int32_t
    norm_off, int32_t n_norm, void* stream);
int emsa_stem_pack_input_t(int32_t dtype, const float* x, void* xp, int32_t n, int32_t c,
    int32_t h, int32_t w, void* stream);
/* The squeeze + excitation of BOTH inputs of an SE-add fusion (reference: emsanet/model.py
 * fuse_rgb_depth -> nicr_segmentation SqueezeAndExciteFusionAdd: se_rgb(rgb) + se_depth(depth)) in
 * two launches: channel sums of xa / xb, then both MLPs fed by the split sums.  Results are
 * bit-identical to emsa_channel_mean_t + emsa_se_mlp_fwd per input.  ws: 2 *
 * emsa_channel_ws_floats(n, hw, c) floats; gap / s: [2][n][c], hid: [2][n][cr] (index 0 = xa). */
int emsa_se_pair_fwd_t(int32_t dtype, const void* xa, const void* xb, float* ws, const float* w1a,
                       const float* b1a, const float* w2a, const float* b2a, const float* w1b,
                       const float* b1b, const float* w2b, const float* b2b, float* gap,
                       float* hid, float* s, int32_t n, int64_t hw, int32_t c, int32_t cr,
                       void* stream);
int emsa_channel_mean_t(int32_t dtype, const void* x, float* gap, float* ws, int32_t n, int64_t
    hw, int32_t c, void* stream);
int emsa_se_scale_bwd_reduce_t(int32_t dtype, const void* dout, const void* x, float* ds, float*
    ws, int32_t n, int64_t hw, int32_t c, void* stream);
int emsa_cast_channels(int32_t src_dtype, const void* x, int32_t ld_x, int32_t dst_dtype, void*
    y, int32_t ld_y, int64_t pixels, int32_t c, void* stream);
/* Module boundary (callers own ordinary PyTorch tensors): dense NHWC copy y[n][h][w][c]
 * of a logical (n, c, h, w) tensor given by its ELEMENT strides -- contiguous NCHW as the reference's
 * decoder tests build their inputs (/root/reference/emsanet/tests/test_interface_decoders.py:73-88)
 * and as NCHW losses return their cotangents, or any other view (strides may be 0: expanded
 * scalars).  Plane-contiguous sources take an LDS-tiled transpose, the rest an element-wise gather;
 * one pass, storage type `dtype` on both sides.                                                    */
int emsa_to_nhwc_t(int32_t dtype, const void* x, void* y, int32_t n, int32_t c, int32_t h,
                   int32_t w, int64_t s_n, int64_t s_c, int64_t s_h, int64_t s_w, void* stream);

/* ------------------------------------------------------------------------------------------
 * hipGraph surgery (emsanet_amd.graph: whole-step capture; new capability, the reference has no
 * graph path -- its low-latency route is ONNX -> TensorRT, inference_time_whole_model.py:350-453).
 * `graph` is a hipGraph_t obtained by stream capture and NOT yet instantiated.
 * emsa_graph_replace_memsets rewrites every memset node (torch's zero-fills of reduction scratch
 * and gradient buffers) as a fill KERNEL node with the same edges: captured memset nodes were
 * found to corrupt replays when eager memsets run in between (ROCm 7.2).
 * ------------------------------------------------------------------------------------------ */
int emsa_memset_async(void* dst, int32_t value, int64_t bytes, void* stream);
int emsa_graph_count_nodes(void* graph, int32_t* n_nodes, int32_t* n_memset, int32_t* n_kernel);
int emsa_graph_replace_memsets(void* graph, int32_t* replaced);

/* ------------------------------------------------------------------------------------------
 * Per-launch timing of the MFMA conv kernels (bench.py roofline): when enabled, every n-th
 * emsa_conv_igemm / emsa_conv_wgrad launch of each class is bracketed by HIP events on its own
 * stream (bracketing EVERY launch costs ~3 % of a training step, every 4th < 1 %).
 * cls 0..3 = conv_igemm_kernel tile configs, 4..7 = conv_wgrad_kernel configs
 * (emsa_prof_name).  emsa_prof_read: call after synchronising; sums since emsa_prof_reset;
 * total_flops = ALGORITHMIC direct-convolution FLOPs (2*pixels*k_ch*n_ch*taps).
 * ------------------------------------------------------------------------------------------ */
int emsa_prof_enable(int32_t every);   /* 0 = off; n = bracket every n-th launch per class */
/* one-shot: algorithmic FLOPs of the NEXT conv launch (convs whose GEMM runs on zero-padded
 * channel counts report their real direct-convolution FLOPs: stem 7x7x3, block-diagonal heads) */
int emsa_prof_next_flops(double flops);
int emsa_prof_reset(void);
int emsa_prof_seen(int32_t cls);       /* launches of the class since reset (sampled or not) */
const char* emsa_prof_name(int32_t cls);
int emsa_prof_read(int32_t cls, double* total_ms, double* total_flops, int32_t* launches);
/* algorithmic HBM bytes of the sampled launches (tracked by the 16-bit kernel classes) */
int emsa_prof_read_bytes(int32_t cls, double* total_bytes);

#ifdef __cplusplus
}
#endif
#endif /* EMSANET_HIP_H */
