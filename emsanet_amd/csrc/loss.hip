// Class-weighted semantic cross-entropy on device ("next" row f-1 of SURVEY.md §8): what
// `task_helper.training_step` does for the semantic head (/root/reference/main.py:131-141),
// numerics pinned by the reference's in-tree oracle class CrossEntropyLossPrevious
// (/root/reference/emsanet/tests/test_semantic_loss.py:15-48):
//     loss = sum_p w[t_p-1] * -log softmax(x_p)[t_p-1]  /  sum_p w[t_p-1]     (t_p = 0: void)
// HBM-bound: the logits (bs=32: 1.57 GB at full resolution) are read once in forward and once in
// backward, the gradient written once.  NHWC logits: one pixel = C contiguous floats; a block
// stages 256 pixels through LDS (coalesced float4) and one thread owns one pixel.
#include "common.h"

namespace {

constexpr int kPix = 256;     // pixels per block

template <int C4>
__device__ __forceinline__ void load_tile(const float* __restrict__ x, int ld, long p0, long pixels,
                                          float* tile) {
  constexpr int LD = C4 * 4 + 4;     // row stride 4k+4 floats: conflict-free ds_read_b128
#pragma unroll
  for (int j = 0; j < C4; ++j) {
    const int idx = threadIdx.x + kPix * j;
    const int px = idx / C4, c4 = idx % C4;
    float4 v = emsa_zero4();
    // chunks at or beyond the row's pixel stride belong to the next pixel: never touched
    if (p0 + px < pixels && c4 * 4 < ld) v = emsa_ld4(x + (p0 + px) * (long)ld + c4 * 4);
    emsa_st4(tile + px * LD + c4 * 4, v);
  }
}

template <int C4>
__global__ __launch_bounds__(kPix) void ce_fwd_kernel(const float* __restrict__ logits, int ld,
                                                      const int64_t* __restrict__ target,
                                                      const float* __restrict__ w, int n_classes,
                                                      long pixels, float eps, float w_sum,
                                                      float* __restrict__ partial) {
  constexpr int LD = C4 * 4 + 4;
  extern __shared__ __attribute__((aligned(16))) float tile[];
  __shared__ float red[2][kPix / 64];
  const long p0 = (long)blockIdx.x * kPix;
  load_tile<C4>(logits, ld, p0, pixels, tile);
  __syncthreads();
  const long p = p0 + threadIdx.x;
  float loss = 0.f, wt = 0.f;
  if (p < pixels) {
    const long t = target[p] - 1;          // 0 = void -> ignored
    if (t >= 0 && t < n_classes) {
      // the pixel's logits in registers: C4 conflict-free ds_read_b128 (row stride 4k+4 floats)
      float v[C4 * 4];
#pragma unroll
      for (int j = 0; j < C4; ++j)
        *reinterpret_cast<float4*>(v + 4 * j) = emsa_ld4(tile + threadIdx.x * LD + 4 * j);
      float m = -INFINITY, xt = 0.f;
#pragma unroll
      for (int c = 0; c < C4 * 4; ++c)
        if (c < n_classes) {
          m = fmaxf(m, v[c]);
          xt = c == t ? v[c] : xt;
        }
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < C4 * 4; ++c)
        if (c < n_classes) s += expf(v[c] - m);
      wt = w[t];
      const float lse = logf(s) + m;
      loss = (1.f - eps) * wt * (lse - xt);
      if (eps != 0.f) {
        // label smoothing, torch semantics: + eps/C * sum_c w_c * -log p_c
        float wx = 0.f;
#pragma unroll
        for (int c = 0; c < C4 * 4; ++c)
          if (c < n_classes) wx += w[c] * v[c];
        loss += eps / (float)n_classes * (lse * w_sum - wx);
      }
    }
  }
  // deterministic block reduction: wave shuffles, then 4 waves through LDS
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    loss += __shfl_down(loss, o);
    wt += __shfl_down(wt, o);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = loss;
    red[1][threadIdx.x >> 6] = wt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < kPix / 64; ++k) { a += red[0][k]; b += red[1][k]; }
    partial[blockIdx.x] = a;
    partial[gridDim.x + blockIdx.x] = b;
  }
}

// out[0] = loss, out[1] = sum of weights (divisor); fixed summation order, fp64
__global__ void ce_finalize_kernel(const float* __restrict__ partial, int n, float* __restrict__ out) {
  __shared__ double sa[256], sb[256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) { a += (double)partial[i]; b += (double)partial[n + i]; }
  sa[threadIdx.x] = a; sb[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sa[threadIdx.x] += sa[threadIdx.x + o]; sb[threadIdx.x] += sb[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sb[0] > 0.0 ? (float)(sa[0] / sb[0]) : 0.f;
    out[1] = (float)sb[0];
  }
}

template <int C4>
__global__ __launch_bounds__(kPix) void ce_bwd_kernel(const float* __restrict__ logits, int ld,
                                                      const int64_t* __restrict__ target,
                                                      const float* __restrict__ w, int n_classes,
                                                      long pixels, float eps, float w_sum,
                                                      const float* __restrict__ sums,
                                                      const float* __restrict__ gout,
                                                      float* __restrict__ dlogits, int ld_d) {
  constexpr int LD = C4 * 4 + 4;
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const long p0 = (long)blockIdx.x * kPix;
  load_tile<C4>(logits, ld, p0, pixels, tile);
  __syncthreads();
  const long p = p0 + threadIdx.x;
  float* row = tile + threadIdx.x * LD;
  const float div = sums[1];
  const float g = div > 0.f ? gout[0] / div : 0.f;
  long t = -1;
  if (p < pixels) {
    t = target[p] - 1;
    if (t >= n_classes) t = -1;
  }
  float v[C4 * 4];
  if (t >= 0) {
#pragma unroll
    for (int j = 0; j < C4; ++j) *reinterpret_cast<float4*>(v + 4 * j) = emsa_ld4(row + 4 * j);
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < C4 * 4; ++c)
      if (c < n_classes) m = fmaxf(m, v[c]);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C4 * 4; ++c) {
      v[c] = c < n_classes ? expf(v[c] - m) : 0.f;
      s += v[c];
    }
    const float k = g * w[t] * (1.f - eps), inv = 1.f / s;
    const float ks = g * eps / (float)n_classes;     // label-smoothing term (0 by default)
#pragma unroll
    for (int c = 0; c < C4 * 4; ++c) {
      const float pc = v[c] * inv;
      float d = k * (pc - (c == t ? 1.f : 0.f));
      if (eps != 0.f && c < n_classes) d += ks * (w_sum * pc - w[c]);
      v[c] = c < n_classes ? d : 0.f;
    }
  } else {
#pragma unroll
    for (int c = 0; c < C4 * 4; ++c) v[c] = 0.f;
  }
#pragma unroll
  for (int j = 0; j < C4; ++j) emsa_st4(row + 4 * j, *reinterpret_cast<float4*>(v + 4 * j));
  __syncthreads();
#pragma unroll
  for (int j = 0; j < C4; ++j) {
    const int idx = threadIdx.x + kPix * j;
    const int px = idx / C4, c4 = idx % C4;
    if (p0 + px < pixels && c4 * 4 < ld_d)
      emsa_st4(dlogits + (p0 + px) * (long)ld_d + c4 * 4, emsa_ld4(tile + px * LD + c4 * 4));
  }
}

template <int C4>
int launch_fwd(const float* x, int ld, const int64_t* t, const float* w, int nc, long pixels,
               float eps, float w_sum, float* partial, float* out, hipStream_t st) {
  const int grid = (int)((pixels + kPix - 1) / kPix);
  const size_t lds = (size_t)kPix * (C4 * 4 + 4) * sizeof(float);
  hipLaunchKernelGGL(ce_fwd_kernel<C4>, dim3(grid), dim3(kPix), lds, st, x, ld, t, w, nc, pixels,
                     eps, w_sum, partial);
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, partial, grid, out);
  return emsa_launch_status();
}
template <int C4>
int launch_bwd(const float* x, int ld, const int64_t* t, const float* w, int nc, long pixels,
               float eps, float w_sum, const float* sums, const float* gout, float* dx, int ld_d,
               hipStream_t st) {
  const int grid = (int)((pixels + kPix - 1) / kPix);
  const size_t lds = (size_t)kPix * (C4 * 4 + 4) * sizeof(float);
  hipLaunchKernelGGL(ce_bwd_kernel<C4>, dim3(grid), dim3(kPix), lds, st, x, ld, t, w, nc, pixels,
                     eps, w_sum, sums, gout, dx, ld_d);
  return emsa_launch_status();
}


// ------------------------------------------------------------------------------------------
// Instance-decoder losses, one fused pass over the five prediction channels (SURVEY.md 8f-1:
// MSE centre / L1 offset with foreground masks, von-Mises orientation loss, defaults
// /root/reference/emsanet/args.py:739-770).  The loss classes live in the un-vendored
// nicr_mt_scene_analysis library -> restated from the published definitions (oracle/
// instance_loss_oracle.py, PARITY UNPINNED):
//   centre  = sum_p m_c (c_p - c*_p)^2 / sum m_c                      (m_c = 1 without a mask)
//   offset  = sum_{p in fg} (|dy_p - dy*_p| + |dx_p - dx*_p|) / (2 n_fg)
//   orient. = sum_{p in fgo} (1 - exp(kappa (cos(theta_p - theta*_p) - 1))) / n_fgo,
//             theta_p = atan2 of the predicted (sin, cos) pair, theta*_p the target angle
// out[0..2] = the three losses, out[3..5] = their divisors.
struct InstLossArgs {
  const float* center; int ld_c;
  const float* offset; int ld_o;
  const float* orient; int ld_r;          // may be NULL (no orientation task)
  const float* center_gt;                 // [pixels]
  const float* offset_gt;                 // [pixels][2]
  const float* orient_gt;                 // [pixels] angle in rad (NULL with orient)
  const uint8_t* center_mask;             // [pixels] or NULL
  const uint8_t* fg;                      // [pixels] instance foreground
  const uint8_t* fg_orient;               // [pixels] foreground with an orientation label
  long pixels;
  float kappa;
};

__device__ __forceinline__ void block_sum6(float (&v)[6], float* partial, int n_blocks) {
  __shared__ float red[6][4];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < 6; ++k) red[k][threadIdx.x >> 6] = v[k];
  __syncthreads();
  if (threadIdx.x < 6)
    partial[(size_t)threadIdx.x * n_blocks + blockIdx.x] =
        red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
}

__global__ __launch_bounds__(256) void inst_loss_fwd_kernel(const InstLossArgs a,
                                                            float* __restrict__ partial) {
  float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long p = blockIdx.x * 256L + threadIdx.x; p < a.pixels; p += gridDim.x * 256L) {
    const float mc = a.center_mask ? (a.center_mask[p] ? 1.f : 0.f) : 1.f;
    const float dc = a.center[p * a.ld_c] - a.center_gt[p];
    v[0] += mc * dc * dc;
    v[3] += mc;
    if (a.fg[p]) {
      v[1] += fabsf(a.offset[p * a.ld_o] - a.offset_gt[2 * p]) +
              fabsf(a.offset[p * a.ld_o + 1] - a.offset_gt[2 * p + 1]);
      v[4] += 2.f;
    }
    if (a.orient && a.fg_orient[p]) {
      const float s = a.orient[p * a.ld_r], c = a.orient[p * a.ld_r + 1];
      const float r = rsqrtf(fmaxf(s * s + c * c, 1e-12f));
      float sg, cg;
      sincosf(a.orient_gt[p], &sg, &cg);
      const float cosd = (s * sg + c * cg) * r;
      v[2] += 1.f - expf(a.kappa * (cosd - 1.f));
      v[5] += 1.f;
    }
  }
  block_sum6(v, partial, gridDim.x);
}

__global__ void inst_loss_finalize_kernel(const float* __restrict__ partial, int n,
                                          float* __restrict__ out) {
  __shared__ double sh[6][64];
  const int k = threadIdx.x >> 6, l = threadIdx.x & 63;       // 384 threads = 6 x 64
  double acc = 0.0;
  for (int i = l; i < n; i += 64) acc += (double)partial[(size_t)k * n + i];
  sh[k][l] = acc;
  __syncthreads();
  if (l == 0) {
    for (int i = 1; i < 64; ++i) acc += sh[k][i];
    sh[k][0] = acc;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double num = sh[threadIdx.x][0], den = sh[threadIdx.x + 3][0];
    out[threadIdx.x] = den > 0.0 ? (float)(num / den) : 0.f;
    out[threadIdx.x + 3] = (float)den;
  }
}

// d_center / d_offset / d_orient (same strides as the predictions) for upstream gradients g[0..2]
__global__ __launch_bounds__(256) void inst_loss_bwd_kernel(const InstLossArgs a,
                                                            const float* __restrict__ sums,
                                                            const float* __restrict__ g,
                                                            float* __restrict__ d_center, int ldd_c,
                                                            float* __restrict__ d_offset, int ldd_o,
                                                            float* __restrict__ d_orient, int ldd_r) {
  const float kc = sums[3] > 0.f ? 2.f * g[0] / sums[3] : 0.f;
  const float ko = sums[4] > 0.f ? g[1] / sums[4] : 0.f;
  const float kr = sums[5] > 0.f ? g[2] / sums[5] : 0.f;
  for (long p = blockIdx.x * 256L + threadIdx.x; p < a.pixels; p += gridDim.x * 256L) {
    const float mc = a.center_mask ? (a.center_mask[p] ? 1.f : 0.f) : 1.f;
    d_center[p * ldd_c] = kc * mc * (a.center[p * a.ld_c] - a.center_gt[p]);
    float o0 = 0.f, o1 = 0.f;
    if (a.fg[p]) {
      const float e0 = a.offset[p * a.ld_o] - a.offset_gt[2 * p];
      const float e1 = a.offset[p * a.ld_o + 1] - a.offset_gt[2 * p + 1];
      o0 = e0 > 0.f ? ko : (e0 < 0.f ? -ko : 0.f);
      o1 = e1 > 0.f ? ko : (e1 < 0.f ? -ko : 0.f);
    }
    d_offset[p * ldd_o] = o0;
    d_offset[p * ldd_o + 1] = o1;
    if (a.orient) {
      float r0 = 0.f, r1 = 0.f;
      if (a.fg_orient[p]) {
        const float s = a.orient[p * a.ld_r], c = a.orient[p * a.ld_r + 1];
        const float n2 = fmaxf(s * s + c * c, 1e-12f), r = rsqrtf(n2);
        float sg, cg;
        sincosf(a.orient_gt[p], &sg, &cg);
        const float dot = s * sg + c * cg, cosd = dot * r;
        // L = 1 - exp(kappa (cosd - 1)),  dL/dcosd = -kappa exp(.),  dcosd/ds = (sg - cosd s r) r
        const float dl = -a.kappa * expf(a.kappa * (cosd - 1.f)) * kr;
        r0 = dl * (sg - cosd * s * r) * r;
        r1 = dl * (cg - cosd * c * r) * r;
      }
      d_orient[p * ldd_r] = r0;
      d_orient[p * ldd_r + 1] = r1;
    }
  }
}

int inst_grid(long pixels) {
  long g = (pixels + 255) / 256;
  return (int)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int emsa_ce_semantic_blocks(int64_t pixels) { return (int)((pixels + kPix - 1) / kPix); }

extern "C" int emsa_ce_semantic_fwd(const float* logits, int32_t ld, const int64_t* target,
                                    const float* weights, int32_t n_classes, int64_t pixels,
                                    float label_smoothing, float weights_sum, float* partial,
                                    float* out, void* stream) {
  if (!logits || !target || !weights || !partial || !out) return EMSA_E_ARG;
  if (n_classes < 1 || n_classes > 64 || (ld & 3) || ld < ((n_classes + 3) & ~3) ||
      (((uintptr_t)logits) & 15) || label_smoothing < 0.f || label_smoothing >= 1.f)
    return EMSA_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int c4 = (n_classes + 3) / 4;
  const float e = label_smoothing, ws = weights_sum;
  if (c4 <= 4) return launch_fwd<4>(logits, ld, target, weights, n_classes, pixels, e, ws, partial, out, st);
  if (c4 <= 10) return launch_fwd<10>(logits, ld, target, weights, n_classes, pixels, e, ws, partial, out, st);
  return launch_fwd<16>(logits, ld, target, weights, n_classes, pixels, e, ws, partial, out, st);
}

extern "C" int emsa_ce_semantic_bwd(const float* logits, int32_t ld, const int64_t* target,
                                    const float* weights, int32_t n_classes, int64_t pixels,
                                    float label_smoothing, float weights_sum, const float* sums,
                                    const float* grad_out, float* dlogits, int32_t ld_d,
                                    void* stream) {
  if (!logits || !target || !weights || !sums || !grad_out || !dlogits) return EMSA_E_ARG;
  if (n_classes < 1 || n_classes > 64 || (ld & 3) || (ld_d & 3) || ld < ((n_classes + 3) & ~3) ||
      ld_d < ((n_classes + 3) & ~3) || (((uintptr_t)logits) & 15) || (((uintptr_t)dlogits) & 15) ||
      label_smoothing < 0.f || label_smoothing >= 1.f)
    return EMSA_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int c4 = (n_classes + 3) / 4;
  const float e = label_smoothing, ws = weights_sum;
  if (c4 <= 4)
    return launch_bwd<4>(logits, ld, target, weights, n_classes, pixels, e, ws, sums, grad_out, dlogits, ld_d, st);
  if (c4 <= 10)
    return launch_bwd<10>(logits, ld, target, weights, n_classes, pixels, e, ws, sums, grad_out, dlogits, ld_d, st);
  return launch_bwd<16>(logits, ld, target, weights, n_classes, pixels, e, ws, sums, grad_out, dlogits, ld_d, st);
}

extern "C" int emsa_instance_loss_blocks(int64_t pixels) { return inst_grid((long)pixels); }

static int fill_inst_args(InstLossArgs& a, const float* center, int32_t ld_c, const float* offset,
                          int32_t ld_o, const float* orient, int32_t ld_r,
                          const float* center_gt, const float* offset_gt, const float* orient_gt,
                          const uint8_t* center_mask, const uint8_t* fg, const uint8_t* fg_orient,
                          int64_t pixels, float kappa) {
  if (!center || !offset || !center_gt || !offset_gt || !fg) return EMSA_E_ARG;
  if (orient && (!orient_gt || !fg_orient)) return EMSA_E_ARG;
  if (ld_c < 1 || ld_o < 2 || (orient && ld_r < 2) || pixels < 1) return EMSA_E_SHAPE;
  a.center = center; a.ld_c = ld_c; a.offset = offset; a.ld_o = ld_o; a.orient = orient;
  a.ld_r = ld_r; a.center_gt = center_gt; a.offset_gt = offset_gt; a.orient_gt = orient_gt;
  a.center_mask = center_mask; a.fg = fg; a.fg_orient = fg_orient; a.pixels = (long)pixels;
  a.kappa = kappa;
  return EMSA_OK;
}

extern "C" int emsa_instance_loss_fwd(const float* center, int32_t ld_c, const float* offset,
                                      int32_t ld_o, const float* orient, int32_t ld_r,
                                      const float* center_gt, const float* offset_gt,
                                      const float* orient_gt, const uint8_t* center_mask,
                                      const uint8_t* fg, const uint8_t* fg_orient, int64_t pixels,
                                      float kappa, float* partial, float* out, void* stream) {
  InstLossArgs a;
  const int rc = fill_inst_args(a, center, ld_c, offset, ld_o, orient, ld_r, center_gt, offset_gt,
                                orient_gt, center_mask, fg, fg_orient, pixels, kappa);
  if (rc != EMSA_OK) return rc;
  if (!partial || !out) return EMSA_E_ARG;
  const int grid = inst_grid(a.pixels);
  hipLaunchKernelGGL(inst_loss_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, partial);
  hipLaunchKernelGGL(inst_loss_finalize_kernel, dim3(1), dim3(384), 0, (hipStream_t)stream, partial,
                     grid, out);
  return emsa_launch_status();
}

extern "C" int emsa_instance_loss_bwd(const float* center, int32_t ld_c, const float* offset,
                                      int32_t ld_o, const float* orient, int32_t ld_r,
                                      const float* center_gt, const float* offset_gt,
                                      const float* orient_gt, const uint8_t* center_mask,
                                      const uint8_t* fg, const uint8_t* fg_orient, int64_t pixels,
                                      float kappa, const float* sums, const float* grad_out,
                                      float* d_center, int32_t ldd_c, float* d_offset,
                                      int32_t ldd_o, float* d_orient, int32_t ldd_r,
                                      void* stream) {
  InstLossArgs a;
  const int rc = fill_inst_args(a, center, ld_c, offset, ld_o, orient, ld_r, center_gt, offset_gt,
                                orient_gt, center_mask, fg, fg_orient, pixels, kappa);
  if (rc != EMSA_OK) return rc;
  if (!sums || !grad_out || !d_center || !d_offset || (orient && !d_orient)) return EMSA_E_ARG;
  if (ldd_c < 1 || ldd_o < 2 || (orient && ldd_r < 2)) return EMSA_E_SHAPE;
  hipLaunchKernelGGL(inst_loss_bwd_kernel, dim3(inst_grid(a.pixels)), dim3(256), 0,
                     (hipStream_t)stream, a, sums, grad_out, d_center, ldd_c, d_offset, ldd_o,
                     d_orient, ldd_r);
  return emsa_launch_status();
}
