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
    if (p0 + px < pixels) v = emsa_ld4(x + (p0 + px) * (long)ld + c4 * 4);
    emsa_st4(tile + px * LD + c4 * 4, v);
  }
}

template <int C4>
__global__ __launch_bounds__(kPix) void ce_fwd_kernel(const float* __restrict__ logits, int ld,
                                                      const int64_t* __restrict__ target,
                                                      const float* __restrict__ w, int n_classes,
                                                      long pixels, float* __restrict__ partial) {
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
      loss = wt * (logf(s) + m - xt);
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
                                                      long pixels, const float* __restrict__ sums,
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
    const float k = g * w[t], inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < C4 * 4; ++c)
      v[c] = c < n_classes ? k * (v[c] * inv - (c == t ? 1.f : 0.f)) : 0.f;
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
    if (p0 + px < pixels) emsa_st4(dlogits + (p0 + px) * (long)ld_d + c4 * 4, emsa_ld4(tile + px * LD + c4 * 4));
  }
}

template <int C4>
int launch_fwd(const float* x, int ld, const int64_t* t, const float* w, int nc, long pixels,
               float* partial, float* out, hipStream_t st) {
  const int grid = (int)((pixels + kPix - 1) / kPix);
  const size_t lds = (size_t)kPix * (C4 * 4 + 4) * sizeof(float);
  hipLaunchKernelGGL(ce_fwd_kernel<C4>, dim3(grid), dim3(kPix), lds, st, x, ld, t, w, nc, pixels,
                     partial);
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, partial, grid, out);
  return emsa_launch_status();
}
template <int C4>
int launch_bwd(const float* x, int ld, const int64_t* t, const float* w, int nc, long pixels,
               const float* sums, const float* gout, float* dx, int ld_d, hipStream_t st) {
  const int grid = (int)((pixels + kPix - 1) / kPix);
  const size_t lds = (size_t)kPix * (C4 * 4 + 4) * sizeof(float);
  hipLaunchKernelGGL(ce_bwd_kernel<C4>, dim3(grid), dim3(kPix), lds, st, x, ld, t, w, nc, pixels,
                     sums, gout, dx, ld_d);
  return emsa_launch_status();
}

}  // namespace

extern "C" int emsa_ce_semantic_blocks(int64_t pixels) { return (int)((pixels + kPix - 1) / kPix); }

extern "C" int emsa_ce_semantic_fwd(const float* logits, int32_t ld, const int64_t* target,
                                    const float* weights, int32_t n_classes, int64_t pixels,
                                    float* partial, float* out, void* stream) {
  if (!logits || !target || !weights || !partial || !out) return EMSA_E_ARG;
  if (n_classes < 1 || n_classes > 64 || (ld & 3) || ld < ((n_classes + 3) & ~3) ||
      (((uintptr_t)logits) & 15))
    return EMSA_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int c4 = (n_classes + 3) / 4;
  if (c4 <= 4) return launch_fwd<4>(logits, ld, target, weights, n_classes, pixels, partial, out, st);
  if (c4 <= 10) return launch_fwd<10>(logits, ld, target, weights, n_classes, pixels, partial, out, st);
  return launch_fwd<16>(logits, ld, target, weights, n_classes, pixels, partial, out, st);
}

extern "C" int emsa_ce_semantic_bwd(const float* logits, int32_t ld, const int64_t* target,
                                    const float* weights, int32_t n_classes, int64_t pixels,
                                    const float* sums, const float* grad_out, float* dlogits,
                                    int32_t ld_d, void* stream) {
  if (!logits || !target || !weights || !sums || !grad_out || !dlogits) return EMSA_E_ARG;
  if (n_classes < 1 || n_classes > 64 || (ld & 3) || (ld_d & 3) || ld < ((n_classes + 3) & ~3) ||
      ld_d < ((n_classes + 3) & ~3) || (((uintptr_t)logits) & 15) || (((uintptr_t)dlogits) & 15))
    return EMSA_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int c4 = (n_classes + 3) / 4;
  if (c4 <= 4)
    return launch_bwd<4>(logits, ld, target, weights, n_classes, pixels, sums, grad_out, dlogits, ld_d, st);
  if (c4 <= 10)
    return launch_bwd<10>(logits, ld, target, weights, n_classes, pixels, sums, grad_out, dlogits, ld_d, st);
  return launch_bwd<16>(logits, ld, target, weights, n_classes, pixels, sums, grad_out, dlogits, ld_d, st);
}
