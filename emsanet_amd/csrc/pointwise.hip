// HBM-bound kernels of the EMSANet engine: BatchNorm (+ReLU/+Dropout2d/+residual), max pool,
// squeeze-and-excitation fusion, nearest-x2 + depth-wise 3x3 upsampling, pyramid pooling,
// head activations, weight packing.  All NHWC fp32, float4 per lane along channels so that a
// wave touches whole 128-B lines; grids capped at ~8 blocks/CU with grid-stride loops
// (cdna_hip_programming.md Guideline 11/13).
//
// Reference modules these stand in for are cited per entry point in include/emsanet_hip.h.
#include <type_traits>

#include <cstdlib>

#include <atomic>
#include "common.h"

namespace {

constexpr int kThreads = 256;
#ifndef EMSA_PW_BLOCKS_PER_CU
#define EMSA_PW_BLOCKS_PER_CU 8
#endif
constexpr int kMaxBlocks = 256 * EMSA_PW_BLOCKS_PER_CU;

inline int grid_for(long work_items) {
  long b = (work_items + kThreads - 1) / kThreads;
  if (b < 1) b = 1;
  if (b > kMaxBlocks) b = kMaxBlocks;
  return (int)b;
}

// ------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------
// mode 0: OIHW -> [tap][cout_total][cin_total]; mode 1: OIHW -> [tap][cin_total][cout_total];
// mode 2: [tap][cout_total][cin_total] -> OIHW
// mode 3: OIHW -> both packed layouts (dst = forward, dst2 = data gradient)
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout,
                                   int cin, int kh, int kw, int cout_total, int cout_off,
                                   int cin_total, int cin_off, int mode,
                                   float* __restrict__ dst2 = nullptr) {
  const long total = (long)cout * cin * kh * kw;
  const int taps = kh * kw;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    // i enumerates OIHW
    const int tap = (int)(i % taps);
    const long r = i / taps;
    const int ci = (int)(r % cin), co = (int)(r / cin);
    if (mode == 3) {
      const float v = src[i];
      dst[((long)tap * cout_total + cout_off + co) * cin_total + cin_off + ci] = v;
      dst2[((long)tap * cin_total + cin_off + ci) * cout_total + cout_off + co] = v;
    } else if (mode == 0) {
      dst[((long)tap * cout_total + cout_off + co) * cin_total + cin_off + ci] = src[i];
    } else if (mode == 1) {
      dst[((long)tap * cin_total + cin_off + ci) * cout_total + cout_off + co] = src[i];
    } else {
      dst[i] = src[((long)tap * cout_total + cout_off + co) * cin_total + cin_off + ci];
    }
  }
}

// 16-bit operands of conv_h.hip from the fp32 OIHW parameter: dst = [tap][cout_total][cin_total],
// dst2 = [tap][cin_total][cout_total] (either may be NULL)
template <typename T>
__global__ void pack_weight_t_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                     T* __restrict__ dst2, int cout, int cin, int taps,
                                     int cout_total, int cout_off, int cin_total, int cin_off) {
  const long total = (long)cout * cin * taps;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const long r = i / taps;
    const int ci = (int)(r % cin), co = (int)(r / cin);
    const float v = src[i];
    if (dst) emsa_st1(dst + ((long)tap * cout_total + cout_off + co) * cin_total + cin_off + ci, v);
    if (dst2) emsa_st1(dst2 + ((long)tap * cin_total + cin_off + ci) * cout_total + cout_off + co, v);
  }
}

// stem: NCHW -> zero padded NHWC4 [n][h][w+8][4]
template <typename T>
__global__ void stem_pack_input_kernel(const float* __restrict__ x, T* __restrict__ y, int n,
                                       int c, int h, int w) {
  const int wp = w + 8;
  const long total = (long)n * h * wp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int col = (int)(i % wp);
    const long r = i / wp;
    const int row = (int)(r % h), img = (int)(r / h);
    const int sw = col - 3;
    float4 v = emsa_zero4();
    if (sw >= 0 && sw < w) {
      const long base = ((long)img * c * h + row) * w + sw;
      const long plane = (long)h * w;
      v.x = x[base];
      if (c > 1) v.y = x[base + plane];
      if (c > 2) v.z = x[base + 2 * plane];
      if (c > 3) v.w = x[base + 3 * plane];
    }
    emsa_st4(y + i * 4, v);
  }
}

// One-channel stem (the depth encoder): the four "channel" slots of a packed pixel hold four
// consecutive ROWS of the single input plane instead of one channel and three zeros,
//   y[n][r'][col][r] = x[n][0][r' - 3 + r][col - 3]        (r' in [0, h + 4), zero outside),
// so that one 32-float K chunk covers 8 kw positions x 4 kh rows: the 7x7 kernel is TWO super-taps
// (kh = 0..3 and 4..6) instead of seven row taps -- 2/7 of the matrix work for the same result.
template <typename T>
__global__ void stem_pack_input_rows_kernel(const float* __restrict__ x, T* __restrict__ y, int n,
                                            int h, int w) {
  const int wp = w + 8, hp = h + 4;
  const long total = (long)n * hp * wp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int col = (int)(i % wp);
    const long r = i / wp;
    const int row = (int)(r % hp), img = (int)(r / hp);
    const int sw = col - 3;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (sw >= 0 && sw < w) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int sr = row - 3 + k;
        if (sr >= 0 && sr < h) v[k] = x[((long)img * h + sr) * w + sw];
      }
    }
    emsa_st4(y + i * 4, make_float4(v[0], v[1], v[2], v[3]));
  }
}
// weights of the one-channel stem: OIHW [cout][1][7][7] <-> [2 (super-tap)][cout][32 = 8(kw) x 4(r)],
// kh = 4 * tap + r
template <typename T>
__global__ void stem_pack_weight_rows_kernel(const float* __restrict__ w, T* __restrict__ wp,
                                             int cout, int unpack) {
  const int total = 2 * cout * 32;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int k = i % 32, co = (i / 32) % cout, t = i / (32 * cout);
    const int kw = k / 4, kh = 4 * t + (k % 4);
    const bool real = kw < 7 && kh < 7;
    if (!unpack) {
      emsa_st1(wp + i, real ? w[(co * 7 + kh) * 7 + kw] : 0.f);
    } else if (real) {
      emsa_st1(wp + (co * 7 + kh) * 7 + kw, w[i]);
    }
  }
}

// stem weights: OIHW [cout][cin][7][7] <-> [7(kh)][cout][32 = 8(kw) x 4(c)]
template <typename T>
__global__ void stem_pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wp,
                                        int cout, int cin, int unpack) {
  const int total = 7 * cout * 32;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int k = i % 32, co = (i / 32) % cout, kh = i / (32 * cout);
    const int kw = k / 4, c = k % 4;
    const bool real = kw < 7 && c < cin;
    if (!unpack) {
      emsa_st1(wp + i, real ? w[((co * cin + c) * 7 + kh) * 7 + kw] : 0.f);
    } else if (real) {
      // here `w` is the packed gradient and `wp` the OIHW destination
      emsa_st1(wp + ((co * cin + c) * 7 + kh) * 7 + kw, w[i]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// BatchNorm
// stats = float[3][rows][c]: per-tile sum, M2 (about the tile mean), count.
// Two-level merge in fp64.  Level 1 (grid = channel groups x row slices): per slice
//   S0 = sum n_t, S1 = sum sum_t, Q = sum (M2_t + sum_t^2 / n_t)        [= sum of squares]
// Level 2: var = (Q - S1^2/S0) / S0.  The tile-local M2 keeps the fp32 partials exact to fp32
// roundoff; the cross-tile combination runs entirely in fp64 (53-bit mantissa over 24-bit
// inputs), so the textbook cancellation costs nothing measurable for mean^2/var up to ~1e8.
constexpr int kBnSlices = 64;
__global__ void bn_partial_kernel(const float* __restrict__ stats, int rows, int c,
                                  double* __restrict__ ws) {
  __shared__ double sh[3][8][32];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int ch = blockIdx.x * 32 + cl;
  const int slices = gridDim.y, sl = blockIdx.y;
  const int per = (rows + slices - 1) / slices;
  const int r0 = sl * per, r1 = min(r0 + per, rows);
  double s0 = 0.0, s1 = 0.0, q = 0.0;
  if (ch < c)
    for (int r = r0 + rg; r < r1; r += 8) {
      const double nb = (double)stats[((long)2 * rows + r) * c + ch];
      if (nb > 0.0) {
        const double sb = (double)stats[((long)0 * rows + r) * c + ch];
        s0 += nb;
        s1 += sb;
        q += (double)stats[((long)1 * rows + r) * c + ch] + sb * sb / nb;
      }
    }
  sh[0][rg][cl] = s0; sh[1][rg][cl] = s1; sh[2][rg][cl] = q;
  __syncthreads();
  if (rg == 0 && ch < c) {
    s0 = s1 = q = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s0 += sh[0][k][cl]; s1 += sh[1][k][cl]; q += sh[2][k][cl]; }
    double* o = ws + ((long)sl * c + ch) * 3;
    o[0] = s0; o[1] = s1; o[2] = q;
  }
}

__global__ void bn_finalize_kernel(const double* __restrict__ ws, int slices, int c,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float eps, float momentum, float* running_mean,
                                   float* running_var, float* __restrict__ scale,
                                   float* __restrict__ shift, float* __restrict__ save_mean,
                                   float* __restrict__ save_invstd) {
  // 32 channels x 8 slice groups per workgroup (one thread per channel walking up to 64 slices
  // was a 10 us chain of dependent round trips, 124 times per training step)
  __shared__ double sh[3][8][32];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int ch = blockIdx.x * 32 + cl;
  double n = 0.0, s1 = 0.0, q = 0.0;
  if (ch < c)
    for (int sl = rg; sl < slices; sl += 8) {
      const double* o = ws + ((long)sl * c + ch) * 3;
      n += o[0]; s1 += o[1]; q += o[2];
    }
  sh[0][rg][cl] = n; sh[1][rg][cl] = s1; sh[2][rg][cl] = q;
  __syncthreads();
  if (rg != 0 || ch >= c) return;
  n = s1 = q = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { n += sh[0][k][cl]; s1 += sh[1][k][cl]; q += sh[2][k][cl]; }
  const double mean = n > 0.0 ? s1 / n : 0.0;
  double m2 = q - s1 * mean;
  if (m2 < 0.0) m2 = 0.0;
  const double var = n > 0.0 ? m2 / n : 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[ch] * invstd;
  scale[ch] = sc;
  shift[ch] = beta[ch] - (float)mean * sc;
  save_mean[ch] = (float)mean;
  save_invstd[ch] = invstd;
  if (running_mean) {
    const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
    running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)mean;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
  }
}

// One launch instead of two when the statistics arrive in few rows (<= 512: conv_rs.hip writes one
// row per persistent workgroup; the /32 stage of the fp32 kernels; small images): level 1 and level 2 of the merge above in the same
// workgroup, same fp64 arithmetic, same summation order per row group.
template <int RG>
__global__ __launch_bounds__(4 * RG) void bn_finalize_rows_kernel(const float* __restrict__ stats, int rows, int c,
                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                        float eps, float momentum, float* running_mean,
                                        float* running_var, float* __restrict__ scale,
                                        float* __restrict__ shift, float* __restrict__ save_mean,
                                        float* __restrict__ save_invstd) {
  // 4 channels x 64 row groups per workgroup (grid = c / 4: 16 ... 128 workgroups), branch-free and
  // unrolled: <= 8 row iterations per thread, all loads of a thread in flight together.  (32 x 8
  // with a branch around empty rows ran 64 dependent L2 round trips: 21 us; 16 x 16 still 13-19.)
  // RG = 64 row groups (<= 512 rows) or 256 (1024 threads; round 5: the fp32 kernels' 513 ... 4800
  // per-tile rows too -- one launch of ~8 us instead of bn_partial + bn_finalize, 9 + 5 us)
  __shared__ double sh[3][RG][4];
  const int cl = threadIdx.x & 3, rg = threadIdx.x >> 2;
  const int ch = blockIdx.x * 4 + cl;
  const int chc = ch < c ? ch : c - 1;
  double n = 0.0, s1 = 0.0, q = 0.0;
#pragma unroll 8
  for (int r = rg; r < rows; r += RG) {
    const float nbf = stats[((long)2 * rows + r) * c + chc];
    const float sbf = stats[((long)0 * rows + r) * c + chc];
    const float m2f = stats[((long)1 * rows + r) * c + chc];
    const double nb = (double)nbf, sb = (double)sbf;
    const bool ok = nbf > 0.f;
    n += ok ? nb : 0.0;
    s1 += ok ? sb : 0.0;
    // 1 / nb: the float reciprocal of the (integer) count + one Newton step in fp64 (relative error
    // ~1e-14) instead of an fp64 division (~25 instructions, up to 19 per thread in the wide form)
    const double r0 = (double)(1.0f / (ok ? nbf : 1.f));
    const double rn = r0 * (2.0 - (ok ? nb : 1.0) * r0);
    q += ok ? (double)m2f + sb * sb * rn : 0.0;
  }
  // second level: the 16 row groups of a wave meet through lane shuffles (xor 4, 8, 16, 32: lanes
  // with the same channel), the waves through LDS -- a fixed tree, so the result does not depend on
  // the run.  (One thread summing the 64 / 256 row groups from LDS was a serial chain of 64 dependent
  // fp64 additions: ~3 of the launch's 7-10 us.)
#pragma unroll
  for (int off = 4; off < 64; off <<= 1) {
    n += __shfl_xor(n, off);
    s1 += __shfl_xor(s1, off);
    q += __shfl_xor(q, off);
  }
  constexpr int NW = 4 * RG / 64;                  // waves of the workgroup
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) < 4) { sh[0][wv][cl] = n; sh[1][wv][cl] = s1; sh[2][wv][cl] = q; }
  __syncthreads();
  if (rg != 0 || ch >= c) return;
  n = s1 = q = 0.0;
#pragma unroll
  for (int k = 0; k < NW; ++k) { n += sh[0][k][cl]; s1 += sh[1][k][cl]; q += sh[2][k][cl]; }
  const double mean = n > 0.0 ? s1 / n : 0.0;
  double m2 = q - s1 * mean;
  if (m2 < 0.0) m2 = 0.0;
  const double var = n > 0.0 ? m2 / n : 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[ch] * invstd;
  scale[ch] = sc;
  shift[ch] = beta[ch] - (float)mean * sc;
  save_mean[ch] = (float)mean;
  save_invstd[ch] = invstd;
  if (running_mean) {
    const double unbiased = n > 1.0 ? m2 / (n - 1.0) : var;
    running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * (float)mean;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * (float)unbiased;
  }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm,
                               const float* rv, float eps, int c, float* scale, float* shift,
                               float* invstd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < c) {
    const float is = 1.0f / sqrtf(rv[i] + eps);
    const float sc = gamma[i] * is;
    scale[i] = sc;
    shift[i] = beta[i] - rm[i] * sc;
    if (invstd) invstd[i] = is;
  }
}

// ---- the arithmetic of the BatchNorm passes, spelled out ----------------------------------------
// One fma / mul / add per step, none left to the compiler's contraction heuristics: the general and
// the fast form of a pass (below) must round alike, and what clang fuses depends on the code around
// an expression (a product hoisted out of a loop is no longer fused into the sum behind it).
__device__ __forceinline__ float bn_mul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float bn_add(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
// y = x * scale + shift
__device__ __forceinline__ float bn_affine(float x, float sc, float sh) { return __builtin_fmaf(x, sc, sh); }
// one term of sum g * xhat:  acc + (g * (x - mean)) * invstd
__device__ __forceinline__ float bn_acc_xhat(float acc, float g, float x, float mu, float is) {
  return __builtin_fmaf(bn_mul(g, bn_add(x, -mu)), is, acc);
}
// dx = gamma * invstd * (g - sum_g / N - xhat * sum_gxhat / N), gi = gamma * invstd (bn_mul)
__device__ __forceinline__ float bn_dx_train(float g, float x, float mu, float is, float gi, float db,
                                             float dg, float inv_count) {
  const float xh = bn_mul(bn_add(x, -mu), is);
  const float t = __builtin_fmaf(-db, inv_count, g);
  return bn_mul(gi, __builtin_fmaf(-bn_mul(xh, dg), inv_count, t));
}

// ---- vector access per lane: 16 bytes = 4 fp32 or 8 16-bit channels ---------------------------
// (the first 16-bit versions of the three BatchNorm passes moved 4 channels = 8 bytes per lane and
//  ran at the SAME time as their fp32 twins on half the bytes: 35 vs 43, 35 vs 34, 23 vs 32 us)
template <typename T> struct VecIO {
  static constexpr int V = 8;
  typedef typename std::conditional<std::is_same<T, emsa_f16>::value, _Float16, __bf16>::type E;
  typedef E ev __attribute__((ext_vector_type(8)));
  typedef float fv __attribute__((ext_vector_type(8)));
  static __device__ __forceinline__ void load(const T* p, float (&v)[8]) {
    const fv f = __builtin_convertvector(*reinterpret_cast<const ev*>(p), fv);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = f[k];
  }
  // the 16 bytes as they are (the fast BatchNorm passes issue all loads of an iteration first and
  // convert where the values are used)
  typedef ev raw;
  static __device__ __forceinline__ raw load_raw(const T* p) { return *reinterpret_cast<const ev*>(p); }
  static __device__ __forceinline__ void cvt(const raw& r, float (&v)[8]) {
    const fv f = __builtin_convertvector(r, fv);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = f[k];
  }
  static __device__ __forceinline__ void store(T* p, const float (&v)[8]) {
    fv f;
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = v[k];
    *reinterpret_cast<ev*>(p) = __builtin_convertvector(f, ev);
  }
};
template <> struct VecIO<float> {
  static constexpr int V = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 f = emsa_ld4(p);
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
  }
  typedef float4 raw;
  static __device__ __forceinline__ raw load_raw(const float* p) { return emsa_ld4(p); }
  static __device__ __forceinline__ void cvt(const raw& f, float (&v)[4]) {
    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    emsa_st4(p, make_float4(v[0], v[1], v[2], v[3]));
  }
};
template <int V>
__device__ __forceinline__ void ldf(const float* p, float (&v)[V]) {       // V fp32 parameters
#pragma unroll
  for (int k = 0; k < V; k += 4) {
    const float4 f = emsa_ld4(p + k);
    v[k] = f.x; v[k + 1] = f.y; v[k + 2] = f.z; v[k + 3] = f.w;
  }
}

// `cvn` = channel vectors per pixel (c / V), `totalv` = pixels * cvn.  mask_bits: (y > 0) as 1 bit
// per element for the backward pass (which otherwise re-reads y only for this): the wave's 64
// vector indices [i & ~63, +64) -> V words, one per component; lanes past the end are inactive and
// contribute 0 bits
template <typename T>
__global__ void bn_act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                  const float* __restrict__ drop,
                                  const T* __restrict__ residual, long hw, int cvn, long totalv,
                                  int act, uint64_t* __restrict__ mask_bits) {
  constexpr int V = VecIO<T>::V;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < totalv;
       i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % cvn);
    const long pix = i / cvn;
    float v[V], sc[V], sh[V];
    VecIO<T>::load(x + i * V, v);
    ldf<V>(scale + cv * V, sc);
    ldf<V>(shift + cv * V, sh);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = bn_affine(v[k], sc[k], sh[k]);
    if (drop) {
      float d[V];
      ldf<V>(drop + (pix / hw) * (long)cvn * V + cv * V, d);
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = bn_mul(v[k], d[k]);
    }
    if (residual) {
      float r[V];
      VecIO<T>::load(residual + i * V, r);
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = bn_add(v[k], r[k]);
    }
    if (act == EMSA_ACT_RELU) {
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    VecIO<T>::store(y + i * V, v);
    if (mask_bits) {
      uint64_t b[V];
#pragma unroll
      for (int k = 0; k < V; ++k) b[k] = __ballot(v[k] > 0.f);
      if ((i & 63) == 0) {
        uint64_t* m = mask_bits + (i >> 6) * V;
#pragma unroll
        for (int k = 0; k < V; ++k) m[k] = b[k];
      }
    }
  }
}

// ---- fast forms of the three BatchNorm passes (round 5) ------------------------------------------
// The loops above cost the bf16 step 2.4-3.8 TB/s of algorithmic bytes where a streaming kernel
// reaches 5 (profiles/r05_l_pointwise_bench16*.txt): per iteration two 64-bit divisions (i % cvn,
// pix / hw: ~200 instructions), the per-channel vectors re-loaded from global memory, and the
// tensor loads SERIALISED -- x, wait, drop (its address depends on the division), wait, residual,
// wait: three memory round trips where one is needed.  The fast forms, taken when the channel-vector
// count is a power of two <= 256 (every BatchNorm of the model) and the tensor has < 2^31 vectors:
//   * 32-bit indices; the grid stride is a multiple of cvn, so a thread's channel vector never
//     changes: scale / shift / gamma / mean / invstd are loaded ONCE into registers;
//   * optional operands are template parameters -- no control flow between a load and its use, all
//     loads of an iteration (two elements per iteration) are issued before the first is consumed;
//   * the same arithmetic in the same order (bn_affine / bn_mul / bn_dx_train ...: every fma spelled
//     out): results are bit-identical to the loops above, which stay as the general form
//     (EMSA_BN_FAST=0 selects them for A/B runs; tests/test_ops16_gpu.py compares the two).
template <typename T, bool DROP, bool RES>
__global__ __launch_bounds__(kThreads) void bn_act_fwd_fast_kernel(
    const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ drop, const T* __restrict__ residual,
    uint32_t hw, int cvn_log2, uint32_t totalv, int act, uint64_t* __restrict__ mask_bits) {
  constexpr int V = VecIO<T>::V;
  typedef typename VecIO<T>::raw Raw;
  const uint32_t stride = gridDim.x * kThreads;
  uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  const uint32_t cvn = 1u << cvn_log2, cv = i & (cvn - 1);
  const uint32_t dpix = stride >> cvn_log2;
  uint32_t pix = i >> cvn_log2;
  float sc[V], sh[V];
  ldf<V>(scale + cv * V, sc);
  ldf<V>(shift + cv * V, sh);
  auto finish = [&](uint32_t idx, const Raw& rx, const Raw& rr, const float (&d)[V]) {
    float v[V];
    VecIO<T>::cvt(rx, v);
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = bn_affine(v[k], sc[k], sh[k]);
    if constexpr (DROP) {
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = bn_mul(v[k], d[k]);
    }
    if constexpr (RES) {
      float r[V];
      VecIO<T>::cvt(rr, r);
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = bn_add(v[k], r[k]);
    }
    if (act == EMSA_ACT_RELU) {
#pragma unroll
      for (int k = 0; k < V; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    VecIO<T>::store(y + (size_t)idx * V, v);
    if (mask_bits) {
      uint64_t b[V];
#pragma unroll
      for (int k = 0; k < V; ++k) b[k] = __ballot(v[k] > 0.f);
      if ((idx & 63) == 0) {
        uint64_t* m = mask_bits + (size_t)(idx >> 6) * V;
#pragma unroll
        for (int k = 0; k < V; ++k) m[k] = b[k];
      }
    }
  };
  auto ld_drop = [&](uint32_t px, float (&d)[V]) {
    if constexpr (DROP) ldf<V>(drop + ((size_t)(px / hw) * cvn + cv) * V, d);
  };
  // wave-uniform loop bounds (the ballots want whole waves): iw = the wave's first index
  uint32_t iw = i & ~63u;
  for (; iw + stride + 64 <= totalv; iw += 2 * stride, i += 2 * stride, pix += 2 * dpix) {
    const uint32_t i1 = i + stride;
    const Raw x0 = VecIO<T>::load_raw(x + (size_t)i * V);
    const Raw x1 = VecIO<T>::load_raw(x + (size_t)i1 * V);
    Raw r0 = x0, r1 = x1;
    if constexpr (RES) {
      r0 = VecIO<T>::load_raw(residual + (size_t)i * V);
      r1 = VecIO<T>::load_raw(residual + (size_t)i1 * V);
    }
    float d0[V], d1[V];
    ld_drop(pix, d0);
    ld_drop(pix + dpix, d1);
    finish(i, x0, r0, d0);
    finish(i1, x1, r1, d1);
  }
  for (; iw < totalv; iw += stride, i += stride, pix += dpix) {
    if (i < totalv) {
      const Raw x0 = VecIO<T>::load_raw(x + (size_t)i * V);
      Raw r0 = x0;
      if constexpr (RES) r0 = VecIO<T>::load_raw(residual + (size_t)i * V);
      float d0[V];
      ld_drop(pix, d0);
      finish(i, x0, r0, d0);
    }
  }
}

// (y > 0) of vector index i from the bit mask written by bn_act_fwd_kernel
template <int V>
__device__ __forceinline__ void relu_maskv(const uint64_t* __restrict__ bits, long i, bool (&m)[V]) {
  const uint64_t* w = bits + (i >> 6) * V;
  const int sh = (int)(i & 63);
#pragma unroll
  for (int k = 0; k < V; ++k) m[k] = (w[k] >> sh) & 1ull;
}

// generic "per-channel reduction of up to two float4 quantities over a pixel range":
// block = (c4n columns) x (256/c4n row lanes); the functor returns the two float4 terms.
template <typename F>
__device__ __forceinline__ void column_reduce(long p0, long p1, int c4n, F f, float4& o1,
                                              float4& o2, bool& leader, int& c4_out) {
  extern __shared__ __attribute__((aligned(16))) float cred[];   // [2][lanes][c4n*4]
  const int lanes = kThreads / c4n;
  const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
  float4 a1 = emsa_zero4(), a2 = emsa_zero4();
  if (rl < lanes) {
    for (long p = p0 + rl; p < p1; p += lanes) {
      float4 t1, t2;
      f(p, c4, t1, t2);
      a1.x += t1.x; a1.y += t1.y; a1.z += t1.z; a1.w += t1.w;
      a2.x += t2.x; a2.y += t2.y; a2.z += t2.z; a2.w += t2.w;
    }
    emsa_st4(cred + ((0 * lanes + rl) * c4n + c4) * 4, a1);
    emsa_st4(cred + ((1 * lanes + rl) * c4n + c4) * 4, a2);
  }
  __syncthreads();
  leader = rl == 0;
  c4_out = c4;
  o1 = o2 = emsa_zero4();
  if (leader) {
    for (int k = 0; k < lanes; ++k) {
      const float4 t1 = emsa_ld4(cred + ((0 * lanes + k) * c4n + c4) * 4);
      const float4 t2 = emsa_ld4(cred + ((1 * lanes + k) * c4n + c4) * 4);
      o1.x += t1.x; o1.y += t1.y; o1.z += t1.z; o1.w += t1.w;
      o2.x += t2.x; o2.y += t2.y; o2.z += t2.z; o2.w += t2.w;
    }
  }
}

// masked / dropped gradient of one vector and the BatchNorm input beside it (shared by the two
// backward passes)
template <typename T, int V>
__device__ __forceinline__ void bn_bwd_load(const T* __restrict__ dy, const T* __restrict__ y,
                                            const uint64_t* __restrict__ mask_bits,
                                            const float* __restrict__ drop, long i, long pix,
                                            long hw, int cvn, int cv, int act, float (&g)[V],
                                            float (&gres)[V]) {
  VecIO<T>::load(dy + i * V, g);
  if (act == EMSA_ACT_RELU) {
    if (mask_bits) {
      bool m[V];
      relu_maskv<V>(mask_bits, i, m);
#pragma unroll
      for (int k = 0; k < V; ++k) g[k] = m[k] ? g[k] : 0.f;
    } else {
      float yy[V];
      VecIO<T>::load(y + i * V, yy);
#pragma unroll
      for (int k = 0; k < V; ++k) g[k] = yy[k] > 0.f ? g[k] : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) gres[k] = g[k];       // gradient of the residual operand
  if (drop) {
    float d[V];
    ldf<V>(drop + ((pix / hw) * cvn + cv) * V, d);
#pragma unroll
    for (int k = 0; k < V; ++k) g[k] = bn_mul(g[k], d[k]);
  }
}

// block = (cvn channel vectors) x (256 / cvn row lanes) over a chunk of pixels -> partial sums
// (sum g, sum g * xhat) per channel: partial = [2][rows_alloc][c], rows [0, gridDim.x) written here
template <typename T>
__global__ void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                     const uint64_t* __restrict__ mask_bits,
                                     const T* __restrict__ x, const float* __restrict__ mean,
                                     const float* __restrict__ invstd,
                                     const float* __restrict__ drop, long pixels, long hw, int cvn,
                                     int act, int rows_alloc, float* __restrict__ partial) {
  constexpr int V = VecIO<T>::V;
  extern __shared__ __attribute__((aligned(16))) float cred[];   // [2][lanes][c]
  const int rows = gridDim.x;
  const long chunk = (pixels + rows - 1) / rows;
  const long p0 = blockIdx.x * chunk, p1 = min(p0 + chunk, pixels);
  const int lanes = kThreads / cvn, c = cvn * V;
  const int cv = threadIdx.x % cvn, rl = threadIdx.x / cvn;
  float a1[V], a2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) a1[k] = a2[k] = 0.f;
  if (rl < lanes) {
    float mu[V], is[V];
    ldf<V>(mean + cv * V, mu);
    ldf<V>(invstd + cv * V, is);
    for (long p = p0 + rl; p < p1; p += lanes) {
      const long i = p * cvn + cv;
      float g[V], gres[V], xx[V];
      bn_bwd_load<T, V>(dy, y, mask_bits, drop, i, p, hw, cvn, cv, act, g, gres);
      VecIO<T>::load(x + i * V, xx);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        a1[k] = bn_add(a1[k], g[k]);
        a2[k] = bn_acc_xhat(a2[k], g[k], xx[k], mu[k], is[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      cred[(0 * lanes + rl) * c + cv * V + k] = a1[k];
      cred[(1 * lanes + rl) * c + cv * V + k] = a2[k];
    }
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < 2 * c; ch += blockDim.x) {
    const int which = ch / c, cc = ch % c;
    float a = 0.f;
    for (int k = 0; k < lanes; ++k) a += cred[(which * lanes + k) * c + cc];
    partial[((long)which * rows_alloc + blockIdx.x) * c + cc] = a;
  }
}

// the same for tensors with more channel vectors than a workgroup has threads (fp32 with C > 1024: the
// 2048-channel stage of the bottleneck ResNets): grid = (rows, channel-vector groups), one thread per
// channel vector walks the row chunk alone -- no LDS, fixed order
template <typename T>
__global__ void bn_bwd_reduce_wide_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                          const uint64_t* __restrict__ mask_bits,
                                          const T* __restrict__ x, const float* __restrict__ mean,
                                          const float* __restrict__ invstd,
                                          const float* __restrict__ drop, long pixels, long hw,
                                          int cvn, int act, int rows_alloc,
                                          float* __restrict__ partial) {
  constexpr int V = VecIO<T>::V;
  const int cv = blockIdx.y * kThreads + threadIdx.x;
  if (cv >= cvn) return;
  const int rows = gridDim.x, c = cvn * V;
  const long chunk = (pixels + rows - 1) / rows;
  const long p0 = blockIdx.x * chunk, p1 = min(p0 + chunk, pixels);
  float a1[V], a2[V], mu[V], is[V];
#pragma unroll
  for (int k = 0; k < V; ++k) a1[k] = a2[k] = 0.f;
  ldf<V>(mean + cv * V, mu);
  ldf<V>(invstd + cv * V, is);
  for (long p = p0; p < p1; ++p) {
    const long i = p * cvn + cv;
    float g[V], gres[V], xx[V];
    bn_bwd_load<T, V>(dy, y, mask_bits, drop, i, p, hw, cvn, cv, act, g, gres);
    VecIO<T>::load(x + i * V, xx);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      a1[k] = bn_add(a1[k], g[k]);
      a2[k] = bn_acc_xhat(a2[k], g[k], xx[k], mu[k], is[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) {
    partial[((long)0 * rows_alloc + blockIdx.x) * c + cv * V + k] = a1[k];
    partial[((long)1 * rows_alloc + blockIdx.x) * c + cv * V + k] = a2[k];
  }
}

// partial = float[2][rows_alloc][c], rows_alloc = rows + kBwdSlices.  Level 1 (grid = channel
// groups x kBwdSlices): slice sums of rows [0, rows) in fp64 -> rows [rows, rows_alloc).  Level 2
// (merging the kBwdSlices slice sums) is the prologue of bn_bwd_apply_kernel.  A single-level
// version (c/32 workgroups looping over up to 1024 rows) was latency bound at 35 us per layer.
constexpr int kBwdSlices = 16;
__global__ void bn_bwd_sum_kernel(float* __restrict__ partial, int rows, int rows_alloc, int c) {
  __shared__ double red[2][8][32];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int ch = blockIdx.x * 32 + cl, sl = blockIdx.y;
  const int per = (rows + kBwdSlices - 1) / kBwdSlices;
  const int r0 = sl * per, r1 = min(rows, r0 + per);
  double a1 = 0.0, a2 = 0.0;
  if (ch < c) {
#pragma unroll 4
    for (int r = r0 + rg; r < r1; r += 8) {
      a1 += (double)partial[((long)0 * rows_alloc + r) * c + ch];
      a2 += (double)partial[((long)1 * rows_alloc + r) * c + ch];
    }
  }
  red[0][rg][cl] = a1;
  red[1][rg][cl] = a2;
  __syncthreads();
  if (rg == 0 && ch < c) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      a1 += red[0][k][cl];
      a2 += red[1][k][cl];
    }
    partial[((long)0 * rows_alloc + rows + sl) * c + ch] = (float)a1;
    partial[((long)1 * rows_alloc + rows + sl) * c + ch] = (float)a2;
  }
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                    const uint64_t* __restrict__ mask_bits,
                                    const T* __restrict__ x, const float* __restrict__ gamma,
                                    const float* __restrict__ mean,
                                    const float* __restrict__ invstd,
                                    const float* __restrict__ drop,
                                    const float* __restrict__ partial, int rows,
                                    int rows_alloc, float* __restrict__ dbeta_out,
                                    float* __restrict__ dgamma_out, long hw, int cvn,
                                    long totalv, float inv_count, int act, int train,
                                    T* __restrict__ dx, T* __restrict__ dres) {
  constexpr int V = VecIO<T>::V;
  // level 2 of the (sum dy, sum dy*xhat) reduction: merge the slice sums of bn_bwd_sum_kernel
  extern __shared__ __attribute__((aligned(16))) float sums[];   // [2][c]
  const int c = cvn * V;
  float* const dbeta = sums;
  float* const dgamma = sums + c;
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int sl = 0; sl < kBwdSlices; ++sl) {
      a1 += (double)partial[((long)0 * rows_alloc + rows + sl) * c + ch];
      a2 += (double)partial[((long)1 * rows_alloc + rows + sl) * c + ch];
    }
    dbeta[ch] = (float)a1;
    dgamma[ch] = (float)a2;
    if (blockIdx.x == 0) {
      dbeta_out[ch] = (float)a1;
      dgamma_out[ch] = (float)a2;
    }
  }
  __syncthreads();
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < totalv;
       i += (long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % cvn);
    const long pix = i / cvn;
    float g[V], gres[V], o[V], ga[V], is[V];
    bn_bwd_load<T, V>(dy, y, mask_bits, drop, i, pix, hw, cvn, cv, act, g, gres);
    if (dres) VecIO<T>::store(dres + i * V, gres);
    ldf<V>(gamma + cv * V, ga);
    ldf<V>(invstd + cv * V, is);
    if (train) {
      float xx[V], mu[V], db[V], dg[V];
      VecIO<T>::load(x + i * V, xx);
      ldf<V>(mean + cv * V, mu);
      ldf<V>(dbeta + cv * V, db);
      ldf<V>(dgamma + cv * V, dg);
#pragma unroll
      for (int k = 0; k < V; ++k)
        o[k] = bn_dx_train(g[k], xx[k], mu[k], is[k], bn_mul(ga[k], is[k]), db[k], dg[k], inv_count);
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = bn_mul(bn_mul(g[k], ga[k]), is[k]);
    }
    VecIO<T>::store(dx + i * V, o);
  }
}

// fast form of bn_bwd_reduce_kernel (see bn_act_fwd_fast_kernel): same thread -> (pixel lane,
// channel vector) mapping and the same summation order, i.e. bit-identical partial rows; MASK = 0
// none / 1 the bit mask / 2 the activation output y
template <typename T, int MASK, bool DROP>
struct BnBwdElem {
  typename VecIO<T>::raw g, x, y;
  ulonglong2 w[VecIO<T>::V / 2];
  float d[VecIO<T>::V];
};
template <typename T, int MASK, bool DROP>
__device__ __forceinline__ void bn_bwd_issue(BnBwdElem<T, MASK, DROP>& e, const T* __restrict__ dy,
                                             const T* __restrict__ y,
                                             const uint64_t* __restrict__ mask_bits,
                                             const T* __restrict__ x, const float* __restrict__ drop,
                                             uint32_t i, uint32_t pix, uint32_t hw, uint32_t cvn,
                                             uint32_t cv, bool want_x) {
  constexpr int V = VecIO<T>::V;
  e.g = VecIO<T>::load_raw(dy + (size_t)i * V);
  if (want_x || MASK == 3) e.x = VecIO<T>::load_raw(x + (size_t)i * V);
  if constexpr (MASK == 1) {
    const ulonglong2* w = reinterpret_cast<const ulonglong2*>(mask_bits + (size_t)(i >> 6) * V);
#pragma unroll
    for (int k = 0; k < V / 2; ++k) e.w[k] = w[k];
  }
  if constexpr (MASK == 2) e.y = VecIO<T>::load_raw(y + (size_t)i * V);
  if constexpr (DROP) ldf<V>(drop + ((size_t)(pix / hw) * cvn + cv) * V, e.d);
}
// -> g (masked, dropped), gres (masked): the values bn_bwd_load produces
// MASK = 3 (round 6): the ReLU decision recomputed from the BatchNorm's INPUT x with the forward pass's
// own scale / shift, (x * msc + msh) > 0 with the same single fma the folded loaders use
// (conv_rs.hip INBN, conv_mfma.hip XBN): the normalised tensor and its bit mask never existed.
template <typename T, int MASK, bool DROP>
__device__ __forceinline__ void bn_bwd_grad(const BnBwdElem<T, MASK, DROP>& e, uint32_t i,
                                            float (&g)[VecIO<T>::V], float (&gres)[VecIO<T>::V],
                                            const float* msc = nullptr, const float* msh = nullptr) {
  constexpr int V = VecIO<T>::V;
  VecIO<T>::cvt(e.g, g);
  if constexpr (MASK == 3) {
    float xx[V];
    VecIO<T>::cvt(e.x, xx);
#pragma unroll
    for (int k = 0; k < V; ++k) g[k] = bn_affine(xx[k], msc[k], msh[k]) > 0.f ? g[k] : 0.f;
  }
  if constexpr (MASK == 1) {
    const int sh = (int)(i & 63);
#pragma unroll
    for (int k = 0; k < V / 2; ++k) {
      g[2 * k] = ((e.w[k].x >> sh) & 1ull) ? g[2 * k] : 0.f;
      g[2 * k + 1] = ((e.w[k].y >> sh) & 1ull) ? g[2 * k + 1] : 0.f;
    }
  }
  if constexpr (MASK == 2) {
    float yy[V];
    VecIO<T>::cvt(e.y, yy);
#pragma unroll
    for (int k = 0; k < V; ++k) g[k] = yy[k] > 0.f ? g[k] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < V; ++k) gres[k] = g[k];
  if constexpr (DROP) {
#pragma unroll
    for (int k = 0; k < V; ++k) g[k] = bn_mul(g[k], e.d[k]);
  }
}

template <typename T, int MASK, bool DROP>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_fast_kernel(
    const T* __restrict__ dy, const T* __restrict__ y, const uint64_t* __restrict__ mask_bits,
    const T* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ drop, uint32_t pixels, uint32_t hw, int cvn, int rows_alloc,
    float* __restrict__ partial, const float* __restrict__ mask_scale = nullptr,
    const float* __restrict__ mask_shift = nullptr) {
  constexpr int V = VecIO<T>::V;
  typedef BnBwdElem<T, MASK, DROP> Elem;
  extern __shared__ __attribute__((aligned(16))) float cred[];   // [2][lanes][c]
  const uint32_t rows = gridDim.x;
  const uint32_t chunk = (pixels + rows - 1) / rows;
  const uint32_t p0 = blockIdx.x * chunk, p1 = min(p0 + chunk, pixels);
  const int lanes = kThreads / cvn, c = cvn * V;
  const int cv = threadIdx.x % cvn, rl = threadIdx.x / cvn;
  float a1[V], a2[V];
#pragma unroll
  for (int k = 0; k < V; ++k) a1[k] = a2[k] = 0.f;
  if (rl < lanes) {
    float mu[V], is[V];
    ldf<V>(mean + cv * V, mu);
    ldf<V>(invstd + cv * V, is);
    [[maybe_unused]] float msc[V], msh[V];
    if constexpr (MASK == 3) {
      ldf<V>(mask_scale + cv * V, msc);
      ldf<V>(mask_shift + cv * V, msh);
    }
    auto use = [&](const Elem& e, uint32_t i) {
      float g[V], gres[V], xx[V];
      bn_bwd_grad<T, MASK, DROP>(e, i, g, gres, msc, msh);
      VecIO<T>::cvt(e.x, xx);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        a1[k] = bn_add(a1[k], g[k]);
        a2[k] = bn_acc_xhat(a2[k], g[k], xx[k], mu[k], is[k]);
      }
    };
    uint32_t p = p0 + rl;
    for (; p + lanes < p1; p += 2 * lanes) {
      Elem e0, e1;
      const uint32_t i0 = p * cvn + cv, i1 = (p + lanes) * cvn + cv;
      bn_bwd_issue<T, MASK, DROP>(e0, dy, y, mask_bits, x, drop, i0, p, hw, cvn, cv, true);
      bn_bwd_issue<T, MASK, DROP>(e1, dy, y, mask_bits, x, drop, i1, p + lanes, hw, cvn, cv, true);
      use(e0, i0);
      use(e1, i1);
    }
    if (p < p1) {
      Elem e0;
      const uint32_t i0 = p * cvn + cv;
      bn_bwd_issue<T, MASK, DROP>(e0, dy, y, mask_bits, x, drop, i0, p, hw, cvn, cv, true);
      use(e0, i0);
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      cred[(0 * lanes + rl) * c + cv * V + k] = a1[k];
      cred[(1 * lanes + rl) * c + cv * V + k] = a2[k];
    }
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < 2 * c; ch += blockDim.x) {
    const int which = ch / c, cc = ch % c;
    float a = 0.f;
    for (int k = 0; k < lanes; ++k) a += cred[(which * lanes + k) * c + cc];
    partial[((long)which * rows_alloc + blockIdx.x) * c + cc] = a;
  }
}

// fast form of bn_bwd_apply_kernel: a thread's channel vector is fixed (grid stride = a multiple of
// cvn), its gamma / invstd / mean / sums live in registers
template <typename T, int MASK, bool DROP, bool TRAIN>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_fast_kernel(
    const T* __restrict__ dy, const T* __restrict__ y, const uint64_t* __restrict__ mask_bits,
    const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ invstd, const float* __restrict__ drop,
    const float* __restrict__ partial, int rows, int rows_alloc, float* __restrict__ dbeta_out,
    float* __restrict__ dgamma_out, uint32_t hw, int cvn_log2, uint32_t totalv, float inv_count,
    T* __restrict__ dx, T* __restrict__ dres, const float* __restrict__ mask_scale = nullptr,
    const float* __restrict__ mask_shift = nullptr) {
  constexpr int V = VecIO<T>::V;
  typedef BnBwdElem<T, MASK, DROP> Elem;
  extern __shared__ __attribute__((aligned(16))) float sums[];   // [2][c]
  const uint32_t cvn = 1u << cvn_log2;
  const int c = (int)cvn * V;
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int sl = 0; sl < kBwdSlices; ++sl) {
      a1 += (double)partial[((long)0 * rows_alloc + rows + sl) * c + ch];
      a2 += (double)partial[((long)1 * rows_alloc + rows + sl) * c + ch];
    }
    sums[ch] = (float)a1;
    sums[c + ch] = (float)a2;
    if (blockIdx.x == 0) {
      dbeta_out[ch] = (float)a1;
      dgamma_out[ch] = (float)a2;
    }
  }
  __syncthreads();
  const uint32_t stride = gridDim.x * kThreads;
  uint32_t i = blockIdx.x * kThreads + threadIdx.x;
  const uint32_t cv = i & (cvn - 1), dpix = stride >> cvn_log2;
  uint32_t pix = i >> cvn_log2;
  float ga[V], is[V], mu[V], db[V], dg[V];
  ldf<V>(gamma + cv * V, ga);
  ldf<V>(invstd + cv * V, is);
  if constexpr (TRAIN) {
    ldf<V>(mean + cv * V, mu);
    ldf<V>(sums + cv * V, db);
    ldf<V>(sums + c + cv * V, dg);
  }
  [[maybe_unused]] float msc[V], msh[V];
  if constexpr (MASK == 3) {
    ldf<V>(mask_scale + cv * V, msc);
    ldf<V>(mask_shift + cv * V, msh);
  }
  auto use = [&](const Elem& e, uint32_t idx) {
    float g[V], gres[V], o[V];
    bn_bwd_grad<T, MASK, DROP>(e, idx, g, gres, msc, msh);
    if (dres) VecIO<T>::store(dres + (size_t)idx * V, gres);
    if constexpr (TRAIN) {
      float xx[V];
      VecIO<T>::cvt(e.x, xx);
#pragma unroll
      for (int k = 0; k < V; ++k)
        o[k] = bn_dx_train(g[k], xx[k], mu[k], is[k], bn_mul(ga[k], is[k]), db[k], dg[k], inv_count);
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] = bn_mul(bn_mul(g[k], ga[k]), is[k]);
    }
    VecIO<T>::store(dx + (size_t)idx * V, o);
  };
  for (; i + stride < totalv; i += 2 * stride, pix += 2 * dpix) {
    Elem e0, e1;
    bn_bwd_issue<T, MASK, DROP>(e0, dy, y, mask_bits, x, drop, i, pix, hw, cvn, cv, TRAIN);
    bn_bwd_issue<T, MASK, DROP>(e1, dy, y, mask_bits, x, drop, i + stride, pix + dpix, hw, cvn, cv, TRAIN);
    use(e0, i);
    use(e1, i + stride);
  }
  if (i < totalv) {
    Elem e0;
    bn_bwd_issue<T, MASK, DROP>(e0, dy, y, mask_bits, x, drop, i, pix, hw, cvn, cv, TRAIN);
    use(e0, i);
  }
}

// state (optional): {base seed, training step} in DEVICE memory; the step's seed is
// base + 0x632BE5AB * step like EMSANet._dropout_seed computes it on the host (a training step
// captured in a hipGraph draws fresh masks at every replay)
__global__ void dropout2d_mask_kernel(float* mask, int n, int c, float p, uint32_t seed,
                                      uint32_t layer_id, const uint32_t* __restrict__ state) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * c) return;
  if (state) seed = state[0] + 0x632BE5ABu * state[1];
  const uint32_t nn = i / c, cc = i % c;
  const uint32_t key = emsa_lowbias32(seed + layer_id * 0x9E3779B1u);
  const uint32_t h = emsa_lowbias32(key + nn * 0x85EBCA77u + cc * 0xC2B2AE3Du);
  const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
  mask[i] = u >= p ? 1.0f / (1.0f - p) : 0.f;
}

// all Dropout2d masks of a training step in ONE launch (50 launches of ~5 us before): job j =
// {offset into `masks` (floats), channels, layer id, p}; blockIdx.y = job
__global__ void dropout2d_mask_batch_kernel(float* __restrict__ masks,
                                            const EmsaDropoutJob* __restrict__ jobs, int n,
                                            uint32_t seed, const uint32_t* __restrict__ state) {
  const EmsaDropoutJob jb = jobs[blockIdx.y];
  if (state) seed = state[0] + 0x632BE5ABu * state[1];
  const uint32_t key = emsa_lowbias32(seed + jb.layer_id * 0x9E3779B1u);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n * jb.c; i += gridDim.x * blockDim.x) {
    const uint32_t nn = i / jb.c, cc = i % jb.c;
    const uint32_t h = emsa_lowbias32(key + nn * 0x85EBCA77u + cc * 0xC2B2AE3Du);
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    masks[jb.offset + i] = u >= jb.p ? 1.0f / (1.0f - jb.p) : 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// max pool 3x3 s2 p1
// ------------------------------------------------------------------------------------------
// I: index type (uint32_t when every element index fits 31 bits -- idx32_ok: a 64-bit division costs
// ~120 instructions, a 32-bit one ~25, and these loops decompose their index per element)
template <typename T, typename I = long>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                   int8_t* __restrict__ idx, int n, int h, int w, int c4n) {
  const int oh_n = (h + 1) / 2, ow_n = (w + 1) / 2;
  const I total = (I)n * oh_n * ow_n * c4n;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total;
       i += (I)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % (I)c4n);
    I r = i / (I)c4n;
    const int ow = (int)(r % (I)ow_n); r /= (I)ow_n;
    const int oh = (int)(r % (I)oh_n);
    const int img = (int)(r / (I)oh_n);
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int bi[4] = {0, 0, 0, 0};
    bool first = true;
    // all nine taps are loaded first, at clamped coordinates (a window's out-of-range taps used to be
    // skipped by a branch around the load: nine dependent memory round trips per output)
    float4 tap[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = min(max(oh * 2 - 1 + kh, 0), h - 1), iw = min(max(ow * 2 - 1 + kw, 0), w - 1);
        tap[kh * 3 + kw] = emsa_ld4(x + (size_t)(((((I)img * h + ih) * w + iw) * c4n + c4) * 4));
      }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
        const bool in = ih >= 0 && ih < h && iw >= 0 && iw < w;
        const float4 v = tap[kh * 3 + kw];
        const int t = kh * 3 + kw;
        if (in && (first || v.x > best.x)) { best.x = v.x; bi[0] = t; }
        if (in && (first || v.y > best.y)) { best.y = v.y; bi[1] = t; }
        if (in && (first || v.z > best.z)) { best.z = v.z; bi[2] = t; }
        if (in && (first || v.w > best.w)) { best.w = v.w; bi[3] = t; }
        first = first && !in;
      }
    emsa_st4(y + (size_t)i * 4, best);
    *reinterpret_cast<char4*>(idx + (size_t)i * 4) = make_char4(bi[0], bi[1], bi[2], bi[3]);
  }
}

// One thread = a 2x2 quad of INPUT pixels x V channels (16 bytes): the quad (2qh.., 2qw..) lies
// under exactly the four windows (qh..qh+1, qw..qw+1), so four (dy, argmax) pairs are read for four
// stores -- the pixel-per-thread form read 2.25 windows per pixel (4.5 loads per store) and moved
// 2.4 TB/s.  Window (oh, ow) covers rows 2oh-1 .. 2oh+1: tap index kh = ih - 2oh + 1.
template <typename T, typename I = long>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const int8_t* __restrict__ idx,
                                   T* __restrict__ dx, int n, int h, int w, int cvn) {
  constexpr int V = VecIO<T>::V;
  const int oh_n = (h + 1) / 2, ow_n = (w + 1) / 2;      // also the number of quads per direction
  const I total = (I)n * oh_n * ow_n * cvn;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total;
       i += (I)gridDim.x * blockDim.x) {
    const int cv = (int)(i % (I)cvn);
    I r = i / (I)cvn;
    const int qw = (int)(r % (I)ow_n); r /= (I)ow_n;
    const int qh = (int)(r % (I)oh_n);
    const int img = (int)(r / (I)oh_n);
    float acc[2][2][V];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int k = 0; k < V; ++k) acc[a][b2][k] = 0.f;
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        const int oh = qh + dh, ow = qw + dw;
        if (oh >= oh_n || ow >= ow_n) continue;
        const size_t o = (size_t)(((((I)img * oh_n + oh) * ow_n + ow) * cvn + cv) * V);
        float g[V];
        VecIO<T>::load(dy + o, g);
        int8_t kk[V];
        if constexpr (V == 8) {
          const uint2 raw = *reinterpret_cast<const uint2*>(idx + o);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            kk[k] = (int8_t)((raw.x >> (8 * k)) & 0xFF);
            kk[4 + k] = (int8_t)((raw.y >> (8 * k)) & 0xFF);
          }
        } else {
          const uint32_t raw = *reinterpret_cast<const uint32_t*>(idx + o);
#pragma unroll
          for (int k = 0; k < 4; ++k) kk[k] = (int8_t)((raw >> (8 * k)) & 0xFF);
        }
        // quad pixel (a, b) = input (2qh + a, 2qw + b): kh = a - 2 dh + 1, kw = b - 2 dw + 1
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int kh = a - 2 * dh + 1;
          if (kh < 0 || kh > 2) continue;
#pragma unroll
          for (int b2 = 0; b2 < 2; ++b2) {
            const int kw = b2 - 2 * dw + 1;
            if (kw < 0 || kw > 2) continue;
            const int t = kh * 3 + kw;
#pragma unroll
            for (int k = 0; k < V; ++k) acc[a][b2][k] += (kk[k] == t) ? g[k] : 0.f;
          }
        }
      }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int ih = 2 * qh + a, iw = 2 * qw + b2;
        if (ih < h && iw < w)
          VecIO<T>::store(dx + (size_t)(((((I)img * h + ih) * w + iw) * cvn + cv) * V), acc[a][b2]);
      }
  }
}

// ------------------------------------------------------------------------------------------
// squeeze-and-excitation
// ------------------------------------------------------------------------------------------
// stage 1: ws[n][split][c] = sum_{hw chunk} a*b (b may be NULL -> sum a); stage 2 sums the splits
// in a fixed order -> bit-reproducible (no atomics: the SE weighting feeds every later layer, and
// run-to-run noise there flips ReLU masks downstream)
// nblk_a: the first nblk_a workgroups reduce `a`, the ones behind them `a2` (b == NULL): the two
// squeeze inputs of an SE-add fusion in ONE launch (emsa_se_pair_fwd_t)
template <typename T, bool HAS_B>
__global__ void channel_dot_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                   float* __restrict__ ws, long hw, int cvn, int splits,
                                   const T* __restrict__ a2 = nullptr, int nblk_a = 0x7fffffff) {
  // 16 bytes per lane in every storage type (8 channels of bf16 / fp16, 4 of fp32): block =
  // (cvn channel vectors) x (256 / cvn row lanes) over this split's pixels
  constexpr int V = VecIO<T>::V;
  extern __shared__ __attribute__((aligned(16))) float cred[];   // [lanes][c]
  const bool second = (int)blockIdx.x >= nblk_a;
  const int blk = second ? blockIdx.x - nblk_a : blockIdx.x;
  if (second) a = a2;
  const int img = blk / splits, sp = blk % splits;
  const long chunk = (hw + splits - 1) / splits;
  const long p0 = img * hw + sp * chunk, p1 = min(img * hw + (sp + 1) * chunk, (img + 1) * hw);
  const int lanes = kThreads / cvn, c = cvn * V;
  const int cv = threadIdx.x % cvn, rl = threadIdx.x / cvn;
  if (rl < lanes) {
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
    // Four pixels per iteration, their loads issued together (HAS_B is a template flag: with a
    // run-time `if (b)` inside the loop every load was followed by `s_waitcnt vmcnt(0)` whatever the
    // unroll pragma said -- one 16-byte load in flight per thread); the additions keep their order.
    typedef typename VecIO<T>::raw Raw;
    auto term = [&](const Raw& ra, const Raw& rb) {
      float va[V];
      VecIO<T>::cvt(ra, va);
      if constexpr (HAS_B) {
        float vb[V];
        VecIO<T>::cvt(rb, vb);
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += va[k] * vb[k];
      } else {
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] += va[k];
      }
    };
    long p = p0 + rl;
    for (; p + 3 * lanes < p1; p += 4 * lanes) {
      Raw ra[4], rb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = ((p + u * lanes) * cvn + cv) * V;
        ra[u] = VecIO<T>::load_raw(a + i);
        rb[u] = ra[u];
        if constexpr (HAS_B) rb[u] = VecIO<T>::load_raw(b + i);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) term(ra[u], rb[u]);
    }
    for (; p < p1; p += lanes) {
      const long i = (p * cvn + cv) * V;
      const Raw ra = VecIO<T>::load_raw(a + i);
      Raw rb = ra;
      if constexpr (HAS_B) rb = VecIO<T>::load_raw(b + i);
      term(ra, rb);
    }
#pragma unroll
    for (int k = 0; k < V; ++k) cred[rl * c + cv * V + k] = acc[k];
  }
  __syncthreads();
  // (round 2 blamed a `blockDim.x` read in this loop for a hipGraph-replay discrepancy; round 3
  //  reproduced the same wrong loss with either form and traced it to the MEMSET nodes of the
  //  captured graph -- csrc/graph_tools.hip, DESIGN.md 5b.  The compile-time stride stays: it is
  //  the launch configuration anyway.)
  for (int ch = threadIdx.x; ch < c; ch += kThreads) {
    float t = 0.f;
    for (int k = 0; k < lanes; ++k) t += cred[k * c + ch];
    ws[(long)blockIdx.x * c + ch] = t;
  }
}

// channel_dot_kernel for more channel vectors than threads (fp32, C > 1024): grid = (workgroups of the
// narrow form, channel-vector groups), one thread per channel vector, pixels in order
template <typename T, bool HAS_B>
__global__ void channel_dot_wide_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                        float* __restrict__ ws, long hw, int cvn, int splits,
                                        const T* __restrict__ a2 = nullptr, int nblk_a = 0x7fffffff) {
  constexpr int V = VecIO<T>::V;
  const int cv = blockIdx.y * kThreads + threadIdx.x;
  if (cv >= cvn) return;
  const bool second = (int)blockIdx.x >= nblk_a;
  const int blk = second ? blockIdx.x - nblk_a : blockIdx.x;
  if (second) a = a2;
  const int img = blk / splits, sp = blk % splits;
  const long chunk = (hw + splits - 1) / splits;
  const long p0 = img * hw + sp * chunk, p1 = min(img * hw + (sp + 1) * chunk, (img + 1) * hw);
  float acc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0.f;
  for (long p = p0; p < p1; ++p) {
    const long i = (p * cvn + cv) * V;
    float va[V];
    VecIO<T>::load(a + i, va);
    if constexpr (HAS_B) {
      float vb[V];
      VecIO<T>::load(b + i, vb);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += va[k] * vb[k];
    } else {
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += va[k];
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) ws[(long)blockIdx.x * cvn * V + cv * V + k] = acc[k];
}

__global__ void channel_dot_finish_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                          int n, int c, int splits, float scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * c) return;
  const int img = i / c, ch = i % c;
  float a = 0.f;
#pragma unroll 8
  for (int sp = 0; sp < splits; ++sp) a += ws[((long)img * splits + sp) * c + ch];
  out[i] = a * scale;
}

// hidden layer of the excitation MLP: hsh[r] = relu(b1[r] + w1[r][:] . g[:]), all hidden units at once:
// G = 256 / cr (<= 64) neighbouring lanes share a unit, lane j of a group sums the channels j, j + G, ...
// (coalesced pieces of the row of w1), then a fixed butterfly over the group -- one THREAD per unit
// walked its row alone (c dependent steps: 7-18 us of a batch-1 graph whose other nodes take 5).
// Same order in every caller (single / paired forward): bit-identical results.  blockDim.x = 256.
__device__ __forceinline__ void se_hidden_layer(const float* __restrict__ w1,
                                                const float* __restrict__ b1, const float* g,
                                                float* hsh, float* __restrict__ hid_row, int c,
                                                int cr) {
  int G = 64;
  while (G > 1 && G * cr > 256) G >>= 1;          // lanes per hidden unit (power of two)
  const int per_pass = 256 / G;
  const int j = threadIdx.x % G, u = threadIdx.x / G;
  for (int r0 = 0; r0 < cr; r0 += per_pass) {
    const int r = r0 + u;
    float a = 0.f;
    if (r < cr) {
      const float* row = w1 + (long)r * c;
#pragma unroll 8
      for (int k = j; k < c; k += G) a += row[k] * g[k];
    }
    for (int o = G >> 1; o >= 1; o >>= 1) a += __shfl_xor(a, o);
    if (j == 0 && r < cr) {
      a = fmaxf(a + b1[r], 0.f);
      hsh[r] = a;
      hid_row[r] = a;
    }
  }
}

__global__ void se_mlp_fwd_kernel(const float* __restrict__ gap, const float* __restrict__ w1,
                                  const float* __restrict__ b1, const float* __restrict__ w2,
                                  const float* __restrict__ b2, float* __restrict__ hid,
                                  float* __restrict__ s, int c, int cr) {
  extern __shared__ float sm[];   // [c] gap + [cr] hidden
  float* g = sm;
  float* hsh = sm + c;
  const int img = blockIdx.x;
  for (int i = threadIdx.x; i < c; i += blockDim.x) g[i] = gap[(long)img * c + i];
  __syncthreads();
  se_hidden_layer(w1, b1, g, hsh, hid + (long)img * cr, c, cr);
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    float a = b2[i];
#pragma unroll 4
    for (int r = 0; r < cr; ++r) a += w2[(long)i * cr + r] * hsh[r];
    s[(long)img * c + i] = 1.f / (1.f + expf(-a));
  }
}

// Both excitation MLPs of an SE-add fusion in one launch, fed by the split sums of
// channel_dot_kernel directly (the channel_dot_finish pass in its loader, same summation order):
// workgroup (img, m): gap[m][img][:] = scale * sum_sp ws[m][img][sp][:], then the MLP of modality
// m.  7 launches per fusion -> 3 (the batch-1 inference graph is a chain of launches).
__global__ void se_mlp_pair_fwd_kernel(const float* __restrict__ ws, int splits, float scale,
                                       const float* __restrict__ w1a, const float* __restrict__ b1a,
                                       const float* __restrict__ w2a, const float* __restrict__ b2a,
                                       const float* __restrict__ w1b, const float* __restrict__ b1b,
                                       const float* __restrict__ w2b, const float* __restrict__ b2b,
                                       float* __restrict__ gap, float* __restrict__ hid,
                                       float* __restrict__ s, int n, int c, int cr) {
  extern __shared__ float sm[];   // [c] gap + [cr] hidden
  float* g = sm;
  float* hsh = sm + c;
  const int img = blockIdx.x % n, m = blockIdx.x / n;
  const float* w1 = m ? w1b : w1a; const float* b1 = m ? b1b : b1a;
  const float* w2 = m ? w2b : w2a; const float* b2 = m ? b2b : b2a;
  const long row = (long)m * n + img;
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    float a = 0.f;
#pragma unroll 8
    for (int sp = 0; sp < splits; ++sp) a += ws[(row * splits + sp) * c + i];
    a *= scale;
    g[i] = a;
    gap[row * c + i] = a;
  }
  __syncthreads();
  se_hidden_layer(w1, b1, g, hsh, hid + row * cr, c, cr);
  __syncthreads();
  for (int i = threadIdx.x; i < c; i += blockDim.x) {
    float a = b2[i];
#pragma unroll 4
    for (int r = 0; r < cr; ++r) a += w2[(long)i * cr + r] * hsh[r];
    s[row * c + i] = 1.f / (1.f + expf(-a));
  }
}

// grid = 8 blocks; every block recomputes dz1 [n][cr] in LDS, then owns 1/8 of the outputs
// SE MLP backward in three small launches instead of one 8-workgroup kernel in which every
// workgroup recomputed dz1 with a serial loop over the channels (108 us average, 390 us at 512
// channels; ten launches per training step):
//   A (one workgroup per image): g = ds*s*(1-s);  dz1 = relu'(hid) * (g . w2);  dgap = dz1 . w1
//   B (grid over outputs): dw1 = dz1^T gap, db1      C: dw2 = g^T hid, db2
// dz1 [n][cr] travels in the dw2 buffer (c*cr >= n*cr floats), which C overwrites last.
__global__ void se_mlp_bwd_a_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                    const float* __restrict__ hid, const float* __restrict__ s,
                                    const float* __restrict__ ds, float* __restrict__ dgap,
                                    float* __restrict__ dz1_out, int c, int cr) {
  extern __shared__ float sm[];          // g[c], part[groups][cr], dz1[cr]
  float* g = sm;
  const int groups = blockDim.x / cr;    // k groups per hidden unit (cr <= 32)
  float* part = sm + c;
  float* dz1 = part + groups * cr;
  const int img = blockIdx.x, tid = threadIdx.x;
  for (int k = tid; k < c; k += blockDim.x) {
    const float sv = s[(long)img * c + k];
    g[k] = ds[(long)img * c + k] * sv * (1.f - sv);
  }
  __syncthreads();
  const int r = tid % cr, kg = tid / cr;
  if (kg < groups) {
    float a = 0.f;
    for (int k = kg; k < c; k += groups) a += g[k] * w2[(long)k * cr + r];
    part[kg * cr + r] = a;
  }
  __syncthreads();
  if (tid < cr) {
    float a = 0.f;
    for (int q = 0; q < groups; ++q) a += part[q * cr + tid];
    a = hid[(long)img * cr + tid] > 0.f ? a : 0.f;
    dz1[tid] = a;
    dz1_out[(long)img * cr + tid] = a;
  }
  __syncthreads();
  for (int k = tid; k < c; k += blockDim.x) {
    float a = 0.f;
    for (int q = 0; q < cr; ++q) a += dz1[q] * w1[(long)q * c + k];
    dgap[(long)img * c + k] = a;
  }
}
// MODE 0: dw1 [cr][c] = sum_img dz1[img][r] gap[img][k], db1;  MODE 1: dw2 [c][cr] = sum_img
// g[img][k] hid[img][r], db2.  One thread per output element, the workgroups behind them the bias.
template <int MODE>
__global__ void se_mlp_bwd_w_kernel(const float* __restrict__ gap, const float* __restrict__ hid,
                                    const float* __restrict__ s, const float* __restrict__ ds,
                                    const float* __restrict__ dz1, float* __restrict__ dw,
                                    float* __restrict__ db, int n, int c, int cr, int wblocks) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < wblocks) {
    const int i = blockIdx.x * blockDim.x + tid;
    if (i >= c * cr) return;
    float a = 0.f;
    if (MODE == 0) {
      const int r = i / c, k = i % c;
      for (int img = 0; img < n; ++img) a += dz1[(long)img * cr + r] * gap[(long)img * c + k];
    } else {
      const int k = i / cr, r = i % cr;
      for (int img = 0; img < n; ++img) {
        const float sv = s[(long)img * c + k];
        a += ds[(long)img * c + k] * sv * (1.f - sv) * hid[(long)img * cr + r];
      }
    }
    dw[i] = a;
  } else {
    const int i = (blockIdx.x - wblocks) * blockDim.x + tid;
    if (i >= (MODE == 0 ? cr : c)) return;
    float a = 0.f;
    for (int img = 0; img < n; ++img) {
      if (MODE == 0) {
        a += dz1[(long)img * cr + i];
      } else {
        const float sv = s[(long)img * c + i];
        a += ds[(long)img * c + i] * sv * (1.f - sv);
      }
    }
    db[i] = a;
  }
}

__global__ void se_mlp_bwd_kernel(const float* __restrict__ gap, const float* __restrict__ w1,
                                  const float* __restrict__ w2, const float* __restrict__ hid,
                                  const float* __restrict__ s, const float* __restrict__ ds,
                                  float* __restrict__ dgap, float* __restrict__ dw1,
                                  float* __restrict__ db1, float* __restrict__ dw2,
                                  float* __restrict__ db2, int n, int c, int cr) {
  extern __shared__ float dz1[];   // [n][cr]
  const int tid = threadIdx.x, nb = gridDim.x, b = blockIdx.x;
  for (int i = tid; i < n * cr; i += blockDim.x) {
    const int img = i / cr, r = i % cr;
    float a = 0.f;
    for (int k = 0; k < c; ++k) {
      const float sv = s[(long)img * c + k];
      a += ds[(long)img * c + k] * sv * (1.f - sv) * w2[(long)k * cr + r];
    }
    dz1[i] = hid[i] > 0.f ? a : 0.f;
  }
  __syncthreads();
  // dw2 [c][cr], db2 [c]
  for (int i = b * blockDim.x + tid; i < c * cr; i += nb * blockDim.x) {
    const int k = i / cr, r = i % cr;
    float a = 0.f;
    for (int img = 0; img < n; ++img) {
      const float sv = s[(long)img * c + k];
      a += ds[(long)img * c + k] * sv * (1.f - sv) * hid[(long)img * cr + r];
    }
    dw2[i] = a;
  }
  for (int k = b * blockDim.x + tid; k < c; k += nb * blockDim.x) {
    float a = 0.f;
    for (int img = 0; img < n; ++img) {
      const float sv = s[(long)img * c + k];
      a += ds[(long)img * c + k] * sv * (1.f - sv);
    }
    db2[k] = a;
  }
  // dw1 [cr][c], db1 [cr]
  for (int i = b * blockDim.x + tid; i < cr * c; i += nb * blockDim.x) {
    const int r = i / c, k = i % c;
    float a = 0.f;
    for (int img = 0; img < n; ++img) a += dz1[img * cr + r] * gap[(long)img * c + k];
    dw1[i] = a;
  }
  for (int r = b * blockDim.x + tid; r < cr; r += nb * blockDim.x) {
    float a = 0.f;
    for (int img = 0; img < n; ++img) a += dz1[img * cr + r];
    db1[r] = a;
  }
  // dgap [n][c]
  for (int i = b * blockDim.x + tid; i < n * c; i += nb * blockDim.x) {
    const int img = i / c, k = i % c;
    float a = 0.f;
    for (int r = 0; r < cr; ++r) a += dz1[img * cr + r] * w1[(long)r * c + k];
    dgap[i] = a;
  }
}

// HAS_B / HAS_E: the optional second operand is a template parameter -- all loads of an element are
// issued before the first is used (as a run-time `if (b)` the loads sat behind a branch and the
// element took two memory round trips); I: index type, see maxpool_fwd_kernel
template <typename T, bool HAS_B, typename I>
__global__ void se_scale_add_fwd_kernel(const T* __restrict__ a, const float* __restrict__ sa,
                                        const T* __restrict__ b, const float* __restrict__ sb,
                                        T* __restrict__ out, I hw, int cvn, I totalv) {
  constexpr int V = VecIO<T>::V;
  typedef typename VecIO<T>::raw Raw;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < totalv;
       i += (I)gridDim.x * blockDim.x) {
    const I pix = i / (I)cvn;
    const int cv = (int)(i - pix * (I)cvn);
    const I img = pix / hw;
    const size_t so = (size_t)(img * (I)cvn + cv) * V;
    const Raw ra = VecIO<T>::load_raw(a + (size_t)i * V);
    Raw rb = ra;
    if constexpr (HAS_B) rb = VecIO<T>::load_raw(b + (size_t)i * V);
    float va[V], ka[V], o[V];
    ldf<V>(sa + so, ka);
    float kb[V];
    if constexpr (HAS_B) ldf<V>(sb + so, kb);
    VecIO<T>::cvt(ra, va);
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = va[k] * ka[k];
    if constexpr (HAS_B) {
      float vb[V];
      VecIO<T>::cvt(rb, vb);
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] += vb[k] * kb[k];
    }
    VecIO<T>::store(out + (size_t)i * V, o);
  }
}

template <typename T, bool HAS_E, typename I>
__global__ void se_scale_bwd_apply_kernel(const T* __restrict__ dout,
                                          const float* __restrict__ s,
                                          const float* __restrict__ dgap,
                                          const T* __restrict__ extra, T* __restrict__ dx,
                                          I hw, int cvn, I totalv, float inv_hw) {
  constexpr int V = VecIO<T>::V;
  typedef typename VecIO<T>::raw Raw;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < totalv;
       i += (I)gridDim.x * blockDim.x) {
    const I pix = i / (I)cvn;
    const int cv = (int)(i - pix * (I)cvn);
    const I img = pix / hw;
    const size_t so = (size_t)(img * (I)cvn + cv) * V;
    const Raw rg = VecIO<T>::load_raw(dout + (size_t)i * V);
    Raw re = rg;
    if constexpr (HAS_E) re = VecIO<T>::load_raw(extra + (size_t)i * V);
    float g[V], k_[V], dg[V], o[V];
    ldf<V>(s + so, k_);
    ldf<V>(dgap + so, dg);
    VecIO<T>::cvt(rg, g);
#pragma unroll
    for (int k = 0; k < V; ++k) o[k] = g[k] * k_[k] + dg[k] * inv_hw;
    if constexpr (HAS_E) {
      float e[V];
      VecIO<T>::cvt(re, e);
#pragma unroll
      for (int k = 0; k < V; ++k) o[k] += e[k];
    }
    VecIO<T>::store(dx + (size_t)i * V, o);
  }
}

// ------------------------------------------------------------------------------------------
// nearest x2 + depth-wise 3x3 (zero pad)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 dw_weight4(const float* __restrict__ w, int c, int t) {
  return make_float4(w[(c + 0) * 9 + t], w[(c + 1) * 9 + t], w[(c + 2) * 9 + t],
                     w[(c + 3) * 9 + t]);
}

// One thread = (input pixel (ih,iw), 4 channels): it owns the 2x2 output quad (2ih+a, 2iw+b).
// Output (2ih+a, 2iw+b) reads up-sampled rows 2ih+a-1 .. 2ih+a+1, i.e. input rows
// ih-1+ (a+kh)/2 ... : with r = (a + kh + 1) >> 1 in {0,1,2} selecting input row ih-1+r.  So the
// quad needs the 3x3 input neighbourhood once (9 float4 loads for 4 outputs) and per output the
// 9 taps collapse onto 2x2 / 2x3 / 3x2 / ... neighbourhood entries.
// T = storage type of x / skip, TO = storage type of y (the last up-sampling of a head writes the
// model's fp32 output straight from 16-bit features)
// NT: the output goes out with non-temporal stores (a pure write stream of up to 1.57 GB that no
// later kernel finds in a cache anyway)
// (round 5: SKIP is a template parameter and every load of an element -- the 3x3 input neighbourhood
//  at CLAMPED coordinates, zeroed by a select, and the four skip values -- is issued before the first
//  is used.  The border tests used to be branches around the loads: nine `s_waitcnt vmcnt(0)` in a row,
//  then four more behind the stores -- 13 serialised memory round trips per thread and iteration.)
template <typename T, typename TO, bool NT = false, typename I = long, bool SKIP = false>
__global__ __launch_bounds__(kThreads) void up2x_dw_fwd_kernel(const T* __restrict__ x, const float* __restrict__ wdw,
                                   const float* __restrict__ bias, const T* __restrict__ skip,
                                   TO* __restrict__ y, int n, int h, int w, int c4n,
                                   const T* __restrict__ x2 = nullptr,
                                   const float* __restrict__ wdw2 = nullptr,
                                   const float* __restrict__ bias2 = nullptr,
                                   const T* __restrict__ skip2 = nullptr,
                                   TO* __restrict__ y2 = nullptr) {
  // twin launch (emsa_up2x_dw3x3_fwd_pair_t, grid.y = 2): the second half of the workgroups runs the
  // same shape on the "2" tensors (the other decoder of the model)
  if (blockIdx.y != 0) {
    x = x2; wdw = wdw2; bias = bias2; skip = skip2; y = y2;
  }
  // the depth-wise weights [c][9], transposed to [9][c] in LDS once per workgroup: 9 ds_read_b128
  // per thread instead of 36 strided scalar global loads (the kernel was load-instruction bound)
  extern __shared__ __attribute__((aligned(16))) float wl[];
  const int C = c4n * 4;
  for (int j = threadIdx.x; j < 9 * C; j += blockDim.x) wl[(j % 9) * C + j / 9] = wdw[j];
  __syncthreads();
  const I total = (I)n * h * w * c4n;
  const int ow_n = 2 * w;
  for (I i = blockIdx.x * (I)blockDim.x + threadIdx.x; i < total;
       i += (I)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % (I)c4n);
    I r = i / (I)c4n;
    const int iw = (int)(r % (I)w); r /= (I)w;
    const int ih = (int)(r % (I)h);
    const int img = (int)(r / (I)h);
    float4 nb[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int hh = ih - 1 + a, ww = iw - 1 + b;
        const int hc = min(max(hh, 0), h - 1), wc_ = min(max(ww, 0), w - 1);
        nb[a][b] = emsa_ld4(x + (size_t)(((((I)img * h + hc) * w + wc_) * c4n + c4) * 4));
      }
    float4 sk[2][2];
    if constexpr (SKIP) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          sk[a][b] = emsa_ld4(skip + (size_t)(((((I)img * 2 * h + 2 * ih + a) * ow_n + 2 * iw + b) * c4n + c4) * 4));
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int hh = ih - 1 + a, ww = iw - 1 + b;
        const bool in = hh >= 0 && hh < h && ww >= 0 && ww < w;
        nb[a][b].x = in ? nb[a][b].x : 0.f; nb[a][b].y = in ? nb[a][b].y : 0.f;
        nb[a][b].z = in ? nb[a][b].z : 0.f; nb[a][b].w = in ? nb[a][b].w : 0.f;
      }
    float4 k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = emsa_ld4(wl + t * C + c4 * 4);
    const float4 bv = bias ? emsa_ld4(bias + c4 * 4) : emsa_zero4();
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float4 acc = bv;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            // up-sampled coordinate 2ih+a+kh-1 -> input row ih-1 + ((a+kh+1)>>1); rows outside
            // the UP-SAMPLED image (zero padding) are exactly the nb entries already zeroed,
            // except that up-row -1 / 2h map to input rows -1 / h which are zero as well
            const float4 v = nb[(a + kh + 1) >> 1][(b + kw + 1) >> 1];
            const float4 kk = k[kh * 3 + kw];
            acc.x += v.x * kk.x; acc.y += v.y * kk.y; acc.z += v.z * kk.z; acc.w += v.w * kk.w;
          }
        const size_t o = (size_t)(((((I)img * 2 * h + 2 * ih + a) * ow_n + 2 * iw + b) * c4n + c4) * 4);
        if constexpr (SKIP) {
          acc.x += sk[a][b].x; acc.y += sk[a][b].y; acc.z += sk[a][b].z; acc.w += sk[a][b].w;
        }
        if constexpr (NT && std::is_same<TO, float>::value) {
          const f32x4 v = {acc.x, acc.y, acc.z, acc.w};
          __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(y + o));
        } else {
          emsa_st4(y + o, acc);
        }
      }
  }
}

// (Round 3: a ROW-TILED output-centric form -- three input rows of a 62-column segment staged in
//  LDS, 2x2-collapsed taps from LDS, consecutive lanes writing consecutive 16 bytes of the output
//  row -- was built for the wide maps (c <= 64) and measured SLOWER than the quad kernel above on
//  every shape: 40 channels 240x320 -> 480x640 909 vs 704 us, 64 channels + skip 519 vs 277 us,
//  8 channels 164 vs 126 us (tools/pointwise_bench.py, profiles/r03_d_pointwise_bench.txt): eight
//  ds_read_b128 per 16 output bytes and 46,000 three-barrier workgroups cost more than the
//  half-covered store sectors do.  Removed.)
// (An output-centric form -- one thread per OUTPUT pixel, consecutive lanes writing consecutive
//  bytes, per-parity collapsed 2x2 taps -- was built to get rid of the half-covered sectors this
//  kernel's stores leave behind at 40 channels; measured it is 0.2-0.4 % SLOWER per training step
//  in both storage types: 16 L1 loads per four outputs instead of 9 cost what the stores gained.)
// dx(ih,iw) = sum over the 4x4 output neighbourhood (2ih-1 .. 2ih+2) of dy * (collapsed taps):
// output (oh,ow) touches input row ih through taps kh with ((oh+kh-1)>>1) == ih.
template <typename T, typename TO>
__global__ void up2x_dw_bwd_data_kernel(const TO* __restrict__ dy,
                                        const float* __restrict__ wdw, T* __restrict__ dx,
                                        int n, int h, int w, int c4n) {
  // collapsed taps per (p, q) of the 4x4 output neighbourhood, [16][c] in LDS once per workgroup:
  // taps kh with oh+kh-1 in {2ih, 2ih+1}  <=>  kh in {2-p, 3-p} intersect [0,2] (same for kw, q)
  extern __shared__ __attribute__((aligned(16))) float wc[];
  const int C = c4n * 4;
  for (int j = threadIdx.x; j < 16 * C; j += blockDim.x) {
    const int c = j % C, pq = j / C, pp = pq >> 2, qq = pq & 3;
    float a = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
      if (kh != 2 - pp && kh != 3 - pp) continue;
      for (int kw = 0; kw < 3; ++kw) {
        if (kw != 2 - qq && kw != 3 - qq) continue;
        a += wdw[c * 9 + kh * 3 + kw];
      }
    }
    wc[pq * C + c] = a;
  }
  __syncthreads();
  const int oh_n = 2 * h, ow_n = 2 * w;
  const long total = (long)n * h * w * c4n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % c4n);
    long r = i / c4n;
    const int iw = (int)(r % w); r /= w;
    const int ih = (int)(r % h);
    const int img = (int)(r / h);
    float4 a = emsa_zero4();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int oh = 2 * ih - 1 + p;
      if (oh < 0 || oh >= oh_n) continue;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ow = 2 * iw - 1 + q;
        if (ow < 0 || ow >= ow_n) continue;
        const float4 g = emsa_ld4(dy + ((((long)img * oh_n + oh) * ow_n + ow) * c4n + c4) * 4);
        const float4 ws = emsa_ld4(wc + (p * 4 + q) * C + c4 * 4);
        a.x += g.x * ws.x; a.y += g.y * ws.y; a.z += g.z * ws.z; a.w += g.w * ws.w;
      }
    }
    emsa_st4(dx + i * 4, a);
  }
}

// dw[c][9] += sum dy*up(x) ; db[c] += sum dy.  block = c4n columns x lanes over a chunk of INPUT
// pixels; one thread visits an input pixel, loads its 3x3 neighbourhood once and the 2x2 output
// quad's dy, and accumulates all 9 taps of the 4 outputs (13 loads per 36x4 FMAs).
template <typename T, typename TO>
__global__ void up2x_dw_bwd_weight_kernel(const TO* __restrict__ dy, const T* __restrict__ x,
                                          float* __restrict__ dwt, float* __restrict__ db, int n,
                                          int h, int w, int c4n, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float wred[];   // [lanes][10][c4n*4]
  const int oh_n = 2 * h, ow_n = 2 * w;
  const long pixels = (long)n * h * w;
  const long chunk = (pixels + nblocks - 1) / nblocks;
  const long p0 = blockIdx.x * chunk, p1 = min(p0 + chunk, pixels);
  const int lanes = kThreads / c4n;
  const int c4 = threadIdx.x % c4n, rl = threadIdx.x / c4n;
  float4 acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = emsa_zero4();
  if (rl < lanes) {
    for (long p = p0 + rl; p < p1; p += lanes) {
      const int iw = (int)(p % w);
      const long r = p / w;
      const int ih = (int)(r % h);
      const int img = (int)(r / h);
      float4 nb[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const int hh = ih - 1 + a, ww = iw - 1 + b;
          nb[a][b] = (hh >= 0 && hh < h && ww >= 0 && ww < w)
                         ? emsa_ld4(x + ((((long)img * h + hh) * w + ww) * c4n + c4) * 4)
                         : emsa_zero4();
        }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          const float4 g = emsa_ld4(
              dy + ((((long)img * oh_n + 2 * ih + a) * ow_n + 2 * iw + b) * c4n + c4) * 4);
          acc[9].x += g.x; acc[9].y += g.y; acc[9].z += g.z; acc[9].w += g.w;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              const float4 v = nb[(a + kh + 1) >> 1][(b + kw + 1) >> 1];
              float4& t = acc[kh * 3 + kw];
              t.x += g.x * v.x; t.y += g.y * v.y; t.z += g.z * v.z; t.w += g.w * v.w;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) emsa_st4(wred + ((rl * 10 + t) * c4n + c4) * 4, acc[t]);
  }
  __syncthreads();
  const int c = c4n * 4;
  for (int o = threadIdx.x; o < 10 * c; o += blockDim.x) {
    const int t = o / c, ch = o % c;
    float a = 0.f;
    for (int k = 0; k < lanes; ++k) a += wred[(k * 10 + t) * c + ch];
    if (t < 9)
      unsafeAtomicAdd(dwt + ch * 9 + t, a);
    else
      unsafeAtomicAdd(db + ch, a);
  }
}

// dx, dw and db in ONE pass over dy (the backward of the learned up-sampling reads a tensor four
// times the size of everything else it touches; the two kernels above each stream it, and each
// thread of them re-reads its 4x4 / 3x3 neighbourhood through the vector L1 at 16 loads per 16
// bytes of result: 2.0-2.4 TB/s).  Here a workgroup stages a tile -- kUpTH x TW input pixels, i.e.
// the (2 kUpTH + 2) x (2 TW + 2) patch of dy and the (kUpTH + 2) x (TW + 2) patch of x, CC channels
// deep -- in LDS (every element leaves global memory once per tile) and all neighbourhood
// re-reads are LDS reads.  A workgroup is bound to ONE channel chunk and walks its share of the
// spatial tiles, so the ten weight / bias gradient accumulators stay in registers over the whole
// launch and are reduced once at the end.
//   * Staging is LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no staging
//     registers) into TWO tile buffers: the next tile's loads are in flight while this one is
//     consumed, one barrier per tile.  The first version staged through registers: with the 40
//     accumulators that made ~200 VGPRs = 2 workgroups per CU and no overlap inside a workgroup
//     (1.2 ms for the 1.6 GB gradient of the semantic map, barely better than the two-pass form).
//     The tiles are kept in their storage type; pixels outside the image are fetched from a
//     16-byte zero page (a flat DMA has no out-of-range-reads-zero).
//   * one thread = (input pixel of the tile, 4 channels): the 4x4 dy neighbourhood is streamed
//     against the collapsed taps for dx; the pixel's own 2x2 output quad stays in registers and
//     meets the streamed 3x3 x neighbourhood for the 9 taps.
//   * CC and TW are template parameters: with run-time strides the ~45 LDS addresses of a thread
//     are loop invariants the compiler keeps in registers across the tile loop.
constexpr int kUpTH = 4;
__device__ const uint4 kUpZeroPage = {0u, 0u, 0u, 0u};
// tile buffers of up2x_dw_bwd_fused_kernel: a third one (two tiles of loads in flight per workgroup)
// where buffers + collapsed taps of TWO workgroups still fit a CU's 160 KB
constexpr int up2x_nbuf(int stage_bytes, int cc) {
  return 3 * stage_bytes + 16 * cc * 4 <= 80 * 1024 ? 3 : 2;
}

template <typename E>
__device__ __forceinline__ float4 up_lds4(const E* p) {       // 4 consecutive channels from LDS
  if constexpr (std::is_same<E, float>::value) {
    return *reinterpret_cast<const float4*>(p);
  } else {
    return emsa_ld4(p);
  }
}
__device__ __forceinline__ void up_dma16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)src,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T, typename TO, int CC, int TW, typename I = long>
__global__ __launch_bounds__(kThreads, 2) void up2x_dw_bwd_fused_kernel(
    const TO* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ wdw,
    T* __restrict__ dx, float* __restrict__ dwt, float* __restrict__ db, int n, int h, int w,
    int C, int bpc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char upsm[];
  const int tid = threadIdx.x;
  constexpr int tpp = CC >> 2;                // threads per input pixel
  constexpr int npx = kUpTH * TW;
  constexpr int DH = 2 * kUpTH + 2, DW = 2 * TW + 2, XH = kUpTH + 2, XW = TW + 2;
  constexpr int kDyBytes = DH * DW * CC * (int)sizeof(TO);
  constexpr int kXBytes = XH * XW * CC * (int)sizeof(T);
  constexpr int kStage = kDyBytes + kXBytes;  // one tile buffer (multiple of 16)
  constexpr int NBUF = up2x_nbuf(kStage, CC); // tile buffers: 3 where two workgroups per CU still fit
  constexpr int dy_upp = CC * (int)sizeof(TO) / 16, x_upp = CC * (int)sizeof(T) / 16;   // 16-B units per pixel
  constexpr int dy_units = DH * DW * dy_upp, x_units = XH * XW * x_upp;
  float* wc = reinterpret_cast<float*>(upsm + NBUF * kStage);  // [16][CC] collapsed taps (data gradient)
  const int chunk = blockIdx.x / bpc, bic = blockIdx.x % bpc;
  const int c0 = chunk * CC;                  // (C is CC or a multiple of it: every chunk is full)
  // (the chunk's 9 * CC weights go through LDS first: summed straight from global memory the up to
  //  four taps of an entry were four dependent round trips -- `s_waitcnt vmcnt(0)` behind every load --
  //  times four entries per thread, ~10 us before the first tile)
  {
    float* wraw = reinterpret_cast<float*>(upsm);              // [CC][9], overlays tile buffer 0
    for (int j = tid; j < 9 * CC; j += kThreads) wraw[j] = wdw[c0 * 9 + j];
    __syncthreads();
    for (int j = tid; j < 16 * CC; j += kThreads) {
      const int c = j % CC, pq = j / CC, pp = pq >> 2, qq = pq & 3;
      float a = 0.f;
      for (int kh = 0; kh < 3; ++kh) {
        if (kh != 2 - pp && kh != 3 - pp) continue;
        for (int kw = 0; kw < 3; ++kw) {
          if (kw != 2 - qq && kw != 3 - qq) continue;
          a += wraw[c * 9 + kh * 3 + kw];
        }
      }
      wc[j] = a;
    }
    __syncthreads();                                           // (the first DMA overwrites wraw)
  }
  const bool active = tid < npx * tpp;
  const int c4 = tid % tpp, pl = tid / tpp, lw = pl % TW, lh = pl / TW;
  float4 acc[10];
#pragma unroll
  for (int t = 0; t < 10; ++t) acc[t] = emsa_zero4();
  const int tiles_w = (w + TW - 1) / TW, tiles_h = (h + kUpTH - 1) / kUpTH;
  // (tile indices are 32-bit: their decomposition is SCALAR code, and as 64-bit divisions it was
  //  ~1,000 scalar instructions per tile and wave -- the CU's one scalar unit bounded the kernel)
  const int ntiles = n * tiles_h * tiles_w;
  const int OH = 2 * h, OW = 2 * w;
  const int wave_u0 = tid & ~63;              // first unit of this wave within a 256-unit round

  // issue the DMA of tile t into buffer `buf`: unit u = (pixel of the patch, 16-byte piece)
  auto issue = [&](int t, int buf) {
    const int r = t / tiles_w;
    const int tw_i = t - r * tiles_w;
    const int img = r / tiles_h, th_i = r - img * tiles_h;
    const int h0 = th_i * kUpTH, w0 = tw_i * TW;
    unsigned char* sdy = upsm + buf * kStage;
    unsigned char* sx = sdy + kDyBytes;
#pragma unroll
    for (int u0 = 0; u0 < dy_units; u0 += kThreads) {
      const int u = u0 + tid;
      if (u < dy_units) {
        const int cu = u % dy_upp, px = u / dy_upp, col = px % DW, row = px / DW;
        const int oh = 2 * h0 - 1 + row, ow = 2 * w0 - 1 + col;
        const bool in = oh >= 0 && oh < OH && ow >= 0 && ow < OW;
        const void* src = in ? (const void*)(reinterpret_cast<const unsigned char*>(
                                                 dy + (size_t)((((I)img * OH + oh) * OW + ow) * C + c0)) +
                                             cu * 16)
                             : (const void*)&kUpZeroPage;
        up_dma16(src, sdy + (u0 + wave_u0) * 16);
      }
    }
#pragma unroll
    for (int u0 = 0; u0 < x_units; u0 += kThreads) {
      const int u = u0 + tid;
      if (u < x_units) {
        const int cu = u % x_upp, px = u / x_upp, col = px % XW, row = px / XW;
        const int hh = h0 - 1 + row, ww = w0 - 1 + col;
        const bool in = hh >= 0 && hh < h && ww >= 0 && ww < w;
        const void* src = in ? (const void*)(reinterpret_cast<const unsigned char*>(
                                                 x + (size_t)((((I)img * h + hh) * w + ww) * C + c0)) +
                                             cu * 16)
                             : (const void*)&kUpZeroPage;
        up_dma16(src, sx + (u0 + wave_u0) * 16);
      }
    }
  };

#pragma unroll
  for (int k = 0; k < NBUF - 1; ++k)
    if (bic + k * bpc < ntiles) issue(bic + k * bpc, k);
  // NBUF = 3: tile t + 1 may still be in flight while tile t is consumed.  vmcnt counts this wave's
  // loads AND stores; loads return in order among themselves, so "at most kInFlight operations
  // outstanding" proves tile t has landed as long as tile t + 1 was issued behind it with at least
  // kInFlight DMA instructions in this wave (the FULL 256-unit rounds of a tile: every wave issues
  // those) -- whatever the dx stores in between do
  constexpr int kInFlight = dy_units / kThreads + x_units / kThreads;
  static_assert(NBUF == 2 || (kInFlight >= 1 && kInFlight <= 15), "counted wait out of range");
  int buf = 0;
  for (int t = bic; t < ntiles; t += bpc, buf = buf + 1 == NBUF ? 0 : buf + 1) {
    // this wave's pieces of tile t have landed; after the barrier everybody's have, and everybody
    // is done reading the buffer of tile t - bpc -> refill it while tile t is consumed
    if (NBUF == 3 && t + bpc < ntiles)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBUF == 3 ? kInFlight : 0) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + (NBUF - 1) * bpc < ntiles)
      issue(t + (NBUF - 1) * bpc, buf + NBUF - 1 >= NBUF ? buf - 1 : buf + NBUF - 1);
    const int r = t / tiles_w;
    const int tw_i = t - r * tiles_w;
    const int img = r / tiles_h, th_i = r - img * tiles_h;
    const int h0 = th_i * kUpTH, w0 = tw_i * TW;
    const TO* dyt = reinterpret_cast<const TO*>(upsm + buf * kStage);
    const T* xt = reinterpret_cast<const T*>(upsm + buf * kStage + kDyBytes);
    if (active) {
      // data gradient: the 4x4 dy neighbourhood against the collapsed taps, streamed
      float4 d = emsa_zero4();
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 g = up_lds4(dyt + ((2 * lh + p) * DW + 2 * lw + q) * CC + c4 * 4);
          const float4 ws = *reinterpret_cast<const float4*>(wc + (p * 4 + q) * CC + c4 * 4);
          d.x += g.x * ws.x; d.y += g.y * ws.y; d.z += g.z * ws.z; d.w += g.w * ws.w;
        }
      const int ih = h0 + lh, iw = w0 + lw;
      if (dx != nullptr && ih < h && iw < w)
        emsa_st4(dx + (size_t)((((I)img * h + ih) * w + iw) * C + c0 + c4 * 4), d);
      // weight gradient: the pixel's own 2x2 output quad stays in registers, the 3x3 x
      // neighbourhood is streamed (entry (r, s) meets the taps with ((a+kh+1)>>1, (b+kw+1)>>1) ==
      // (r, s); all conditions fold at compile time)
      float4 g[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
          g[a][b2] = up_lds4(dyt + ((2 * lh + 1 + a) * DW + 2 * lw + 1 + b2) * CC + c4 * 4);
          acc[9].x += g[a][b2].x; acc[9].y += g[a][b2].y;
          acc[9].z += g[a][b2].z; acc[9].w += g[a][b2].w;
        }
#pragma unroll
      for (int r2 = 0; r2 < 3; ++r2)
#pragma unroll
        for (int s2 = 0; s2 < 3; ++s2) {
          const float4 v = up_lds4(xt + ((lh + r2) * XW + lw + s2) * CC + c4 * 4);
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
              if (((a + kh + 1) >> 1) != r2) continue;
#pragma unroll
              for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                  if (((b2 + kw + 1) >> 1) != s2) continue;
                  float4& tt = acc[kh * 3 + kw];
                  const float4 gg = g[a][b2];
                  tt.x += gg.x * v.x; tt.y += gg.y * v.y; tt.z += gg.z * v.z; tt.w += gg.w * v.w;
                }
            }
        }
    }
  }
  // the accumulators of the npx pixel lanes -> one sum per (tap, channel), five taps per round so
  // that the scratch fits the tile area
  float* wred = reinterpret_cast<float*>(upsm);          // [npx][5][CC]
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    if (active)
#pragma unroll
      for (int t = 0; t < 5; ++t)
        *reinterpret_cast<float4*>(wred + ((pl * 5 + t) * CC) + c4 * 4) = acc[half * 5 + t];
    __syncthreads();
    for (int o = tid; o < 5 * CC; o += kThreads) {
      const int t = o / CC, ch = o % CC;
      float a = 0.f;
      for (int k = 0; k < npx; ++k) a += wred[(k * 5 + t) * CC + ch];
      const int tap = half * 5 + t;
      if (tap < 9)
        unsafeAtomicAdd(dwt + (c0 + ch) * 9 + tap, a);
      else
        unsafeAtomicAdd(db + c0 + ch, a);
    }
  }
}

// ------------------------------------------------------------------------------------------
// pyramid pooling
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void bin_range(int i, int size, int bins, int& b0, int& b1) {
  b0 = (i * size) / bins;
  b1 = ((i + 1) * size + bins - 1) / bins;
}

template <typename T>
__global__ void adaptive_avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                            int n, int h, int w, int c, int bins) {
  const long total = (long)n * bins * bins * c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long r = i / c;
    const int bj = (int)(r % bins); r /= bins;
    const int bi = (int)(r % bins);
    const int img = (int)(r / bins);
    int h0, h1, w0, w1;
    bin_range(bi, h, bins, h0, h1);
    bin_range(bj, w, bins, w0, w1);
    float a = 0.f;
    for (int hh = h0; hh < h1; ++hh)
      for (int ww = w0; ww < w1; ++ww) a += emsa_ld1(x + (((long)img * h + hh) * w + ww) * c + ch);
    emsa_st1(y + i, a / (float)((h1 - h0) * (w1 - w0)));
  }
}

// One workgroup per (image, bin): (channel vectors of 16 bytes) x (pixel lanes) threads walk the
// bin's pixels with four loads in flight each, LDS reduction over the pixel lanes.  (The kernel
// above gives every output element to one thread: at batch 1 the 1-bin branch of the pyramid
// pooling was 512 threads summing 300 pixels one load at a time -- 78 us for 300 KB.)
template <typename T>
__global__ void adaptive_avgpool_fwd_wg_kernel(const T* __restrict__ x, T* __restrict__ y, int h,
                                               int w, int c, int bins) {
  constexpr int V = VecIO<T>::V;
  extern __shared__ __attribute__((aligned(16))) float pred[];       // [lanes][c]
  const int img = blockIdx.x / (bins * bins), b = blockIdx.x % (bins * bins);
  const int bi = b / bins, bj = b % bins;
  int h0, h1, w0, w1;
  bin_range(bi, h, bins, h0, h1);
  bin_range(bj, w, bins, w0, w1);
  const int cvn = c / V, lanes = blockDim.x / cvn;
  const int cv = threadIdx.x % cvn, rl = threadIdx.x / cvn;
  const int bw = w1 - w0, npx = (h1 - h0) * bw;
  if (rl < lanes) {
    float acc[V];
#pragma unroll
    for (int k = 0; k < V; ++k) acc[k] = 0.f;
#pragma unroll 4
    for (int p = rl; p < npx; p += lanes) {
      const int hh = h0 + p / bw, ww = w0 + p % bw;
      float v[V];
      VecIO<T>::load(x + (((long)img * h + hh) * w + ww) * c + cv * V, v);
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += v[k];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) pred[rl * c + cv * V + k] = acc[k];
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
    float t = 0.f;
    for (int k = 0; k < lanes; ++k) t += pred[k * c + ch];
    emsa_st1(y + ((long)blockIdx.x) * c + ch, t / (float)npx);
  }
}

template <typename T>
__global__ void adaptive_avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx,
                                            int n, int h, int w, int c, int bins, int accumulate) {
  const long total = (long)n * h * w * c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long r = i / c;
    const int ww = (int)(r % w); r /= w;
    const int hh = (int)(r % h);
    const int img = (int)(r / h);
    float a = 0.f;
    for (int bi = 0; bi < bins; ++bi) {
      int h0, h1;
      bin_range(bi, h, bins, h0, h1);
      if (hh < h0 || hh >= h1) continue;
      for (int bj = 0; bj < bins; ++bj) {
        int w0, w1;
        bin_range(bj, w, bins, w0, w1);
        if (ww < w0 || ww >= w1) continue;
        a += emsa_ld1(dy + (((long)img * bins + bi) * bins + bj) * c + ch) /
             (float)((h1 - h0) * (w1 - w0));
      }
    }
    emsa_st1(dx + i, accumulate ? emsa_ld1(dx + i) + a : a);
  }
}

__device__ __forceinline__ void bilinear_src(int o, int in, int out, int& i0, int& i1, float& l) {
  const float scale = (float)in / (float)out;
  float s = scale * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l = s - (float)i0;
}

template <typename T>
__global__ void bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n,
                                    int ih, int iw, int oh, int ow, int c, int ld_y) {
  // no fma contraction: the compiler contracted the interpolation differently in the peeled first
  // iteration and in the steady-state body of the grid-stride loop, so an element's last bit depended
  // on how many iterations its thread ran, i.e. on the batch size (found by
  // tools/batch_invariance_probe.py: the only batch-dependent rounding left in the eval forward once
  // the channel reductions are partitioned by map size, emsa_set_batch_invariant)
#pragma clang fp contract(off)
  const long total = (long)n * oh * ow * c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long r = i / c;
    const int xo = (int)(r % ow); r /= ow;
    const int yo = (int)(r % oh);
    const int img = (int)(r / oh);
    int h0, h1, w0, w1;
    float lh, lw;
    bilinear_src(yo, ih, oh, h0, h1, lh);
    bilinear_src(xo, iw, ow, w0, w1, lw);
    const T* b = x + (long)img * ih * iw * c + ch;
    const float v00 = emsa_ld1(b + ((long)h0 * iw + w0) * c), v01 = emsa_ld1(b + ((long)h0 * iw + w1) * c);
    const float v10 = emsa_ld1(b + ((long)h1 * iw + w0) * c), v11 = emsa_ld1(b + ((long)h1 * iw + w1) * c);
    emsa_st1(y + (((long)img * oh + yo) * ow + xo) * ld_y + ch,
             (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11));
  }
}

// dx is ALWAYS fp32.  GATHER form (round 6): one thread per dx element walks the output rows / columns
// whose two source taps include its own row / column and sums their contributions in a fixed
// order -- no atomics, no zero-fill, bit-reproducible.  (The scatter form with fp32 atomics was the one
// non-reproducible link of the ACTIVATION-gradient chain: in 16-bit storage its 1e-7 jitter flips
// a few roundings of the pyramid-pooling branch gradients, and 30 blocks of bf16 rounding further
// up the encoder those few flips have grown into a different 1.5 % noise realisation of every
// encoder gradient, run to run -- tools/grad_repeat_probe.py.)
template <typename T>
__global__ void bilinear_bwd_kernel(const T* __restrict__ dy, float* __restrict__ dx, int n,
                                    int ih, int iw, int oh, int ow, int c, int ld_dy) {
#pragma clang fp contract(off)
  const long total = (long)n * ih * iw * c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long r = i / c;
    const int xi = (int)(r % iw); r /= iw;
    const int yi = (int)(r % ih);
    const int img = (int)(r / ih);
    const T* b = dy + (long)img * oh * ow * ld_dy + ch;
    float a = 0.f;
    // output rows / columns that can touch input row yi / column xi: source coordinate in
    // (yi - 1, yi + 1), i.e. |o - (yi + 0.5) * oh / ih + 0.5| < oh / ih; one output row of margin on
    // both sides, the membership test below decides (the window only bounds the walk: a 640-wide
    // map would otherwise cost 640 source computations per element and row)
    const float ry = (float)oh / (float)ih, rx = (float)ow / (float)iw;
    int y_lo = (int)(((float)yi - 1.f) * ry) - 2, y_hi = (int)(((float)yi + 2.f) * ry) + 2;
    int x_lo = (int)(((float)xi - 1.f) * rx) - 2, x_hi = (int)(((float)xi + 2.f) * rx) + 2;
    if (yi == 0) y_lo = 0;                             // (clamped border: everything above maps here)
    if (yi == ih - 1) y_hi = oh - 1;
    if (xi == 0) x_lo = 0;
    if (xi == iw - 1) x_hi = ow - 1;
    y_lo = y_lo < 0 ? 0 : y_lo; y_hi = y_hi > oh - 1 ? oh - 1 : y_hi;
    x_lo = x_lo < 0 ? 0 : x_lo; x_hi = x_hi > ow - 1 ? ow - 1 : x_hi;
    for (int yo = y_lo; yo <= y_hi; ++yo) {
      int h0, h1;
      float lh;
      bilinear_src(yo, ih, oh, h0, h1, lh);
      // (h0 == h1 at the clamped border: both taps land on the same row, as in the scatter form)
      const float wy = (h0 == yi ? 1.f - lh : 0.f) + (h1 == yi ? lh : 0.f);
      if (h0 != yi && h1 != yi) continue;
      for (int xo = x_lo; xo <= x_hi; ++xo) {
        int w0, w1;
        float lw;
        bilinear_src(xo, iw, ow, w0, w1, lw);
        if (w0 != xi && w1 != xi) continue;
        const float wx = (w0 == xi ? 1.f - lw : 0.f) + (w1 == xi ? lw : 0.f);
        a += emsa_ld1(b + ((long)yo * ow + xo) * ld_dy) * (wy * wx);
      }
    }
    dx[i] = a;
  }
}

// 'nearest' up-sampling of the pyramid-pooling branches (`--upsampling-context-module nearest`,
// /root/reference/emsanet/args.py:250-256; torch.nn.functional.interpolate(mode='nearest'): source index
// floor(dst * in / out) in float arithmetic, clamped).  Backward: gather, one thread per dx element over
// the window of outputs that map to it, fixed order.
__device__ __forceinline__ int nearest_src(int o, int in, int out) {
  const float scale = (float)in / (float)out;
  const int i = (int)floorf((float)o * scale);
  return i < in - 1 ? i : in - 1;
}
template <typename T>
__global__ void nearest_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int ih, int iw,
                                   int oh, int ow, int c, int ld_y) {
  const long total = (long)n * oh * ow * c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long r = i / c;
    const int xo = (int)(r % ow); r /= ow;
    const int yo = (int)(r % oh);
    const int img = (int)(r / oh);
    const int yi = nearest_src(yo, ih, oh), xi = nearest_src(xo, iw, ow);
    emsa_st1(y + (((long)img * oh + yo) * ow + xo) * ld_y + ch,
             emsa_ld1(x + (((long)img * ih + yi) * iw + xi) * c + ch));
  }
}
// out = a + b (dense, same storage type; fp32 sum rounded once): the skip add behind a plain
// 'nearest' / 'bilinear' decoder up-sampling (the learned up-sampling adds its skip in its own kernel)
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x)
    emsa_st1(y + i, emsa_ld1(a + i) + emsa_ld1(b + i));
}
template <typename T>
__global__ void nearest_bwd_kernel(const T* __restrict__ dy, float* __restrict__ dx, int n, int ih,
                                   int iw, int oh, int ow, int c, int ld_dy) {
  const long total = (long)n * ih * iw * c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long r = i / c;
    const int xi = (int)(r % iw); r /= iw;
    const int yi = (int)(r % ih);
    const int img = (int)(r / ih);
    const T* b = dy + (long)img * oh * ow * ld_dy + ch;
    // outputs o with nearest_src(o) == yi lie in [yi * oh / ih, (yi + 1) * oh / ih) up to float
    // rounding: one output of margin on both sides, the membership test decides
    const float ry = (float)oh / (float)ih, rx = (float)ow / (float)iw;
    int y_lo = (int)((float)yi * ry) - 1, y_hi = (int)((float)(yi + 1) * ry) + 1;
    int x_lo = (int)((float)xi * rx) - 1, x_hi = (int)((float)(xi + 1) * rx) + 1;
    y_lo = y_lo < 0 ? 0 : y_lo; y_hi = y_hi > oh - 1 ? oh - 1 : y_hi;
    x_lo = x_lo < 0 ? 0 : x_lo; x_hi = x_hi > ow - 1 ? ow - 1 : x_hi;
    float a = 0.f;
    for (int yo = y_lo; yo <= y_hi; ++yo) {
      if (nearest_src(yo, ih, oh) != yi) continue;
      for (int xo = x_lo; xo <= x_hi; ++xo)
        if (nearest_src(xo, iw, ow) == xi) a += emsa_ld1(b + ((long)yo * ow + xo) * ld_dy);
    }
    dx[i] = a;
  }
}

// ------------------------------------------------------------------------------------------
// head activations, copies
// ------------------------------------------------------------------------------------------
// channels [0, n_sig): sigmoid; [n_sig, n_sig + n_tanh): tanh; channels [norm_off, norm_off +
// n_norm) (n_norm 0 or 2: the orientation biternion, oracle Spec.ORIENTATION_L2_NORMALIZE) are
// L2-normalised over the channel pair like F.normalize(dim=1, eps=1e-12); the rest passes through
template <typename T, typename TO>
__global__ void head_act_fwd_kernel(const T* __restrict__ x, TO* __restrict__ y, long total,
                                    int c, int n_sig, int n_tanh, int norm_off, int n_norm) {
  const int o = norm_off, ot = n_sig + n_tanh;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    const float v = emsa_ld1(x + i);
    float r = ch < n_sig ? 1.f / (1.f + expf(-v)) : ch < ot ? tanhf(v) : v;
    if (ch >= o && ch < o + n_norm) {
      const float u = emsa_ld1(x + i + (ch == o ? 1 : -1));
      r = v / fmaxf(sqrtf(v * v + u * u), 1e-12f);
    }
    emsa_st1(y + i, r);
  }
}

// the same for the 8-channel (padded) instance head with fp32 outputs, one thread per PIXEL: 16 / 32
// bytes in, 32 bytes out per thread instead of one element (the map is 315 MB of fp32 at bs 32)
template <typename T>
__global__ void head_act_fwd8_kernel(const T* __restrict__ x, float* __restrict__ y, long pixels,
                                     int n_sig, int n_tanh, int norm_off, int n_norm) {
  constexpr int C = 8, V = VecIO<T>::V;
  const int ot = n_sig + n_tanh;
  for (long px = blockIdx.x * (long)blockDim.x + threadIdx.x; px < pixels;
       px += (long)gridDim.x * blockDim.x) {
    float v[C], part[V];
#pragma unroll
    for (int h = 0; h < C / V; ++h) {
      VecIO<T>::load(x + px * C + h * V, part);
#pragma unroll
      for (int k = 0; k < V; ++k) v[h * V + k] = part[k];
    }
    float r[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      r[ch] = ch < n_sig ? 1.f / (1.f + expf(-v[ch])) : ch < ot ? tanhf(v[ch]) : v[ch];
      if (ch >= norm_off && ch < norm_off + n_norm) {
        const float u = ch == norm_off ? v[ch + 1 < C ? ch + 1 : ch] : v[ch > 0 ? ch - 1 : 0];
        r[ch] = v[ch] / fmaxf(sqrtf(v[ch] * v[ch] + u * u), 1e-12f);
      }
    }
    emsa_st4(y + px * C, make_float4(r[0], r[1], r[2], r[3]));
    emsa_st4(y + px * C + 4, make_float4(r[4], r[5], r[6], r[7]));
  }
}

template <typename T, typename TO>
__global__ void head_act_bwd_kernel(const TO* __restrict__ dy, const TO* __restrict__ y,
                                    const T* __restrict__ x, T* __restrict__ dx,
                                    long total, int c, int n_sig, int n_tanh, int norm_off,
                                    int n_norm) {
  const int o = norm_off, ot = n_sig + n_tanh;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    const float g = emsa_ld1(dy + i), v = emsa_ld1(y + i);
    float r = ch < n_sig ? g * v * (1.f - v) : ch < ot ? g * (1.f - v * v) : g;
    if (ch >= o && ch < o + n_norm) {
      // y = x / max(|x|, eps):  dx = (g - y (y . g)) / |x|   (|x| > eps), g / eps otherwise
      const long j = i + (ch == o ? 1 : -1);
      const float xa = emsa_ld1(x + i), xb = emsa_ld1(x + j);
      const float nrm = sqrtf(xa * xa + xb * xb);
      r = nrm > 1e-12f ? (g - v * (v * g + emsa_ld1(y + j) * emsa_ld1(dy + j))) / nrm : g / 1e-12f;
    }
    emsa_st1(dx + i, r);
  }
}

// head_act_bwd with the GATHER of the task gradients in front of it: the instance head hands out
// channel-slice views of one activated 8-channel tensor (centre | offset | orientation | padding), and
// its backward pass used to copy the three incoming fp32 gradients into a padded 8-channel tensor
// (three strided copies + a fill of the padding channels: 315 MB written and read again per step at
// bs 32) before this kernel ran.  Here one thread owns one PIXEL: it reads its gradient channels from
// up to three sources (channel count cs[k], pixel stride ld[k]; a NULL source = zero gradient), the
// pixel's 8 outputs y, and writes the pixel's 8 input gradients (padding channels: 0) in one piece.
struct HeadGatherArgs {
  const float* g[3];
  int cs[3], ld[3];
};
template <typename T>
__global__ void head_act_bwd_gather_kernel(const HeadGatherArgs a, const float* __restrict__ y,
                                           const T* __restrict__ x, T* __restrict__ dx,
                                           long pixels, int n_sig, int n_tanh, int norm_off,
                                           int n_norm) {
  constexpr int C = 8;
  const int ot = n_sig + n_tanh;
  for (long px = blockIdx.x * (long)blockDim.x + threadIdx.x; px < pixels;
       px += (long)gridDim.x * blockDim.x) {
    // (every register index below is a compile-time constant: which source a channel comes from is
    //  decided per channel, not by walking the sources -- a run-time index into g[] would put it in
    //  scratch memory)
    const int o1 = a.cs[0], o2 = o1 + a.cs[1], o3 = o2 + a.cs[2];
    float g[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
      const float* src = ch < o1 ? a.g[0] : ch < o2 ? a.g[1] : ch < o3 ? a.g[2] : nullptr;
      const int ld = ch < o1 ? a.ld[0] : ch < o2 ? a.ld[1] : a.ld[2];
      const int k = ch < o1 ? ch : ch < o2 ? ch - o1 : ch - o2;
      g[ch] = src ? src[px * ld + k] : 0.f;
    }
    const float4 y0 = emsa_ld4(y + px * C), y1 = emsa_ld4(y + px * C + 4);
    const float v[C] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
    float r[C];
#pragma unroll
    for (int ch = 0; ch < C; ++ch)
      r[ch] = ch < n_sig ? g[ch] * v[ch] * (1.f - v[ch]) : ch < ot ? g[ch] * (1.f - v[ch] * v[ch]) : g[ch];
    if (n_norm) {
      // y = x / max(|x|, eps):  dx = (g - y (y . g)) / |x|   (|x| > eps), g / eps otherwise
      const float xa = emsa_ld1(x + px * C + norm_off), xb = emsa_ld1(x + px * C + norm_off + 1);
      const float nrm = sqrtf(xa * xa + xb * xb);
#pragma unroll
      for (int ch = 0; ch < C; ++ch)
        if (ch == norm_off || ch == norm_off + 1) {
          const bool first = ch == norm_off;
          const float vj = first ? v[ch + 1 < C ? ch + 1 : ch] : v[ch > 0 ? ch - 1 : 0];
          const float gj = first ? g[ch + 1 < C ? ch + 1 : ch] : g[ch > 0 ? ch - 1 : 0];
          r[ch] = nrm > 1e-12f ? (g[ch] - v[ch] * (v[ch] * g[ch] + vj * gj)) / nrm : g[ch] / 1e-12f;
        }
    }
    float rr[VecIO<T>::V];
#pragma unroll
    for (int h = 0; h < C / VecIO<T>::V; ++h) {
#pragma unroll
      for (int k = 0; k < VecIO<T>::V; ++k) rr[k] = r[h * VecIO<T>::V + k];
      VecIO<T>::store(dx + px * C + h * VecIO<T>::V, rr);
    }
  }
}

__global__ void u32_add_kernel(uint32_t* p, uint32_t v) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *p += v;
}

template <typename TS, typename TD>
__global__ void copy_channels_kernel(const TS* __restrict__ x, int ld_x, TD* __restrict__ y,
                                     int ld_y, long pixels, int c) {
  const long total = pixels * c;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long p = i / c;
    const int ch = (int)(i % c);
    emsa_st1(y + p * ld_y + ch, emsa_ld1(x + p * ld_x + ch));
  }
}

// ---- module boundary: any layout of a logical (N, C, H, W) tensor -> dense NHWC -----------------
// The reference's task helpers / tests hand the engine contiguous NCHW tensors (decoder inputs of
// /root/reference/emsanet/tests/test_interface_decoders.py:73-88, cotangents of the NCHW losses).
// Plane-contiguous sources (stride_w = 1, stride_h = W, stride_c = H*W: contiguous NCHW, any image
// stride) go through a 64 (pixels) x 64 (channels) LDS tile: 256-byte reads along the plane, full
// channel rows written; every other layout (expanded scalars, sliced / permuted views) takes the
// element-wise gather form.  One pass either way -- no torch permute().contiguous().
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_tile_kernel(const T* __restrict__ x,
                                                                T* __restrict__ y, int c, long hw,
                                                                long s_n, int tiles_c,
                                                                long tiles_hw) {
  __shared__ T tile[64][65];
  const long b = blockIdx.x;
  const int tc = (int)(b % tiles_c);
  const long thw = (b / tiles_c) % tiles_hw;
  const long img = b / ((long)tiles_c * tiles_hw);
  const int c0 = tc * 64;
  const long p0 = thw * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const T* src = x + img * s_n;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const int ch = c0 + ty + 4 * k;
    const long px = p0 + tx;
    if (ch < c && px < hw) tile[ty + 4 * k][tx] = src[(long)ch * hw + px];
  }
  __syncthreads();
  T* dst = y + img * hw * c;
#pragma unroll 4
  for (int k = 0; k < 16; ++k) {
    const long px = p0 + ty + 4 * k;
    const int ch = c0 + tx;
    if (ch < c && px < hw) dst[px * c + ch] = tile[tx][ty + 4 * k];
  }
}

template <typename T>
__global__ void strided_to_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int c, int h,
                                       int w, long total, long s_n, long s_c, long s_h, long s_w) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % c);
    long r = i / c;
    const int ww = (int)(r % w);
    r /= w;
    const int hh = (int)(r % h);
    const long img = r / h;
    y[i] = x[img * s_n + ch * s_c + hh * s_h + ww * s_w];
  }
}

__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, long n,
                            float alpha) {
  const long n4 = n / 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4;
       i += (long)gridDim.x * blockDim.x) {
    const float4 a = emsa_ld4(x + i * 4);
    float4 b = emsa_ld4(y + i * 4);
    b.x += alpha * a.x; b.y += alpha * a.y; b.z += alpha * a.z; b.w += alpha * a.w;
    emsa_st4(y + i * 4, b);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = n4 * 4 + threadIdx.x;
    y[i] += alpha * x[i];
  }
}

// channel counts the pointwise kernels take: multiples of 4 up to 4096 (the bottleneck ResNets end at
// 2048).  Kernels that map channel vectors onto the threads of one workgroup (reductions) take
// c / V <= kThreads on their fast paths and switch to their *_wide_kernel forms beyond.
inline bool c4_ok(int c) { return c >= 4 && (c & 3) == 0 && c <= 4096; }


// ------------------------------------------------------------------------------------------
// SGD with Nesterov momentum and weight decay over one flat bucket: the update of
// torch.optim.SGD(momentum, weight_decay, nesterov=True) the reference trains with
// (/root/reference/emsanet/optimizer.py:29-36):
//   d = g * grad_scale + wd * p;  buf = first ? d : mu * buf + d;  p -= lr * (d + mu * buf)
// one pass: reads p, g, buf, writes p, buf (5 streams instead of the ~12 of a foreach SGD).
// ------------------------------------------------------------------------------------------
// hp (optional): the hyper-parameters {lr, momentum, weight_decay, grad_scale, first_step} in DEVICE
// memory -- a training step captured in a hipGraph reads the schedule's current values at replay
// time instead of the values baked into the launch
__global__ void sgd_nesterov_kernel(float* __restrict__ p, const float* __restrict__ g,
                                    float* __restrict__ m, long n4, long n, float lr, float mu,
                                    float wd, float gs, int first, const float* __restrict__ hp) {
  if (hp) {
    lr = hp[0]; mu = hp[1]; wd = hp[2]; gs = hp[3];
    first = hp[4] != 0.f;
  }
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4;
       i += (long)gridDim.x * blockDim.x) {
    const float4 pv = emsa_ld4(p + i * 4), gv = emsa_ld4(g + i * 4);
    float4 d = make_float4(gv.x * gs + wd * pv.x, gv.y * gs + wd * pv.y, gv.z * gs + wd * pv.z,
                           gv.w * gs + wd * pv.w);
    float4 b = d;
    if (!first) {
      const float4 mv = emsa_ld4(m + i * 4);
      b = make_float4(mu * mv.x + d.x, mu * mv.y + d.y, mu * mv.z + d.z, mu * mv.w + d.w);
    }
    emsa_st4(m + i * 4, b);
    emsa_st4(p + i * 4, make_float4(pv.x - lr * (d.x + mu * b.x), pv.y - lr * (d.y + mu * b.y),
                                    pv.z - lr * (d.z + mu * b.z), pv.w - lr * (d.w + mu * b.w)));
  }
  // tail (n not a multiple of 4)
  const long t = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (t < n) {
    const float pv = p[t], d = g[t] * gs + wd * pv;
    const float b = first ? d : mu * m[t] + d;
    m[t] = b;
    p[t] = pv - lr * (d + mu * b);
  }
}

// ------------------------------------------------------------------------------------------
// Adam / AdamW / RAdam over one flat bucket: the other three optimizers the reference's factory
// builds (/root/reference/emsanet/optimizer.py:37-57: betas (0.9, 0.999), torch's eps 1e-8, weight
// decay as L2 for adam / radam and decoupled for adamw).  The arithmetic follows torch.optim's
// single-tensor paths:
//   g' = g * grad_scale (+ wd * p for adam / radam);   adamw: p *= 1 - lr * wd
//   m += (g' - m) * (1 - b1);   v = b2 * v + (1 - b2) * g'^2
//   adam / adamw:  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
//   radam:         p -= (m / bc1) * lr * (rho_t > 5 ? sqrt(bc2) / (sqrt(v) + eps) * rect : 1)
// hyper (double[8], host-owned): {lr, b1, b2, eps, wd, grad_scale, mode (0 adam, 1 adamw, 2 radam)}
// state (double[8], DEVICE-owned): {t, lr / bc1, bc1, sqrt(bc2), rect, rho_t > 5} -- advanced by a
// one-thread kernel in front of the bucket launches, so that a step captured in a hipGraph counts
// its own replays; the step-dependent scalars are formed in double like torch's Python floats.
// One pass: reads p, g, m, v, writes p, m, v.
// ------------------------------------------------------------------------------------------
__global__ void adam_advance_kernel(const double* __restrict__ hp, double* __restrict__ st) {
  const double b1 = hp[1], b2 = hp[2];
  const double t = st[0] + 1.0;
  const double bc1 = 1.0 - pow(b1, t), bc2 = 1.0 - pow(b2, t);
  const double rho_inf = 2.0 / (1.0 - b2) - 1.0;
  const double rho_t = rho_inf - 2.0 * t * pow(b2, t) / bc2;
  double rect = 0.0;
  if (rho_t > 5.0)
    rect = sqrt((rho_t - 4.0) * (rho_t - 2.0) * rho_inf / ((rho_inf - 4.0) * (rho_inf - 2.0) * rho_t));
  st[0] = t;
  st[1] = hp[0] / bc1;
  st[2] = bc1;
  st[3] = sqrt(bc2);
  st[4] = rect;
  st[5] = rho_t > 5.0 ? 1.0 : 0.0;
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr, float b1,
                                         float b2, float eps, float wd, float gs, int mode,
                                         float step_size, float bc1, float bc2_sqrt, float rect,
                                         bool use_rect) {
  float gg = g * gs;
  if (mode == 1) p = p * (1.f - lr * wd);
  else if (wd != 0.f) gg = gg + wd * p;
  m = m + (gg - m) * (1.f - b1);
  v = b2 * v + (1.f - b2) * gg * gg;
  if (mode != 2) {
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
  } else {
    const float mh = m / bc1;
    p = use_rect ? p - mh * lr * (bc2_sqrt / (sqrtf(v) + eps)) * rect : p - mh * lr;
  }
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n4, long n, const double* __restrict__ hp,
                            const double* __restrict__ st) {
#pragma clang fp contract(off)
  const float lr = (float)hp[0], b1 = (float)hp[1], b2 = (float)hp[2], eps = (float)hp[3];
  const float wd = (float)hp[4], gs = (float)hp[5];
  const int mode = (int)hp[6];
  const float step_size = (float)st[1], bc1 = (float)st[2], bc2_sqrt = (float)st[3];
  const float rect = (float)st[4];
  const bool use_rect = st[5] != 0.0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4;
       i += (long)gridDim.x * blockDim.x) {
    float4 pv = emsa_ld4(p + i * 4), mv = emsa_ld4(m + i * 4), vv = emsa_ld4(v + i * 4);
    const float4 gv = emsa_ld4(g + i * 4);
    adam_one(pv.x, gv.x, mv.x, vv.x, lr, b1, b2, eps, wd, gs, mode, step_size, bc1, bc2_sqrt, rect, use_rect);
    adam_one(pv.y, gv.y, mv.y, vv.y, lr, b1, b2, eps, wd, gs, mode, step_size, bc1, bc2_sqrt, rect, use_rect);
    adam_one(pv.z, gv.z, mv.z, vv.z, lr, b1, b2, eps, wd, gs, mode, step_size, bc1, bc2_sqrt, rect, use_rect);
    adam_one(pv.w, gv.w, mv.w, vv.w, lr, b1, b2, eps, wd, gs, mode, step_size, bc1, bc2_sqrt, rect, use_rect);
    emsa_st4(m + i * 4, mv);
    emsa_st4(v + i * 4, vv);
    emsa_st4(p + i * 4, pv);
  }
  const long t = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (t < n) {
    float pv = p[t], mv = m[t], vv = v[t];
    adam_one(pv, g[t], mv, vv, lr, b1, b2, eps, wd, gs, mode, step_size, bc1, bc2_sqrt, rect, use_rect);
    m[t] = mv; v[t] = vv; p[t] = pv;
  }
}

}  // namespace

// ==========================================================================================
// C-ABI
// ==========================================================================================
extern "C" const char* emsa_arch(void) { return "gfx950"; }
extern "C" int emsa_version(void) { return 1; }

static int pack_common(const float* src, float* dst, int cout, int cin, int kh, int kw,
                       int cout_total, int cout_off, int cin_total, int cin_off, int mode,
                       void* stream) {
  if (!src || !dst) return EMSA_E_ARG;
  if (cout_off + cout > cout_total || cin_off + cin > cin_total) return EMSA_E_SHAPE;
  const long total = (long)cout * cin * kh * kw;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for(total)), dim3(kThreads), 0,
                     (hipStream_t)stream, src, dst, cout, cin, kh, kw, cout_total, cout_off,
                     cin_total, cin_off, mode, (float*)nullptr);
  return emsa_launch_status();
}

extern "C" int emsa_pack_weight_fwd(const float* w, float* wp, int32_t cout, int32_t cin,
                                    int32_t kh, int32_t kw, int32_t cout_total, int32_t cout_off,
                                    int32_t cin_total, int32_t cin_off, void* stream) {
  return pack_common(w, wp, cout, cin, kh, kw, cout_total, cout_off, cin_total, cin_off, 0,
                     stream);
}
extern "C" int emsa_pack_weight_dgrad(const float* w, float* wp, int32_t cout, int32_t cin,
                                      int32_t kh, int32_t kw, int32_t cout_total,
                                      int32_t cout_off, int32_t cin_total, int32_t cin_off,
                                      void* stream) {
  return pack_common(w, wp, cout, cin, kh, kw, cout_total, cout_off, cin_total, cin_off, 1,
                     stream);
}
extern "C" int emsa_pack_weight_pair(const float* w, float* wp_fwd, float* wp_dgrad, int32_t cout,
                                     int32_t cin, int32_t kh, int32_t kw, void* stream) {
  if (!w || !wp_fwd || !wp_dgrad) return EMSA_E_ARG;
  const long total = (long)cout * cin * kh * kw;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for(total)), dim3(kThreads), 0,
                     (hipStream_t)stream, w, wp_fwd, cout, cin, kh, kw, cout, 0, cin, 0, 3,
                     wp_dgrad);
  return emsa_launch_status();
}
extern "C" int emsa_unpack_wgrad(const float* dwp, float* dw, int32_t cout, int32_t cin,
                                 int32_t kh, int32_t kw, int32_t cout_total, int32_t cout_off,
                                 int32_t cin_total, int32_t cin_off, void* stream) {
  return pack_common(dwp, dw, cout, cin, kh, kw, cout_total, cout_off, cin_total, cin_off, 2,
                     stream);
}

template <typename T>
static int stem_pack_input_impl(const float* x, T* xp, int32_t n, int32_t c, int32_t h, int32_t w,
                                void* stream) {
  if (!x || !xp) return EMSA_E_ARG;
  if (c < 1 || c > 4) return EMSA_E_SHAPE;
  const long total = (long)n * h * (w + 8);
  hipLaunchKernelGGL((stem_pack_input_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0,
                     (hipStream_t)stream, x, xp, n, c, h, w);
  return emsa_launch_status();
}
extern "C" int emsa_stem_pack_input(const float* x, float* xp, int32_t n, int32_t c, int32_t h,
                                    int32_t w, void* stream) {
  return stem_pack_input_impl<float>(x, xp, n, c, h, w, stream);
}
// the network input stays fp32 NCHW (emsanet/model.py:192); the packed NHWC4 image is written in
// the engine's storage type
extern "C" int emsa_stem_pack_input_t(int32_t dtype, const float* x, void* xp, int32_t n, int32_t c,
                                      int32_t h, int32_t w, void* stream) {
  EMSA_DISPATCH_DTYPE(dtype, T, return stem_pack_input_impl<T>(x, (T*)xp, n, c, h, w, stream));
  return EMSA_E_ARG;
}
extern "C" int emsa_stem_pack_weight(const float* w, float* wp, int32_t cout, int32_t cin,
                                     void* stream) {
  if (!w || !wp) return EMSA_E_ARG;
  if (cin < 1 || cin > 4) return EMSA_E_SHAPE;
  hipLaunchKernelGGL(stem_pack_weight_kernel<float>, dim3(grid_for(7L * cout * 32)), dim3(kThreads),
                     0, (hipStream_t)stream, w, wp, cout, cin, 0);
  return emsa_launch_status();
}
extern "C" int emsa_stem_pack_weight_t(int32_t dtype, const float* w, void* wp, int32_t cout,
                                       int32_t cin, void* stream) {
  if (!w || !wp) return EMSA_E_ARG;
  if (cin < 1 || cin > 4) return EMSA_E_SHAPE;
  EMSA_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL(stem_pack_weight_kernel<T>, dim3(grid_for(7L * cout * 32)), dim3(kThreads),
                       0, (hipStream_t)stream, w, (T*)wp, cout, cin, 0);
  });
  return emsa_launch_status();
}
// fp32 OIHW parameter -> 16-bit packed operand layouts of emsa_conv_igemm_t (forward
// [tap][cout_total][cin_total] and / or data gradient [tap][cin_total][cout_total]; a destination
// may be NULL; padding rows / columns of a merged or channel-padded conv must be zeroed by the caller)
extern "C" int emsa_pack_weight_t(int32_t dtype, const float* w, void* wp_fwd, void* wp_dgrad,
                                  int32_t cout, int32_t cin, int32_t kh, int32_t kw,
                                  int32_t cout_total, int32_t cout_off, int32_t cin_total,
                                  int32_t cin_off, void* stream) {
  if (!w || (!wp_fwd && !wp_dgrad)) return EMSA_E_ARG;
  if (cout_off + cout > cout_total || cin_off + cin > cin_total) return EMSA_E_SHAPE;
  const long total = (long)cout * cin * kh * kw;
  EMSA_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL(pack_weight_t_kernel<T>, dim3(grid_for(total)), dim3(kThreads), 0,
                       (hipStream_t)stream, w, (T*)wp_fwd, (T*)wp_dgrad, cout, cin, kh * kw,
                       cout_total, cout_off, cin_total, cin_off);
  });
  return emsa_launch_status();
}
extern "C" int emsa_stem_unpack_wgrad(const float* dwp, float* dw, int32_t cout, int32_t cin,
                                      void* stream) {
  if (!dwp || !dw) return EMSA_E_ARG;
  if (cin < 1 || cin > 4) return EMSA_E_SHAPE;
  hipLaunchKernelGGL(stem_pack_weight_kernel<float>, dim3(grid_for(7L * cout * 32)), dim3(kThreads),
                     0, (hipStream_t)stream, dwp, dw, cout, cin, 1);
  return emsa_launch_status();
}

// one-channel stem in the rows-as-channels layout (see stem_pack_input_rows_kernel): packed image
// [n][h + 4][w + 8][4], packed weights [2][cout][32]
extern "C" int emsa_stem_pack_input_rows_t(int32_t dtype, const float* x, void* xp, int32_t n,
                                           int32_t h, int32_t w, void* stream) {
  if (!x || !xp) return EMSA_E_ARG;
  const long total = (long)n * (h + 4) * (w + 8);
  EMSA_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((stem_pack_input_rows_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0,
                       (hipStream_t)stream, x, (T*)xp, n, h, w);
    return emsa_launch_status();
  });
  return EMSA_E_ARG;
}
extern "C" int emsa_stem_pack_weight_rows_t(int32_t dtype, const float* w, void* wp, int32_t cout,
                                            void* stream) {
  if (!w || !wp) return EMSA_E_ARG;
  EMSA_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL(stem_pack_weight_rows_kernel<T>, dim3(grid_for(2L * cout * 32)),
                       dim3(kThreads), 0, (hipStream_t)stream, w, (T*)wp, cout, 0);
    return emsa_launch_status();
  });
  return EMSA_E_ARG;
}
extern "C" int emsa_stem_unpack_wgrad_rows(const float* dwp, float* dw, int32_t cout, void* stream) {
  if (!dwp || !dw) return EMSA_E_ARG;
  hipLaunchKernelGGL(stem_pack_weight_rows_kernel<float>, dim3(grid_for(2L * cout * 32)),
                     dim3(kThreads), 0, (hipStream_t)stream, dwp, dw, cout, 1);
  return emsa_launch_status();
}

extern "C" int emsa_bn_finalize_ws_bytes(int32_t c) {
  return (int)(kBnSlices * (size_t)c * 3 * sizeof(double));
}

extern "C" int emsa_bn_finalize(const float* stats, int32_t rows, int32_t c, int64_t count,
                                const float* gamma, const float* beta, float eps, float momentum,
                                float* running_mean, float* running_var, float* scale,
                                float* shift, float* save_mean, float* save_invstd, void* ws,
                                void* stream) {
  if (!stats || !gamma || !beta || !scale || !shift || !save_mean || !save_invstd || !ws)
    return EMSA_E_ARG;
  (void)count;
  int slices = (rows + 31) / 32;
  if (slices > kBnSlices) slices = kBnSlices;
  if (slices < 1) slices = 1;
  hipStream_t st = (hipStream_t)stream;
  if (rows <= 512) {
    hipLaunchKernelGGL(bn_finalize_rows_kernel<64>, dim3((c + 3) / 4), dim3(256), 0, st, stats, rows, c,
                       gamma, beta, eps, momentum, running_mean, running_var, scale, shift,
                       save_mean, save_invstd);
    return emsa_launch_status();
  }
  // EMSA_BN_FINALIZE_WIDE=0: the two-launch form for > 512 rows (A/B)
  static const bool wide = [] {
    const char* e = getenv("EMSA_BN_FINALIZE_WIDE");
    return !(e && e[0] == '0');
  }();
  if (wide && rows <= 8192) {
    hipLaunchKernelGGL(bn_finalize_rows_kernel<256>, dim3((c + 3) / 4), dim3(1024), 0, st, stats, rows,
                       c, gamma, beta, eps, momentum, running_mean, running_var, scale, shift,
                       save_mean, save_invstd);
    return emsa_launch_status();
  }
  hipLaunchKernelGGL(bn_partial_kernel, dim3((c + 31) / 32, slices), dim3(256), 0, st, stats, rows,
                     c, (double*)ws);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 31) / 32), dim3(256), 0, st,
                     (const double*)ws, slices, c, gamma, beta, eps, momentum, running_mean,
                     running_var, scale, shift, save_mean, save_invstd);
  return emsa_launch_status();
}

extern "C" int emsa_bn_fold(const float* gamma, const float* beta, const float* rm,
                            const float* rv, float eps, int32_t c, float* scale, float* shift,
                            float* save_invstd, void* stream) {
  if (!gamma || !beta || !rm || !rv || !scale || !shift) return EMSA_E_ARG;
  hipLaunchKernelGGL(bn_fold_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     gamma, beta, rm, rv, eps, c, scale, shift, save_invstd);
  return emsa_launch_status();
}

// 64-bit words of the (y > 0) bit mask of a tensor: V words per 64 channel vectors (V = 4 fp32 or 8
// 16-bit channels per lane); the bound below covers both groupings
extern "C" int64_t emsa_relu_mask_words(int64_t elements) {
  return ((elements / 4 + 63) / 64) * 4 + 8;
}
// 32-bit element indices (maxpool_fwd_kernel's `I`): the largest tensor of the launch has < 2^31
// elements (index + one grid stride cannot wrap)
static bool idx32_ok(long max_elements) {
  static const bool on = [] {                     // EMSA_IDX32=0: 64-bit indices everywhere (A/B runs)
    const char* e = getenv("EMSA_IDX32");
    return !e || atoi(e) != 0;
  }();
  return on && max_elements > 0 && max_elements < (1L << 31);
}
// fast forms of the BatchNorm passes (bn_act_fwd_fast_kernel): log2 of the channel-vector count when
// it is a power of two <= 256 and the indices fit 32 bits, else -1 (general kernels).
// EMSA_BN_FAST=0: always the general kernels (A/B runs)
static bool bn_fast_on() {
  static const bool on = [] {
    const char* e = getenv("EMSA_BN_FAST");
    return !e || atoi(e) != 0;
  }();
  return on;
}
static int bn_fast_log2(int cvn, long totalv, long hw) {
  if (!bn_fast_on() || cvn < 1 || cvn > kThreads || (cvn & (cvn - 1)) != 0) return -1;
  if (totalv < 1 || totalv >= (1L << 31) || hw < 1 || hw >= (1L << 31)) return -1;
  return __builtin_ctz((unsigned)cvn);
}
// channel count admissible for the vectorised BatchNorm passes of storage type T
template <typename T> static bool cv_ok(int c) { return c4_ok(c) && c % VecIO<T>::V == 0; }

template <typename T>
static int bn_act_fwd_impl(const T* x, T* y, const float* scale, const float* shift, const float* drop, const T* residual, int32_t n_img, int64_t hw, int32_t c, int32_t act, uint64_t* mask_bits, void* stream) {
  if (!x || !y || !scale || !shift) return EMSA_E_ARG;
  if (!cv_ok<T>(c)) return EMSA_E_SHAPE;
  constexpr int V = VecIO<T>::V;
  const long totalv = (long)n_img * hw * (c / V);
  const int lg = bn_fast_log2(c / V, totalv, hw);
  if (lg >= 0) {
    const dim3 grid(grid_for(totalv)), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
#define EMSA_FWD_FAST(D, R)                                                                        \
  hipLaunchKernelGGL((bn_act_fwd_fast_kernel<T, D, R>), grid, block, 0, st, x, y, scale, shift,   \
                     drop, residual, (uint32_t)hw, lg, (uint32_t)totalv, act, mask_bits)
    if (drop) {
      if (residual) EMSA_FWD_FAST(true, true); else EMSA_FWD_FAST(true, false);
    } else {
      if (residual) EMSA_FWD_FAST(false, true); else EMSA_FWD_FAST(false, false);
    }
#undef EMSA_FWD_FAST
    return emsa_launch_status();
  }
  hipLaunchKernelGGL((bn_act_fwd_kernel<T>), dim3(grid_for(totalv)), dim3(kThreads), 0,
                     (hipStream_t)stream, x, y, scale, shift, drop, residual, (long)hw, c / V,
                     totalv, act, mask_bits);
  return emsa_launch_status();
}
extern "C" int emsa_bn_act_fwd(const float* x, float* y, const float* scale, const float* shift, const float* drop, const float* residual, int32_t n_img, int64_t hw, int32_t c, int32_t act, uint64_t* mask_bits, void* stream) {
  return bn_act_fwd_impl<float>(x, y, scale, shift, drop, residual, n_img, hw, c, act, mask_bits, stream);
}
extern "C" int emsa_bn_act_fwd_t(int32_t dtype, const void* x, void* y, const float* scale, const float* shift, const float* drop, const void* residual, int32_t n_img, int64_t hw, int32_t c, int32_t act, uint64_t* mask_bits, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { return bn_act_fwd_impl<float>((const float*)x, (float*)y, scale, shift, drop, (const float*)residual, n_img, hw, c, act, mask_bits, stream); }
    case EMSA_DT_BF16: { return bn_act_fwd_impl<emsa_bf16>((const emsa_bf16*)x, (emsa_bf16*)y, scale, shift, drop, (const emsa_bf16*)residual, n_img, hw, c, act, mask_bits, stream); }
    case EMSA_DT_F16: { return bn_act_fwd_impl<emsa_f16>((const emsa_f16*)x, (emsa_f16*)y, scale, shift, drop, (const emsa_f16*)residual, n_img, hw, c, act, mask_bits, stream); }
    default: return EMSA_E_ARG;
  }
}

// number of partial rows (= blocks of the reduce kernel) for `pixels` pixels of `c` channels:
// ~8 K floats per block so that also the low-resolution, wide tensors (C=256/512 at /16, /32)
// fill the chip (they ran at 0.7-2.7 TB/s with one block per 256 pixels)
static int bn_bwd_rows_for(long pixels, int c) {
  long r = (pixels * c + 8191) / 8192;
  if (r > pixels) r = pixels;
  if (r < 1) r = 1;
  static const long cap = [] {                    // EMSA_BN_REDUCE_ROWS: tuning runs (<= 1024)
    const char* e = getenv("EMSA_BN_REDUCE_ROWS");
    const long v = e ? atol(e) : 0;
    return v >= 16 && v <= 1024 ? v : 1024L;
  }();
  if (r > cap) r = cap;
  return (int)r;
}
// rows to ALLOCATE: the per-workgroup partial rows plus kBwdSlices rows of level-1 slice sums
extern "C" int emsa_bn_bwd_rows(int64_t pixels, int32_t c) {
  return bn_bwd_rows_for((long)pixels, c) + kBwdSlices;
}

template <typename T>
static int bn_bwd_reduce_impl(const T* dy, const T* y, const uint64_t* mask_bits, const T* x, const float* save_mean, const float* save_invstd, const float* drop, int32_t n_img, int64_t hw, int32_t c, int32_t act, float* partial, void* stream, const float* mask_scale = nullptr, const float* mask_shift = nullptr) {
  if (!dy || !x || !save_mean || !save_invstd || !partial) return EMSA_E_ARG;
  if (act == EMSA_ACT_RELU && !y && !mask_bits && !mask_scale) return EMSA_E_ARG;
  if ((mask_scale == nullptr) != (mask_shift == nullptr)) return EMSA_E_ARG;
  if (!cv_ok<T>(c)) return EMSA_E_SHAPE;
  const long pixels = (long)n_img * hw;
  const int rows = bn_bwd_rows_for(pixels, c);
  const int cvn = c / VecIO<T>::V, lanes = kThreads / cvn;
  const size_t lds = (size_t)2 * lanes * c * sizeof(float);
  if (bn_fast_on() && cvn <= kThreads && pixels * cvn < (1L << 31) && hw < (1L << 31)) {
    const dim3 grid(rows), block(kThreads);
    hipStream_t st = (hipStream_t)stream;
    const int mask = act == EMSA_ACT_RELU ? (mask_scale ? 3 : mask_bits ? 1 : 2) : 0;
#define EMSA_RED_FAST(M, D)                                                                        \
  hipLaunchKernelGGL((bn_bwd_reduce_fast_kernel<T, M, D>), grid, block, lds, st, dy, y, mask_bits, \
                     x, save_mean, save_invstd, drop, (uint32_t)pixels, (uint32_t)hw, cvn,        \
                     rows + kBwdSlices, partial, mask_scale, mask_shift)
    if (mask == 3) {
      if (drop) return EMSA_E_ARG;
      EMSA_RED_FAST(3, false);
    } else if (drop) {
      if (mask == 1) EMSA_RED_FAST(1, true); else if (mask == 2) EMSA_RED_FAST(2, true); else EMSA_RED_FAST(0, true);
    } else {
      if (mask == 1) EMSA_RED_FAST(1, false); else if (mask == 2) EMSA_RED_FAST(2, false); else EMSA_RED_FAST(0, false);
    }
#undef EMSA_RED_FAST
    return emsa_launch_status();
  }
  if (mask_scale) return EMSA_E_SHAPE;            // (the general loops have no recomputed-mask form)
  if (cvn > kThreads) {
    hipLaunchKernelGGL((bn_bwd_reduce_wide_kernel<T>), dim3(rows, (cvn + kThreads - 1) / kThreads),
                       dim3(kThreads), 0, (hipStream_t)stream, dy, y, mask_bits, x, save_mean,
                       save_invstd, drop, pixels, (long)hw, cvn, act, rows + kBwdSlices, partial);
    return emsa_launch_status();
  }
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), dim3(rows), dim3(kThreads), lds, (hipStream_t)stream,
                     dy, y, mask_bits, x, save_mean, save_invstd, drop, pixels, (long)hw, cvn, act,
                     rows + kBwdSlices, partial);
  return emsa_launch_status();
}
extern "C" int emsa_bn_bwd_reduce(const float* dy, const float* y, const uint64_t* mask_bits, const float* x, const float* save_mean, const float* save_invstd, const float* drop, int32_t n_img, int64_t hw, int32_t c, int32_t act, float* partial, void* stream) {
  return bn_bwd_reduce_impl<float>(dy, y, mask_bits, x, save_mean, save_invstd, drop, n_img, hw, c, act, partial, stream);
}
extern "C" int emsa_bn_bwd_reduce_t(int32_t dtype, const void* dy, const void* y, const uint64_t* mask_bits, const void* x, const float* save_mean, const float* save_invstd, const float* drop, int32_t n_img, int64_t hw, int32_t c, int32_t act, float* partial, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { return bn_bwd_reduce_impl<float>((const float*)dy, (const float*)y, mask_bits, (const float*)x, save_mean, save_invstd, drop, n_img, hw, c, act, partial, stream); }
    case EMSA_DT_BF16: { return bn_bwd_reduce_impl<emsa_bf16>((const emsa_bf16*)dy, (const emsa_bf16*)y, mask_bits, (const emsa_bf16*)x, save_mean, save_invstd, drop, n_img, hw, c, act, partial, stream); }
    case EMSA_DT_F16: { return bn_bwd_reduce_impl<emsa_f16>((const emsa_f16*)dy, (const emsa_f16*)y, mask_bits, (const emsa_f16*)x, save_mean, save_invstd, drop, n_img, hw, c, act, partial, stream); }
    default: return EMSA_E_ARG;
  }
}

template <typename T>
static int bn_bwd_apply_impl(const T* dy, const T* y, const uint64_t* mask_bits, const T* x, const float* gamma, const float* save_mean, const float* save_invstd, const float* drop, float* partial, int32_t rows_alloc, int32_t n_img, int64_t hw, int32_t c, int32_t act, int32_t train, T* dx, T* dres, float* dgamma, float* dbeta, void* stream, int32_t rows_given = -1, const float* mask_scale = nullptr, const float* mask_shift = nullptr) {
  if (!dy || !x || !gamma || !save_mean || !save_invstd || !partial || !dx || !dgamma || !dbeta)
    return EMSA_E_ARG;
  if (act == EMSA_ACT_RELU && !y && !mask_bits && !mask_scale) return EMSA_E_ARG;
  if ((mask_scale == nullptr) != (mask_shift == nullptr)) return EMSA_E_ARG;
  if (!cv_ok<T>(c)) return EMSA_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const long pixels = (long)n_img * hw;
  // rows_given: the partial sums come from a convolution epilogue (one row per pixel tile of that
  // launch) instead of emsa_bn_bwd_reduce
  const int rows = rows_given >= 0 ? rows_given : bn_bwd_rows_for(pixels, c);
  if (rows < 1 || rows_alloc != rows + kBwdSlices) return EMSA_E_ARG;
  hipLaunchKernelGGL(bn_bwd_sum_kernel, dim3((c + 31) / 32, kBwdSlices), dim3(256), 0, st, partial,
                     rows, rows_alloc, c);
  constexpr int V = VecIO<T>::V;
  const long totalv = pixels * (c / V);
  // four workgroups per CU (the other streaming kernels: eight): measured -0.3 ms per step
  static const int ap_cap = [] {                  // EMSA_BN_APPLY_WGS: tuning runs
    const char* e = getenv("EMSA_BN_APPLY_WGS");
    const int v = e ? atoi(e) : 0;
    return v >= 64 ? v : 256 * 4;
  }();
  int ap_grid = grid_for(totalv);
  if (ap_grid > ap_cap) ap_grid = ap_cap;
  const int lg = bn_fast_log2(c / V, totalv, hw);
  if (lg >= 0) {
    const dim3 grid(ap_grid), block(kThreads);
    const size_t lds = (size_t)2 * c * sizeof(float);
    const int mask = act == EMSA_ACT_RELU ? (mask_scale ? 3 : mask_bits ? 1 : 2) : 0;
    const float inv_count = 1.0f / (float)pixels;
#define EMSA_APP_FAST(M, D, TR)                                                                     \
  hipLaunchKernelGGL((bn_bwd_apply_fast_kernel<T, M, D, TR>), grid, block, lds, st, dy, y,         \
                     mask_bits, x, gamma, save_mean, save_invstd, drop, partial, rows, rows_alloc,  \
                     dbeta, dgamma, (uint32_t)hw, lg, (uint32_t)totalv, inv_count, dx, dres,        \
                     mask_scale, mask_shift)
#define EMSA_APP_FAST_M(D, TR)                                                                      \
  do {                                                                                              \
    if (mask == 1) EMSA_APP_FAST(1, D, TR); else if (mask == 2) EMSA_APP_FAST(2, D, TR);            \
    else EMSA_APP_FAST(0, D, TR);                                                                   \
  } while (0)
    if (mask == 3) {
      if (drop || !train) return EMSA_E_ARG;      // (bn1 of an NBt1D block: batch statistics, no dropout)
      EMSA_APP_FAST(3, false, true);
    } else if (drop) {
      if (train) EMSA_APP_FAST_M(true, true); else EMSA_APP_FAST_M(true, false);
    } else {
      if (train) EMSA_APP_FAST_M(false, true); else EMSA_APP_FAST_M(false, false);
    }
#undef EMSA_APP_FAST_M
#undef EMSA_APP_FAST
    return emsa_launch_status();
  }
  if (mask_scale) return EMSA_E_SHAPE;
  hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(ap_grid), dim3(kThreads),
                     (size_t)2 * c * sizeof(float), st, dy, y, mask_bits, x, gamma, save_mean,
                     save_invstd, drop, partial, rows, rows_alloc, dbeta, dgamma, (long)hw, c / V, totalv,
                     1.0f / (float)pixels, act, train, dx, dres);
  return emsa_launch_status();
}
extern "C" int emsa_bn_bwd_apply(const float* dy, const float* y, const uint64_t* mask_bits, const float* x, const float* gamma, const float* save_mean, const float* save_invstd, const float* drop, float* partial, int32_t rows_alloc, int32_t n_img, int64_t hw, int32_t c, int32_t act, int32_t train, float* dx, float* dres, float* dgamma, float* dbeta, void* stream) {
  return bn_bwd_apply_impl<float>(dy, y, mask_bits, x, gamma, save_mean, save_invstd, drop, partial, rows_alloc, n_img, hw, c, act, train, dx, dres, dgamma, dbeta, stream);
}
extern "C" int emsa_bn_bwd_apply_t(int32_t dtype, const void* dy, const void* y, const uint64_t* mask_bits, const void* x, const float* gamma, const float* save_mean, const float* save_invstd, const float* drop, float* partial, int32_t rows_alloc, int32_t n_img, int64_t hw, int32_t c, int32_t act, int32_t train, void* dx, void* dres, float* dgamma, float* dbeta, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { return bn_bwd_apply_impl<float>((const float*)dy, (const float*)y, mask_bits, (const float*)x, gamma, save_mean, save_invstd, drop, partial, rows_alloc, n_img, hw, c, act, train, (float*)dx, (float*)dres, dgamma, dbeta, stream); }
    case EMSA_DT_BF16: { return bn_bwd_apply_impl<emsa_bf16>((const emsa_bf16*)dy, (const emsa_bf16*)y, mask_bits, (const emsa_bf16*)x, gamma, save_mean, save_invstd, drop, partial, rows_alloc, n_img, hw, c, act, train, (emsa_bf16*)dx, (emsa_bf16*)dres, dgamma, dbeta, stream); }
    case EMSA_DT_F16: { return bn_bwd_apply_impl<emsa_f16>((const emsa_f16*)dy, (const emsa_f16*)y, mask_bits, (const emsa_f16*)x, gamma, save_mean, save_invstd, drop, partial, rows_alloc, n_img, hw, c, act, train, (emsa_f16*)dx, (emsa_f16*)dres, dgamma, dbeta, stream); }
    default: return EMSA_E_ARG;
  }
}
// BatchNorm backward from per-tile sums a data-gradient epilogue already produced
// (emsa_conv1d_wino_bnb / emsa_conv_igemm_bnb_t): g = the masked gradient that kernel stored,
// partial = float[2][rows + 16][c] with rows [0, rows) = (sum g, sum g * xhat) per tile.  Merges
// the rows and applies  dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)); dgamma, dbeta.
extern "C" int emsa_bn_bwd_apply_rows_t(int32_t dtype, const void* g, const void* x, const float* gamma, const float* save_mean, const float* save_invstd, float* partial, int32_t rows, int32_t n_img, int64_t hw, int32_t c, int32_t train, void* dx, float* dgamma, float* dbeta, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: return bn_bwd_apply_impl<float>((const float*)g, nullptr, nullptr, (const float*)x, gamma, save_mean, save_invstd, nullptr, partial, rows + kBwdSlices, n_img, hw, c, EMSA_ACT_NONE, train, (float*)dx, (float*)nullptr, dgamma, dbeta, stream, rows);
    case EMSA_DT_BF16: return bn_bwd_apply_impl<emsa_bf16>((const emsa_bf16*)g, nullptr, nullptr, (const emsa_bf16*)x, gamma, save_mean, save_invstd, nullptr, partial, rows + kBwdSlices, n_img, hw, c, EMSA_ACT_NONE, train, (emsa_bf16*)dx, (emsa_bf16*)nullptr, dgamma, dbeta, stream, rows);
    case EMSA_DT_F16: return bn_bwd_apply_impl<emsa_f16>((const emsa_f16*)g, nullptr, nullptr, (const emsa_f16*)x, gamma, save_mean, save_invstd, nullptr, partial, rows + kBwdSlices, n_img, hw, c, EMSA_ACT_NONE, train, (emsa_f16*)dx, (emsa_f16*)nullptr, dgamma, dbeta, stream, rows);
    default: return EMSA_E_ARG;
  }
}

// BatchNorm + ReLU backward with the ReLU decisions RECOMPUTED from the BatchNorm's input:
// mask = (x * mask_scale[c] + mask_shift[c]) > 0, the forward pass's own folded scale / shift (the
// normalised tensor was never written: emsa_conv1d_rs_inbn_t took x itself).  Batch statistics, no
// Dropout2d (the NBt1D block's bn1, ref emsanet/model.py:47-58).  Same partial-row contract as
// emsa_bn_bwd_reduce_t / emsa_bn_bwd_apply_t; power-of-two channel-vector counts only (EMSA_E_SHAPE).
extern "C" int emsa_bn_bwd_reduce_aff_t(int32_t dtype, const void* dy, const void* x, const float* save_mean, const float* save_invstd, const float* mask_scale, const float* mask_shift, int32_t n_img, int64_t hw, int32_t c, float* partial, void* stream) {
  if (!mask_scale || !mask_shift) return EMSA_E_ARG;
  switch (dtype) {
    case EMSA_DT_F32: return bn_bwd_reduce_impl<float>((const float*)dy, nullptr, nullptr, (const float*)x, save_mean, save_invstd, nullptr, n_img, hw, c, EMSA_ACT_RELU, partial, stream, mask_scale, mask_shift);
    case EMSA_DT_BF16: return bn_bwd_reduce_impl<emsa_bf16>((const emsa_bf16*)dy, nullptr, nullptr, (const emsa_bf16*)x, save_mean, save_invstd, nullptr, n_img, hw, c, EMSA_ACT_RELU, partial, stream, mask_scale, mask_shift);
    default: return EMSA_E_ARG;
  }
}
extern "C" int emsa_bn_bwd_apply_aff_t(int32_t dtype, const void* dy, const void* x, const float* gamma, const float* save_mean, const float* save_invstd, const float* mask_scale, const float* mask_shift, float* partial, int32_t rows_alloc, int32_t n_img, int64_t hw, int32_t c, void* dx, float* dgamma, float* dbeta, void* stream) {
  if (!mask_scale || !mask_shift) return EMSA_E_ARG;
  switch (dtype) {
    case EMSA_DT_F32: return bn_bwd_apply_impl<float>((const float*)dy, nullptr, nullptr, (const float*)x, gamma, save_mean, save_invstd, nullptr, partial, rows_alloc, n_img, hw, c, EMSA_ACT_RELU, 1, (float*)dx, (float*)nullptr, dgamma, dbeta, stream, -1, mask_scale, mask_shift);
    case EMSA_DT_BF16: return bn_bwd_apply_impl<emsa_bf16>((const emsa_bf16*)dy, nullptr, nullptr, (const emsa_bf16*)x, gamma, save_mean, save_invstd, nullptr, partial, rows_alloc, n_img, hw, c, EMSA_ACT_RELU, 1, (emsa_bf16*)dx, (emsa_bf16*)nullptr, dgamma, dbeta, stream, -1, mask_scale, mask_shift);
    default: return EMSA_E_ARG;
  }
}

extern "C" int emsa_dropout2d_mask(float* mask, int32_t n, int32_t c, float p, uint32_t seed,
                                   uint32_t layer_id, void* stream) {
  if (!mask) return EMSA_E_ARG;
  hipLaunchKernelGGL(dropout2d_mask_kernel, dim3((n * c + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, mask, n, c, p, seed, layer_id, (const uint32_t*)nullptr);
  return emsa_launch_status();
}
extern "C" int emsa_dropout2d_mask_dev(float* mask, int32_t n, int32_t c, float p,
                                       const uint32_t* state, uint32_t layer_id, void* stream) {
  if (!mask || !state) return EMSA_E_ARG;
  hipLaunchKernelGGL(dropout2d_mask_kernel, dim3((n * c + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, mask, n, c, p, 0u, layer_id, state);
  return emsa_launch_status();
}
// masks of `n_jobs` Dropout2d layers for a batch of `n` images: masks[jobs[j].offset + img * c + ch]
// (same values as emsa_dropout2d_mask per layer); `state` = device {seed, step} or NULL (`seed`)
extern "C" int emsa_dropout2d_mask_batch(float* masks, const EmsaDropoutJob* jobs_device,
                                         int32_t n_jobs, int32_t n, int32_t max_c, uint32_t seed,
                                         const uint32_t* state, void* stream) {
  if (!masks || !jobs_device) return EMSA_E_ARG;
  if (n_jobs < 1 || n < 1 || max_c < 1) return EMSA_E_SHAPE;
  int bx = (n * max_c + 255) / 256;
  if (bx > 64) bx = 64;
  hipLaunchKernelGGL(dropout2d_mask_batch_kernel, dim3(bx, n_jobs), dim3(256), 0,
                     (hipStream_t)stream, masks, jobs_device, n, seed, state);
  return emsa_launch_status();
}
extern "C" int emsa_u32_add(uint32_t* counter, uint32_t value, void* stream) {
  if (!counter) return EMSA_E_ARG;
  hipLaunchKernelGGL(u32_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, value);
  return emsa_launch_status();
}

template <typename T>
static int maxpool3x3s2_fwd_impl(const T* x, T* y, int8_t* idx, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  if (!x || !y || !idx) return EMSA_E_ARG;
  if (!c4_ok(c)) return EMSA_E_SHAPE;
  const long total = (long)n * ((h + 1) / 2) * ((w + 1) / 2) * (c / 4);
  if (idx32_ok((long)n * h * w * c))
    hipLaunchKernelGGL((maxpool_fwd_kernel<T, uint32_t>), dim3(grid_for(total)), dim3(kThreads), 0,
                       (hipStream_t)stream, x, y, idx, n, h, w, c / 4);
  else
    hipLaunchKernelGGL((maxpool_fwd_kernel<T, long>), dim3(grid_for(total)), dim3(kThreads), 0,
                       (hipStream_t)stream, x, y, idx, n, h, w, c / 4);
  return emsa_launch_status();
}
extern "C" int emsa_maxpool3x3s2_fwd(const float* x, float* y, int8_t* idx, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  return maxpool3x3s2_fwd_impl<float>(x, y, idx, n, h, w, c, stream);
}
extern "C" int emsa_maxpool3x3s2_fwd_t(int32_t dtype, const void* x, void* y, int8_t* idx, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { return maxpool3x3s2_fwd_impl<float>((const float*)x, (float*)y, idx, n, h, w, c, stream); }
    case EMSA_DT_BF16: { return maxpool3x3s2_fwd_impl<emsa_bf16>((const emsa_bf16*)x, (emsa_bf16*)y, idx, n, h, w, c, stream); }
    case EMSA_DT_F16: { return maxpool3x3s2_fwd_impl<emsa_f16>((const emsa_f16*)x, (emsa_f16*)y, idx, n, h, w, c, stream); }
    default: return EMSA_E_ARG;
  }
}
template <typename T>
static int maxpool3x3s2_bwd_impl(const T* dy, const int8_t* idx, T* dx, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  if (!dy || !dx || !idx) return EMSA_E_ARG;
  if (!cv_ok<T>(c)) return EMSA_E_SHAPE;
  constexpr int V = VecIO<T>::V;
  const long total = (long)n * ((h + 1) / 2) * ((w + 1) / 2) * (c / V);
  if (idx32_ok((long)n * h * w * c))
    hipLaunchKernelGGL((maxpool_bwd_kernel<T, uint32_t>), dim3(grid_for(total)), dim3(kThreads), 0,
                       (hipStream_t)stream, dy, idx, dx, n, h, w, c / V);
  else
    hipLaunchKernelGGL((maxpool_bwd_kernel<T, long>), dim3(grid_for(total)), dim3(kThreads), 0,
                       (hipStream_t)stream, dy, idx, dx, n, h, w, c / V);
  return emsa_launch_status();
}
extern "C" int emsa_maxpool3x3s2_bwd(const float* dy, const int8_t* idx, float* dx, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  return maxpool3x3s2_bwd_impl<float>(dy, idx, dx, n, h, w, c, stream);
}
extern "C" int emsa_maxpool3x3s2_bwd_t(int32_t dtype, const void* dy, const int8_t* idx, void* dx, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { return maxpool3x3s2_bwd_impl<float>((const float*)dy, idx, (float*)dx, n, h, w, c, stream); }
    case EMSA_DT_BF16: { return maxpool3x3s2_bwd_impl<emsa_bf16>((const emsa_bf16*)dy, idx, (emsa_bf16*)dx, n, h, w, c, stream); }
    case EMSA_DT_F16: { return maxpool3x3s2_bwd_impl<emsa_f16>((const emsa_f16*)dy, idx, (emsa_f16*)dx, n, h, w, c, stream); }
    default: return EMSA_E_ARG;
  }
}

// pixel chunks per image: 512 pixels per workgroup (at most 64 chunks) -- unless that leaves the launch
// with a handful of workgroups each walking its chunk a few rows at a time (batch-1 inference: 300
// pixels x 512 channels was ONE workgroup of 4 row lanes x 75 dependent steps, 26 us in a graph whose
// other nodes take 5; the /2 map 15.7 us on 128 workgroups): then 64-pixel chunks, at most 256.
// The rule depends on the BATCH (n * splits <= 64), so the order in which one sample's pixels are
// summed differs between batch sizes wherever the coarse partition leaves <= 64 workgroups: batch-1
// inference at every stage, and also the /32 maps of a batch-32 step (300 pixels: 1 chunk coarse,
// 32 workgroups -> 5 chunks of 64).  Per-sample results therefore depend on the batch size at fp32
// rounding level: measured 1.0e-4 of the tensor magnitude on the tanh offset map between batch 1 and
// batch 32 (<= 4e-5 on the other outputs; tests/test_model_gpu.py::
// test_full_size_batch_consistency_and_determinism gates 2e-4).  (A rule of the map size alone --
// every sample summed in the same order whatever its batch -- was tried: it re-partitions the /4 ..
// /16 sums of the training shapes too, and the bf16 train-mode gates of tests/test_model16_gpu.py,
// which sit on one draw of a chaotic system, moved: an SE hidden unit flipped against the oracle's.
// Those partitions stay what they were)
// emsa_set_batch_invariant(1): the partition becomes a function of the map size alone (always the
// 512-pixel rule), so a sample's sums -- and with them its whole eval forward -- are the same bits
// whatever batch it sits in; opt-in (slower at batch 1), used by tests/test_timed_size_gpu.py to
// compare the gradients of one bs-32 step with the sum over four bs-8 steps on identical ReLU
// decisions.
static std::atomic<int> g_batch_invariant{0};
static int channel_splits(long hw, int n) {
  int splits = (int)((hw + 511) / 512);
  if (splits > 64) splits = 64;
  if ((long)n * splits <= 64 && !g_batch_invariant.load(std::memory_order_relaxed)) {
    splits = (int)((hw + 63) / 64);
    if (splits > 256) splits = 256;
  }
  if (splits < 1) splits = 1;
  return splits;
}

extern "C" int emsa_set_batch_invariant(int on) {
  return g_batch_invariant.exchange(on ? 1 : 0, std::memory_order_relaxed);
}

template <typename T>
static int channel_dot(const T* a, const T* b, float* out, float* ws, int n, long hw,
                       int c, float scale, hipStream_t st) {
  if (!cv_ok<T>(c)) return EMSA_E_SHAPE;
  const int splits = channel_splits(hw, n);
  const int cvn = c / VecIO<T>::V;
  // one thread column per channel vector: wider tensors (> 1024 fp32 / 2048 16-bit channels) would
  // leave `lanes` = 0 and every sum silently zero -- they take the wide kernel
  if (cvn > kThreads) {
    const dim3 grid(n * splits, (cvn + kThreads - 1) / kThreads);
    if (b)
      hipLaunchKernelGGL((channel_dot_wide_kernel<T, true>), grid, dim3(kThreads), 0, st, a, b, ws, hw,
                         cvn, splits);
    else
      hipLaunchKernelGGL((channel_dot_wide_kernel<T, false>), grid, dim3(kThreads), 0, st, a, b, ws, hw,
                         cvn, splits);
    hipLaunchKernelGGL(channel_dot_finish_kernel, dim3((n * c + 255) / 256), dim3(256), 0, st, ws,
                       out, n, c, splits, scale);
    return emsa_launch_status();
  }
  const int lanes = kThreads / cvn;
  const size_t lds = (size_t)lanes * c * sizeof(float);
  if (b)
    hipLaunchKernelGGL((channel_dot_kernel<T, true>), dim3(n * splits), dim3(kThreads), lds, st, a, b,
                       ws, hw, cvn, splits);
  else
    hipLaunchKernelGGL((channel_dot_kernel<T, false>), dim3(n * splits), dim3(kThreads), lds, st, a, b,
                       ws, hw, cvn, splits);
  hipLaunchKernelGGL(channel_dot_finish_kernel, dim3((n * c + 255) / 256), dim3(256), 0, st, ws,
                     out, n, c, splits, scale);
  return emsa_launch_status();
}

extern "C" int emsa_channel_ws_floats(int32_t n, int64_t hw, int32_t c) {
  return n * channel_splits((long)hw, n) * c;
}
extern "C" int emsa_channel_mean(const float* x, float* gap, float* ws, int32_t n, int64_t hw,
                                 int32_t c, void* stream) {
  if (!x || !gap || !ws) return EMSA_E_ARG;
  return channel_dot<float>(x, nullptr, gap, ws, n, (long)hw, c, 1.0f / (float)hw,
                            (hipStream_t)stream);
}
extern "C" int emsa_channel_mean_t(int32_t dtype, const void* x, float* gap, float* ws, int32_t n,
                                   int64_t hw, int32_t c, void* stream) {
  if (!x || !gap || !ws) return EMSA_E_ARG;
  EMSA_DISPATCH_DTYPE(dtype, T, return channel_dot<T>((const T*)x, (const T*)nullptr, gap, ws, n,
                                                      (long)hw, c, 1.0f / (float)hw,
                                                      (hipStream_t)stream));
  return EMSA_E_ARG;
}
extern "C" int emsa_se_scale_bwd_reduce(const float* dout, const float* x, float* ds, float* ws,
                                        int32_t n, int64_t hw, int32_t c, void* stream) {
  if (!dout || !x || !ds || !ws) return EMSA_E_ARG;
  return channel_dot<float>(dout, x, ds, ws, n, (long)hw, c, 1.0f, (hipStream_t)stream);
}
extern "C" int emsa_se_scale_bwd_reduce_t(int32_t dtype, const void* dout, const void* x, float* ds,
                                          float* ws, int32_t n, int64_t hw, int32_t c,
                                          void* stream) {
  if (!dout || !x || !ds || !ws) return EMSA_E_ARG;
  EMSA_DISPATCH_DTYPE(dtype, T, return channel_dot<T>((const T*)dout, (const T*)x, ds, ws, n,
                                                      (long)hw, c, 1.0f, (hipStream_t)stream));
  return EMSA_E_ARG;
}

extern "C" int emsa_se_mlp_fwd(const float* gap, const float* w1, const float* b1,
                               const float* w2, const float* b2, float* hid, float* s, int32_t n,
                               int32_t c, int32_t cr, void* stream) {
  if (!gap || !w1 || !b1 || !w2 || !b2 || !hid || !s) return EMSA_E_ARG;
  hipLaunchKernelGGL(se_mlp_fwd_kernel, dim3(n), dim3(256), (size_t)(c + cr) * sizeof(float),
                     (hipStream_t)stream, gap, w1, b1, w2, b2, hid, s, c, cr);
  return emsa_launch_status();
}
template <typename T>
static int se_pair_fwd(const T* xa, const T* xb, float* ws, const float* const* wts, float* gap,
                       float* hid, float* s, int n, long hw, int c, int cr, hipStream_t st) {
  if (!cv_ok<T>(c)) return EMSA_E_SHAPE;
  const int splits = channel_splits(hw, n);
  const int cvn = c / VecIO<T>::V;
  if (cvn > kThreads) {
    hipLaunchKernelGGL((channel_dot_wide_kernel<T, false>),
                       dim3(2 * n * splits, (cvn + kThreads - 1) / kThreads), dim3(kThreads), 0, st, xa,
                       (const T*)nullptr, ws, hw, cvn, splits, xb, n * splits);
  } else {
    const int lanes = kThreads / cvn;
    const size_t lds = (size_t)lanes * c * sizeof(float);
    hipLaunchKernelGGL((channel_dot_kernel<T, false>), dim3(2 * n * splits), dim3(kThreads), lds, st, xa,
                       (const T*)nullptr, ws, hw, cvn, splits, xb, n * splits);
  }
  hipLaunchKernelGGL(se_mlp_pair_fwd_kernel, dim3(2 * n), dim3(256),
                     (size_t)(c + cr) * sizeof(float), st, ws, splits, 1.0f / (float)hw, wts[0],
                     wts[1], wts[2], wts[3], wts[4], wts[5], wts[6], wts[7], gap, hid, s, n, c, cr);
  return emsa_launch_status();
}
extern "C" int emsa_se_pair_fwd_t(int32_t dtype, const void* xa, const void* xb, float* ws,
                                  const float* w1a, const float* b1a, const float* w2a,
                                  const float* b2a, const float* w1b, const float* b1b,
                                  const float* w2b, const float* b2b, float* gap, float* hid,
                                  float* s, int32_t n, int64_t hw, int32_t c, int32_t cr,
                                  void* stream) {
  if (!xa || !xb || !ws || !w1a || !b1a || !w2a || !b2a || !w1b || !b1b || !w2b || !b2b || !gap ||
      !hid || !s)
    return EMSA_E_ARG;
  const float* wts[8] = {w1a, b1a, w2a, b2a, w1b, b1b, w2b, b2b};
  EMSA_DISPATCH_DTYPE(dtype, T, return se_pair_fwd<T>((const T*)xa, (const T*)xb, ws, wts, gap, hid,
                                                      s, n, (long)hw, c, cr, (hipStream_t)stream));
  return EMSA_E_ARG;
}
extern "C" int emsa_se_mlp_bwd(const float* gap, const float* w1, const float* w2,
                               const float* hid, const float* s, const float* ds, float* dgap,
                               float* dw1, float* db1, float* dw2, float* db2, int32_t n,
                               int32_t c, int32_t cr, void* stream) {
  if (!gap || !w1 || !w2 || !hid || !s || !ds || !dgap || !dw1 || !db1 || !dw2 || !db2)
    return EMSA_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  if (n <= c && cr <= 32 && cr >= 1) {
    float* dz1 = dw2;                          // scratch until the last launch overwrites it
    const size_t lds = (size_t)(c + (256 / cr) * cr + cr) * sizeof(float);
    hipLaunchKernelGGL(se_mlp_bwd_a_kernel, dim3(n), dim3(256), lds, st, w1, w2, hid, s, ds, dgap,
                       dz1, c, cr);
    const int wblocks = (c * cr + 255) / 256;
    hipLaunchKernelGGL(se_mlp_bwd_w_kernel<0>, dim3(wblocks + (cr + 255) / 256), dim3(256), 0, st,
                       gap, hid, s, ds, dz1, dw1, db1, n, c, cr, wblocks);
    hipLaunchKernelGGL(se_mlp_bwd_w_kernel<1>, dim3(wblocks + (c + 255) / 256), dim3(256), 0, st,
                       gap, hid, s, ds, dz1, dw2, db2, n, c, cr, wblocks);
    return emsa_launch_status();
  }
  if ((size_t)n * cr * sizeof(float) > 60000) return EMSA_E_SHAPE;
  hipLaunchKernelGGL(se_mlp_bwd_kernel, dim3(8), dim3(256), (size_t)n * cr * sizeof(float), st, gap,
                     w1, w2, hid, s, ds, dgap, dw1, db1, dw2, db2, n, c, cr);
  return emsa_launch_status();
}

template <typename T>
static int se_scale_add_fwd_impl(const T* a, const float* sa, const T* b, const float* sb, T* out, int32_t n, int64_t hw, int32_t c, void* stream) {
  if (!a || !sa || !out || ((b == nullptr) != (sb == nullptr))) return EMSA_E_ARG;
  if (!cv_ok<T>(c)) return EMSA_E_SHAPE;
  const long totalv = (long)n * hw * (c / VecIO<T>::V);
  const dim3 grid(grid_for(totalv)), block(kThreads);
  hipStream_t st = (hipStream_t)stream;
  const int cvn = c / VecIO<T>::V;
#define EMSA_SE_FWD(B, I)                                                                          \
  hipLaunchKernelGGL((se_scale_add_fwd_kernel<T, B, I>), grid, block, 0, st, a, sa, b, sb, out,   \
                     (I)hw, cvn, (I)totalv)
  if (idx32_ok((long)n * hw * c)) {
    if (b) EMSA_SE_FWD(true, uint32_t); else EMSA_SE_FWD(false, uint32_t);
  } else {
    if (b) EMSA_SE_FWD(true, long); else EMSA_SE_FWD(false, long);
  }
#undef EMSA_SE_FWD
  return emsa_launch_status();
}
extern "C" int emsa_se_scale_add_fwd(const float* a, const float* sa, const float* b, const float* sb, float* out, int32_t n, int64_t hw, int32_t c, void* stream) {
  return se_scale_add_fwd_impl<float>(a, sa, b, sb, out, n, hw, c, stream);
}
extern "C" int emsa_se_scale_add_fwd_t(int32_t dtype, const void* a, const float* sa, const void* b, const float* sb, void* out, int32_t n, int64_t hw, int32_t c, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { return se_scale_add_fwd_impl<float>((const float*)a, sa, (const float*)b, sb, (float*)out, n, hw, c, stream); }
    case EMSA_DT_BF16: { return se_scale_add_fwd_impl<emsa_bf16>((const emsa_bf16*)a, sa, (const emsa_bf16*)b, sb, (emsa_bf16*)out, n, hw, c, stream); }
    case EMSA_DT_F16: { return se_scale_add_fwd_impl<emsa_f16>((const emsa_f16*)a, sa, (const emsa_f16*)b, sb, (emsa_f16*)out, n, hw, c, stream); }
    default: return EMSA_E_ARG;
  }
}
template <typename T>
static int se_scale_bwd_apply_impl(const T* dout, const float* s, const float* dgap, const T* dx_extra, T* dx, int32_t n, int64_t hw, int32_t c, void* stream) {
  if (!dout || !s || !dgap || !dx) return EMSA_E_ARG;
  if (!cv_ok<T>(c)) return EMSA_E_SHAPE;
  const long totalv = (long)n * hw * (c / VecIO<T>::V);
  const dim3 grid(grid_for(totalv)), block(kThreads);
  hipStream_t st = (hipStream_t)stream;
  const int cvn = c / VecIO<T>::V;
  const float inv_hw = 1.0f / (float)hw;
#define EMSA_SE_BWD(E, I)                                                                          \
  hipLaunchKernelGGL((se_scale_bwd_apply_kernel<T, E, I>), grid, block, 0, st, dout, s, dgap,     \
                     dx_extra, dx, (I)hw, cvn, (I)totalv, inv_hw)
  if (idx32_ok((long)n * hw * c)) {
    if (dx_extra) EMSA_SE_BWD(true, uint32_t); else EMSA_SE_BWD(false, uint32_t);
  } else {
    if (dx_extra) EMSA_SE_BWD(true, long); else EMSA_SE_BWD(false, long);
  }
#undef EMSA_SE_BWD
  return emsa_launch_status();
}
extern "C" int emsa_se_scale_bwd_apply(const float* dout, const float* s, const float* dgap, const float* dx_extra, float* dx, int32_t n, int64_t hw, int32_t c, void* stream) {
  return se_scale_bwd_apply_impl<float>(dout, s, dgap, dx_extra, dx, n, hw, c, stream);
}
extern "C" int emsa_se_scale_bwd_apply_t(int32_t dtype, const void* dout, const float* s, const float* dgap, const void* dx_extra, void* dx, int32_t n, int64_t hw, int32_t c, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { return se_scale_bwd_apply_impl<float>((const float*)dout, s, dgap, (const float*)dx_extra, (float*)dx, n, hw, c, stream); }
    case EMSA_DT_BF16: { return se_scale_bwd_apply_impl<emsa_bf16>((const emsa_bf16*)dout, s, dgap, (const emsa_bf16*)dx_extra, (emsa_bf16*)dx, n, hw, c, stream); }
    case EMSA_DT_F16: { return se_scale_bwd_apply_impl<emsa_f16>((const emsa_f16*)dout, s, dgap, (const emsa_f16*)dx_extra, (emsa_f16*)dx, n, hw, c, stream); }
    default: return EMSA_E_ARG;
  }
}

template <typename T, typename TO>
static int up2x_dw3x3_fwd_impl(const T* x, const float* wdw, const float* bias, const T* skip, TO* y, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  if (!x || !wdw || !y) return EMSA_E_ARG;
  if (!c4_ok(c)) return EMSA_E_SHAPE;
  const long total = (long)n * 4 * h * w * (c / 4);
  // non-temporal output stores for the wide fp32 maps: measured +4..10 % at 40 / 64 / 128 / 256
  // channels (c40 full resolution 659 -> 633 us), -20..-40 % at 8 channels (32-byte pixels)
  // (profiles/r03_d_pointwise_bench.txt); EMSA_UP2X_NT=0 / 1 forces it off / on
  static const int nt_env = [] {
    const char* e = getenv("EMSA_UP2X_NT");
    return e ? (e[0] == '1' ? 1 : 0) : -1;
  }();
  const bool nt = nt_env >= 0 ? nt_env == 1 : c >= 32;
  const dim3 grid(grid_for(total)), block(kThreads);
  const size_t lds = (size_t)9 * c * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
#define EMSA_UP_FWD(NTF, I)                                                                         \
  do {                                                                                              \
    if (skip)                                                                                       \
      hipLaunchKernelGGL((up2x_dw_fwd_kernel<T, TO, NTF, I, true>), grid, block, lds, st, x, wdw,   \
                         bias, skip, y, n, h, w, c / 4);                                            \
    else                                                                                            \
      hipLaunchKernelGGL((up2x_dw_fwd_kernel<T, TO, NTF, I, false>), grid, block, lds, st, x, wdw,  \
                         bias, skip, y, n, h, w, c / 4);                                            \
  } while (0)
  const bool i32 = idx32_ok((long)n * 4 * h * w * c);
  if (nt && std::is_same<TO, float>::value) {
    if (i32) EMSA_UP_FWD(true, uint32_t); else EMSA_UP_FWD(true, long);
  } else {
    if (i32) EMSA_UP_FWD(false, uint32_t); else EMSA_UP_FWD(false, long);
  }
#undef EMSA_UP_FWD
  return emsa_launch_status();
}
// Twin launch of emsa_up2x_dw3x3_fwd_t (16-bit storage in, same type out): the learned x2 up-sampling
// (+ skip) of the semantic | instance decoder modules in ONE launch; each half == its own launch.
template <typename T>
static int up2x_pair_impl(const T* x0, const T* x1, const float* w0, const float* w1, const float* b0,
                          const float* b1, const T* s0, const T* s1, T* y0, T* y1, int n, int h,
                          int w, int c, hipStream_t st) {
  const long total = (long)n * 4 * h * w * (c / 4);
  if ((s0 == nullptr) != (s1 == nullptr)) return EMSA_E_ARG;        // (SKIP is one template flag)
#define EMSA_UP_PAIR(I, SK)                                                                         \
  hipLaunchKernelGGL((up2x_dw_fwd_kernel<T, T, false, I, SK>), dim3(grid_for(total), 2),            \
                     dim3(kThreads), (size_t)9 * c * sizeof(float), st, x0, w0, b0, s0, y0, n, h, w, \
                     c / 4, x1, w1, b1, s1, y1)
  if (idx32_ok((long)n * 4 * h * w * c)) {
    if (s0) EMSA_UP_PAIR(uint32_t, true); else EMSA_UP_PAIR(uint32_t, false);
  } else {
    if (s0) EMSA_UP_PAIR(long, true); else EMSA_UP_PAIR(long, false);
  }
#undef EMSA_UP_PAIR
  return emsa_launch_status();
}
extern "C" int emsa_up2x_dw3x3_fwd_pair_t(int32_t dtype, const void* x0, const void* x1,
                                          const float* wdw0, const float* wdw1, const float* bias0,
                                          const float* bias1, const void* skip0, const void* skip1,
                                          void* y0, void* y1, int32_t n, int32_t h, int32_t w,
                                          int32_t c, void* stream) {
  if (!x0 || !x1 || !wdw0 || !wdw1 || !y0 || !y1 || y0 == y1) return EMSA_E_ARG;
  if ((bias0 == nullptr) != (bias1 == nullptr) || (skip0 == nullptr) != (skip1 == nullptr))
    return EMSA_E_ARG;
  if (!c4_ok(c)) return EMSA_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EMSA_DT_BF16)
    return up2x_pair_impl<emsa_bf16>((const emsa_bf16*)x0, (const emsa_bf16*)x1, wdw0, wdw1, bias0, bias1,
                                     (const emsa_bf16*)skip0, (const emsa_bf16*)skip1, (emsa_bf16*)y0,
                                     (emsa_bf16*)y1, n, h, w, c, st);
  if (dtype == EMSA_DT_F16)
    return up2x_pair_impl<emsa_f16>((const emsa_f16*)x0, (const emsa_f16*)x1, wdw0, wdw1, bias0, bias1,
                                    (const emsa_f16*)skip0, (const emsa_f16*)skip1, (emsa_f16*)y0,
                                    (emsa_f16*)y1, n, h, w, c, st);
  return EMSA_E_SHAPE;
}
extern "C" int emsa_up2x_dw3x3_fwd(const float* x, const float* wdw, const float* bias, const float* skip, float* y, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  return up2x_dw3x3_fwd_impl<float, float>(x, wdw, bias, skip, y, n, h, w, c, stream);
}
extern "C" int emsa_up2x_dw3x3_fwd_t(int32_t dtype, int32_t out_f32, const void* x, const float* wdw, const float* bias, const void* skip, void* y, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { if (out_f32) { using TO_ = float; return up2x_dw3x3_fwd_impl<float, float>((const float*)x, wdw, bias, (const float*)skip, (TO_*)y, n, h, w, c, stream); } else { using TO_ = float; return up2x_dw3x3_fwd_impl<float, float>((const float*)x, wdw, bias, (const float*)skip, (TO_*)y, n, h, w, c, stream); } }
    case EMSA_DT_BF16: { if (out_f32) { using TO_ = float; return up2x_dw3x3_fwd_impl<emsa_bf16, float>((const emsa_bf16*)x, wdw, bias, (const emsa_bf16*)skip, (TO_*)y, n, h, w, c, stream); } else { using TO_ = emsa_bf16; return up2x_dw3x3_fwd_impl<emsa_bf16, emsa_bf16>((const emsa_bf16*)x, wdw, bias, (const emsa_bf16*)skip, (TO_*)y, n, h, w, c, stream); } }
    case EMSA_DT_F16: { if (out_f32) { using TO_ = float; return up2x_dw3x3_fwd_impl<emsa_f16, float>((const emsa_f16*)x, wdw, bias, (const emsa_f16*)skip, (TO_*)y, n, h, w, c, stream); } else { using TO_ = emsa_f16; return up2x_dw3x3_fwd_impl<emsa_f16, emsa_f16>((const emsa_f16*)x, wdw, bias, (const emsa_f16*)skip, (TO_*)y, n, h, w, c, stream); } }
    default: return EMSA_E_ARG;
  }
}
template <typename T, typename TO>
static int up2x_dw3x3_bwd_data_impl(const TO* dy, const float* wdw, T* dx, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  if (!dy || !wdw || !dx) return EMSA_E_ARG;
  if (!c4_ok(c)) return EMSA_E_SHAPE;
  const long total = (long)n * h * w * (c / 4);
  hipLaunchKernelGGL((up2x_dw_bwd_data_kernel<T, TO>), dim3(grid_for(total)), dim3(kThreads),
                     (size_t)16 * c * sizeof(float), (hipStream_t)stream, dy, wdw, dx, n, h, w,
                     c / 4);
  return emsa_launch_status();
}
extern "C" int emsa_up2x_dw3x3_bwd_data(const float* dy, const float* wdw, float* dx, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  return up2x_dw3x3_bwd_data_impl<float, float>(dy, wdw, dx, n, h, w, c, stream);
}
extern "C" int emsa_up2x_dw3x3_bwd_data_t(int32_t dtype, int32_t out_f32, const void* dy, const float* wdw, void* dx, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { if (out_f32) { using TO_ = float; return up2x_dw3x3_bwd_data_impl<float, float>((const TO_*)dy, wdw, (float*)dx, n, h, w, c, stream); } else { using TO_ = float; return up2x_dw3x3_bwd_data_impl<float, float>((const TO_*)dy, wdw, (float*)dx, n, h, w, c, stream); } }
    case EMSA_DT_BF16: { if (out_f32) { using TO_ = float; return up2x_dw3x3_bwd_data_impl<emsa_bf16, float>((const TO_*)dy, wdw, (emsa_bf16*)dx, n, h, w, c, stream); } else { using TO_ = emsa_bf16; return up2x_dw3x3_bwd_data_impl<emsa_bf16, emsa_bf16>((const TO_*)dy, wdw, (emsa_bf16*)dx, n, h, w, c, stream); } }
    case EMSA_DT_F16: { if (out_f32) { using TO_ = float; return up2x_dw3x3_bwd_data_impl<emsa_f16, float>((const TO_*)dy, wdw, (emsa_f16*)dx, n, h, w, c, stream); } else { using TO_ = emsa_f16; return up2x_dw3x3_bwd_data_impl<emsa_f16, emsa_f16>((const TO_*)dy, wdw, (emsa_f16*)dx, n, h, w, c, stream); } }
    default: return EMSA_E_ARG;
  }
}
template <typename T, typename TO>
static int up2x_dw3x3_bwd_weight_impl(const TO* dy, const T* x, float* dw, float* db, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  if (!dy || !x || !dw || !db) return EMSA_E_ARG;
  if (!c4_ok(c) || c > 512) return EMSA_E_SHAPE;
  const long pixels = (long)n * h * w;
  int nblocks = (int)((pixels + 255) / 256);
  if (nblocks > 2048) nblocks = 2048;
  if (nblocks < 1) nblocks = 1;
  const int lanes = kThreads / (c / 4);
  const size_t lds = (size_t)lanes * 10 * c * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)up2x_dw_bwd_weight_kernel<T, TO>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((up2x_dw_bwd_weight_kernel<T, TO>), dim3(nblocks), dim3(kThreads), lds,
                     (hipStream_t)stream, dy, x, dw, db, n, h, w, c / 4, nblocks);
  return emsa_launch_status();
}
extern "C" int emsa_up2x_dw3x3_bwd_weight(const float* dy, const float* x, float* dw, float* db, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  return up2x_dw3x3_bwd_weight_impl<float, float>(dy, x, dw, db, n, h, w, c, stream);
}
extern "C" int emsa_up2x_dw3x3_bwd_weight_t(int32_t dtype, int32_t out_f32, const void* dy, const void* x, float* dw, float* db, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { if (out_f32) { using TO_ = float; return up2x_dw3x3_bwd_weight_impl<float, float>((const TO_*)dy, (const float*)x, dw, db, n, h, w, c, stream); } else { using TO_ = float; return up2x_dw3x3_bwd_weight_impl<float, float>((const TO_*)dy, (const float*)x, dw, db, n, h, w, c, stream); } }
    case EMSA_DT_BF16: { if (out_f32) { using TO_ = float; return up2x_dw3x3_bwd_weight_impl<emsa_bf16, float>((const TO_*)dy, (const emsa_bf16*)x, dw, db, n, h, w, c, stream); } else { using TO_ = emsa_bf16; return up2x_dw3x3_bwd_weight_impl<emsa_bf16, emsa_bf16>((const TO_*)dy, (const emsa_bf16*)x, dw, db, n, h, w, c, stream); } }
    case EMSA_DT_F16: { if (out_f32) { using TO_ = float; return up2x_dw3x3_bwd_weight_impl<emsa_f16, float>((const TO_*)dy, (const emsa_f16*)x, dw, db, n, h, w, c, stream); } else { using TO_ = emsa_f16; return up2x_dw3x3_bwd_weight_impl<emsa_f16, emsa_f16>((const TO_*)dy, (const emsa_f16*)x, dw, db, n, h, w, c, stream); } }
    default: return EMSA_E_ARG;
  }
}

// fused backward (dx optional): geometry of the LDS tiles, see up2x_dw_bwd_fused_kernel.
// Channel chunk / tile width pairs that are instantiated: whole pixels for c <= 64, 64-channel
// chunks above; TW = 256 threads / (CC/4 threads per pixel) / kUpTH rows, capped at 16.
static bool up2x_fused_geom(int c, int vmax, int* CC, int* TW) {
  if (c % vmax) return false;
  if (c == 8 || c == 16) { *CC = c; *TW = 16; return true; }
  if (c == 32) { *CC = 32; *TW = 8; return true; }
  if (c == 40) { *CC = 40; *TW = 6; return true; }
  if (c >= 64 && c % 64 == 0) { *CC = 64; *TW = 4; return true; }
  return false;
}
extern "C" int emsa_up2x_dw3x3_bwd_supported(int32_t c, int32_t esize) {
  int CC, TW;
  return up2x_fused_geom(c, esize == 4 ? 4 : 8, &CC, &TW) ? 1 : 0;
}
template <typename T, typename TO, int CC, int TW>
static int up2x_dw3x3_bwd_launch(const TO* dy, const T* x, const float* wdw, T* dx, float* dw, float* db, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  constexpr int npx = kUpTH * TW;
  constexpr int stage = (2 * kUpTH + 2) * (2 * TW + 2) * CC * (int)sizeof(TO) +
                        (kUpTH + 2) * (TW + 2) * CC * (int)sizeof(T);
  constexpr int tiles = up2x_nbuf(stage, CC) * stage + 16 * CC * (int)sizeof(float);
  constexpr int red = npx * 5 * CC * (int)sizeof(float);
  constexpr size_t lds = (size_t)(tiles > red ? tiles : red);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)up2x_dw_bwd_fused_kernel<T, TO, CC, TW, long>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)up2x_dw_bwd_fused_kernel<T, TO, CC, TW, uint32_t>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const int nchunks = (c + CC - 1) / CC;
  const long ntiles = (long)n * ((h + kUpTH - 1) / kUpTH) * ((w + TW - 1) / TW);
  // workgroups of the launch (each walks its share of the tiles of one channel chunk): 512 = the two
  // per CU that are resident at once.  Round 5 sweep (tools/up2x_bwd_bench.py, profiles/
  // r05_p_up2x_bwd_wgs.txt; 13 decoder shapes, sum of the launch times): 128 / 256 / 384 / 512 / 768 /
  // 1024 / 2048 workgroups -> 6.97 / 3.93 / 3.20 / 2.94 / 3.57 / 3.70 / 4.74 ms -- the 1024 of rounds
  // 3-4 ran a second, half-empty round whose prologue / first-tile latency / 640 atomics per
  // workgroup nothing hid (512-channel map at /32: 117 -> 56 us).  EMSA_UP2X_BWD_WGS: tuning runs
  static const int wgs = [] {
    const char* e = getenv("EMSA_UP2X_BWD_WGS");
    const int v = e ? atoi(e) : 0;
    return v >= 64 ? v : 512;
  }();
  if (ntiles >= (1L << 30)) return EMSA_E_SHAPE;     // (32-bit tile indices in the kernel)
  long bpc = wgs / nchunks;
  if (bpc > ntiles) bpc = ntiles;
  if (bpc < 1) bpc = 1;
  if (idx32_ok((long)n * 4 * h * w * c))
    hipLaunchKernelGGL((up2x_dw_bwd_fused_kernel<T, TO, CC, TW, uint32_t>),
                       dim3((unsigned)(bpc * nchunks)), dim3(kThreads), lds, (hipStream_t)stream, dy, x,
                       wdw, dx, dw, db, n, h, w, c, (int)bpc);
  else
    hipLaunchKernelGGL((up2x_dw_bwd_fused_kernel<T, TO, CC, TW, long>),
                       dim3((unsigned)(bpc * nchunks)), dim3(kThreads), lds, (hipStream_t)stream, dy, x,
                       wdw, dx, dw, db, n, h, w, c, (int)bpc);
  return emsa_launch_status();
}
template <typename T, typename TO>
static int up2x_dw3x3_bwd_impl(const TO* dy, const T* x, const float* wdw, T* dx, float* dw, float* db, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  if (!dy || !x || !wdw || !dw || !db) return EMSA_E_ARG;
  if (!c4_ok(c) || n < 1 || h < 1 || w < 1) return EMSA_E_SHAPE;
  constexpr int vmax = VecIO<T>::V > VecIO<TO>::V ? VecIO<T>::V : VecIO<TO>::V;
  int CC, TW;
  if (!up2x_fused_geom(c, vmax, &CC, &TW)) return EMSA_E_SHAPE;
  switch (CC) {
    case 8: return up2x_dw3x3_bwd_launch<T, TO, 8, 16>(dy, x, wdw, dx, dw, db, n, h, w, c, stream);
    case 16: return up2x_dw3x3_bwd_launch<T, TO, 16, 16>(dy, x, wdw, dx, dw, db, n, h, w, c, stream);
    case 32: return up2x_dw3x3_bwd_launch<T, TO, 32, 8>(dy, x, wdw, dx, dw, db, n, h, w, c, stream);
    case 40: return up2x_dw3x3_bwd_launch<T, TO, 40, 6>(dy, x, wdw, dx, dw, db, n, h, w, c, stream);
    default: return up2x_dw3x3_bwd_launch<T, TO, 64, 4>(dy, x, wdw, dx, dw, db, n, h, w, c, stream);
  }
}
extern "C" int emsa_up2x_dw3x3_bwd(const float* dy, const float* x, const float* wdw, float* dx, float* dw, float* db, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  return up2x_dw3x3_bwd_impl<float, float>(dy, x, wdw, dx, dw, db, n, h, w, c, stream);
}
extern "C" int emsa_up2x_dw3x3_bwd_t(int32_t dtype, int32_t out_f32, const void* dy, const void* x, const float* wdw, void* dx, float* dw, float* db, int32_t n, int32_t h, int32_t w, int32_t c, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: return up2x_dw3x3_bwd_impl<float, float>((const float*)dy, (const float*)x, wdw, (float*)dx, dw, db, n, h, w, c, stream);
    case EMSA_DT_BF16:
      if (out_f32) return up2x_dw3x3_bwd_impl<emsa_bf16, float>((const float*)dy, (const emsa_bf16*)x, wdw, (emsa_bf16*)dx, dw, db, n, h, w, c, stream);
      return up2x_dw3x3_bwd_impl<emsa_bf16, emsa_bf16>((const emsa_bf16*)dy, (const emsa_bf16*)x, wdw, (emsa_bf16*)dx, dw, db, n, h, w, c, stream);
    case EMSA_DT_F16:
      if (out_f32) return up2x_dw3x3_bwd_impl<emsa_f16, float>((const float*)dy, (const emsa_f16*)x, wdw, (emsa_f16*)dx, dw, db, n, h, w, c, stream);
      return up2x_dw3x3_bwd_impl<emsa_f16, emsa_f16>((const emsa_f16*)dy, (const emsa_f16*)x, wdw, (emsa_f16*)dx, dw, db, n, h, w, c, stream);
    default: return EMSA_E_ARG;
  }
}

template <typename T>
static int adaptive_avgpool_fwd_impl(const T* x, T* y, int32_t n, int32_t h, int32_t w, int32_t c, int32_t bins, void* stream) {
  if (!x || !y) return EMSA_E_ARG;
  const long total = (long)n * bins * bins * c;
  // few, large bins (the pyramid pooling's 1 x 1 and 5 x 5 grids): one workgroup per bin
  const int cvn = c / VecIO<T>::V;
  if (cv_ok<T>(c) && cvn <= 1024 && (long)h * w >= 4L * bins * bins) {
    const int threads = cvn >= 256 ? 1024 : (cvn >= 64 ? 1024 : 256);
    const int lanes = threads / cvn;
    hipLaunchKernelGGL((adaptive_avgpool_fwd_wg_kernel<T>), dim3(n * bins * bins), dim3(threads),
                       (size_t)lanes * c * sizeof(float), (hipStream_t)stream, x, y, h, w, c, bins);
    return emsa_launch_status();
  }
  hipLaunchKernelGGL((adaptive_avgpool_fwd_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0,
                     (hipStream_t)stream, x, y, n, h, w, c, bins);
  return emsa_launch_status();
}
extern "C" int emsa_adaptive_avgpool_fwd(const float* x, float* y, int32_t n, int32_t h, int32_t w, int32_t c, int32_t bins, void* stream) {
  return adaptive_avgpool_fwd_impl<float>(x, y, n, h, w, c, bins, stream);
}
extern "C" int emsa_adaptive_avgpool_fwd_t(int32_t dtype, const void* x, void* y, int32_t n, int32_t h, int32_t w, int32_t c, int32_t bins, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { return adaptive_avgpool_fwd_impl<float>((const float*)x, (float*)y, n, h, w, c, bins, stream); }
    case EMSA_DT_BF16: { return adaptive_avgpool_fwd_impl<emsa_bf16>((const emsa_bf16*)x, (emsa_bf16*)y, n, h, w, c, bins, stream); }
    case EMSA_DT_F16: { return adaptive_avgpool_fwd_impl<emsa_f16>((const emsa_f16*)x, (emsa_f16*)y, n, h, w, c, bins, stream); }
    default: return EMSA_E_ARG;
  }
}
template <typename T>
static int adaptive_avgpool_bwd_impl(const T* dy, T* dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t bins, int32_t accumulate, void* stream) {
  if (!dy || !dx) return EMSA_E_ARG;
  const long total = (long)n * h * w * c;
  hipLaunchKernelGGL((adaptive_avgpool_bwd_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0,
                     (hipStream_t)stream, dy, dx, n, h, w, c, bins, accumulate);
  return emsa_launch_status();
}
extern "C" int emsa_adaptive_avgpool_bwd(const float* dy, float* dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t bins, int32_t accumulate, void* stream) {
  return adaptive_avgpool_bwd_impl<float>(dy, dx, n, h, w, c, bins, accumulate, stream);
}
extern "C" int emsa_adaptive_avgpool_bwd_t(int32_t dtype, const void* dy, void* dx, int32_t n, int32_t h, int32_t w, int32_t c, int32_t bins, int32_t accumulate, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { return adaptive_avgpool_bwd_impl<float>((const float*)dy, (float*)dx, n, h, w, c, bins, accumulate, stream); }
    case EMSA_DT_BF16: { return adaptive_avgpool_bwd_impl<emsa_bf16>((const emsa_bf16*)dy, (emsa_bf16*)dx, n, h, w, c, bins, accumulate, stream); }
    case EMSA_DT_F16: { return adaptive_avgpool_bwd_impl<emsa_f16>((const emsa_f16*)dy, (emsa_f16*)dx, n, h, w, c, bins, accumulate, stream); }
    default: return EMSA_E_ARG;
  }
}
template <typename T>
static int bilinear_fwd_impl(const T* x, T* y, int32_t n, int32_t ih, int32_t iw, int32_t oh, int32_t ow, int32_t c, int32_t ld_y, void* stream) {
  if (!x || !y) return EMSA_E_ARG;
  const long total = (long)n * oh * ow * c;
  hipLaunchKernelGGL((bilinear_fwd_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0,
                     (hipStream_t)stream, x, y, n, ih, iw, oh, ow, c, ld_y);
  return emsa_launch_status();
}
extern "C" int emsa_bilinear_fwd(const float* x, float* y, int32_t n, int32_t ih, int32_t iw, int32_t oh, int32_t ow, int32_t c, int32_t ld_y, void* stream) {
  return bilinear_fwd_impl<float>(x, y, n, ih, iw, oh, ow, c, ld_y, stream);
}
extern "C" int emsa_bilinear_fwd_t(int32_t dtype, const void* x, void* y, int32_t n, int32_t ih, int32_t iw, int32_t oh, int32_t ow, int32_t c, int32_t ld_y, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { return bilinear_fwd_impl<float>((const float*)x, (float*)y, n, ih, iw, oh, ow, c, ld_y, stream); }
    case EMSA_DT_BF16: { return bilinear_fwd_impl<emsa_bf16>((const emsa_bf16*)x, (emsa_bf16*)y, n, ih, iw, oh, ow, c, ld_y, stream); }
    case EMSA_DT_F16: { return bilinear_fwd_impl<emsa_f16>((const emsa_f16*)x, (emsa_f16*)y, n, ih, iw, oh, ow, c, ld_y, stream); }
    default: return EMSA_E_ARG;
  }
}
template <typename T>
static int nearest_fwd_impl(const T* x, T* y, int32_t n, int32_t ih, int32_t iw, int32_t oh, int32_t ow, int32_t c, int32_t ld_y, void* stream) {
  if (!x || !y) return EMSA_E_ARG;
  if (n < 1 || ih < 1 || iw < 1 || oh < 1 || ow < 1 || c < 1 || ld_y < c) return EMSA_E_SHAPE;
  const long total = (long)n * oh * ow * c;
  hipLaunchKernelGGL((nearest_fwd_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0,
                     (hipStream_t)stream, x, y, n, ih, iw, oh, ow, c, ld_y);
  return emsa_launch_status();
}
template <typename T>
static int nearest_bwd_impl(const T* dy, float* dx, int32_t n, int32_t ih, int32_t iw, int32_t oh, int32_t ow, int32_t c, int32_t ld_dy, void* stream) {
  if (!dy || !dx) return EMSA_E_ARG;
  if (n < 1 || ih < 1 || iw < 1 || oh < 1 || ow < 1 || c < 1 || ld_dy < c) return EMSA_E_SHAPE;
  const long total = (long)n * ih * iw * c;
  hipLaunchKernelGGL((nearest_bwd_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0,
                     (hipStream_t)stream, dy, dx, n, ih, iw, oh, ow, c, ld_dy);
  return emsa_launch_status();
}
extern "C" int emsa_add_t(int32_t dtype, const void* a, const void* b, void* out, int64_t n, void* stream) {
  if (!a || !b || !out) return EMSA_E_ARG;
  if (n < 1) return EMSA_OK;
  EMSA_DISPATCH_DTYPE(dtype, T,
                      hipLaunchKernelGGL((add_kernel<T>), dim3(grid_for((long)n)), dim3(kThreads), 0,
                                         (hipStream_t)stream, (const T*)a, (const T*)b, (T*)out, (long)n));
  return emsa_launch_status();
}
extern "C" int emsa_nearest_fwd_t(int32_t dtype, const void* x, void* y, int32_t n, int32_t ih, int32_t iw, int32_t oh, int32_t ow, int32_t c, int32_t ld_y, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: return nearest_fwd_impl<float>((const float*)x, (float*)y, n, ih, iw, oh, ow, c, ld_y, stream);
    case EMSA_DT_BF16: return nearest_fwd_impl<emsa_bf16>((const emsa_bf16*)x, (emsa_bf16*)y, n, ih, iw, oh, ow, c, ld_y, stream);
    case EMSA_DT_F16: return nearest_fwd_impl<emsa_f16>((const emsa_f16*)x, (emsa_f16*)y, n, ih, iw, oh, ow, c, ld_y, stream);
    default: return EMSA_E_ARG;
  }
}
extern "C" int emsa_nearest_bwd_t(int32_t dtype, const void* dy, float* dx, int32_t n, int32_t ih, int32_t iw, int32_t oh, int32_t ow, int32_t c, int32_t ld_dy, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: return nearest_bwd_impl<float>((const float*)dy, dx, n, ih, iw, oh, ow, c, ld_dy, stream);
    case EMSA_DT_BF16: return nearest_bwd_impl<emsa_bf16>((const emsa_bf16*)dy, dx, n, ih, iw, oh, ow, c, ld_dy, stream);
    case EMSA_DT_F16: return nearest_bwd_impl<emsa_f16>((const emsa_f16*)dy, dx, n, ih, iw, oh, ow, c, ld_dy, stream);
    default: return EMSA_E_ARG;
  }
}
template <typename T>
static int bilinear_bwd_impl(const T* dy, float* dx, int32_t n, int32_t ih, int32_t iw, int32_t oh, int32_t ow, int32_t c, int32_t ld_dy, void* stream) {
  if (!dy || !dx) return EMSA_E_ARG;
  hipStream_t st = (hipStream_t)stream;
  const long total = (long)n * ih * iw * c;
  hipLaunchKernelGGL((bilinear_bwd_kernel<T>), dim3(grid_for(total)), dim3(kThreads), 0, st, dy, dx, n,
                     ih, iw, oh, ow, c, ld_dy);
  return emsa_launch_status();
}
extern "C" int emsa_bilinear_bwd(const float* dy, float* dx, int32_t n, int32_t ih, int32_t iw, int32_t oh, int32_t ow, int32_t c, int32_t ld_dy, void* stream) {
  return bilinear_bwd_impl<float>(dy, dx, n, ih, iw, oh, ow, c, ld_dy, stream);
}
extern "C" int emsa_bilinear_bwd_t(int32_t dtype, const void* dy, float* dx, int32_t n, int32_t ih, int32_t iw, int32_t oh, int32_t ow, int32_t c, int32_t ld_dy, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { return bilinear_bwd_impl<float>((const float*)dy, dx, n, ih, iw, oh, ow, c, ld_dy, stream); }
    case EMSA_DT_BF16: { return bilinear_bwd_impl<emsa_bf16>((const emsa_bf16*)dy, dx, n, ih, iw, oh, ow, c, ld_dy, stream); }
    case EMSA_DT_F16: { return bilinear_bwd_impl<emsa_f16>((const emsa_f16*)dy, dx, n, ih, iw, oh, ow, c, ld_dy, stream); }
    default: return EMSA_E_ARG;
  }
}

template <typename T, typename TO>
static int head_act_fwd_impl(const T* x, TO* y, int64_t pixels, int32_t c, int32_t n_sig, int32_t n_tanh, int32_t norm_off, int32_t n_norm, void* stream) {
  if (!x || !y) return EMSA_E_ARG;
  if ((n_norm != 0 && n_norm != 2) || n_sig < 0 || n_tanh < 0 || n_sig + n_tanh > c ||
      (n_norm && (norm_off < n_sig + n_tanh || norm_off + n_norm > c)))
    return EMSA_E_SHAPE;
  const long total = (long)pixels * c;
  if constexpr (std::is_same<TO, float>::value) {
    if (c == 8 && !((((uintptr_t)x) | ((uintptr_t)y)) & 15) && norm_off + n_norm <= 8 && norm_off >= 1) {
      hipLaunchKernelGGL((head_act_fwd8_kernel<T>), dim3(grid_for((long)pixels)), dim3(kThreads), 0,
                         (hipStream_t)stream, x, y, (long)pixels, n_sig, n_tanh, norm_off, n_norm);
      return emsa_launch_status();
    }
  }
  hipLaunchKernelGGL((head_act_fwd_kernel<T, TO>), dim3(grid_for(total)), dim3(kThreads), 0,
                     (hipStream_t)stream, x, y, total, c, n_sig, n_tanh, norm_off, n_norm);
  return emsa_launch_status();
}
extern "C" int emsa_head_act_fwd(const float* x, float* y, int64_t pixels, int32_t c, int32_t n_sig, int32_t n_tanh, int32_t norm_off, int32_t n_norm, void* stream) {
  return head_act_fwd_impl<float, float>(x, y, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream);
}
extern "C" int emsa_head_act_fwd_t(int32_t dtype, int32_t out_f32, const void* x, void* y, int64_t pixels, int32_t c, int32_t n_sig, int32_t n_tanh, int32_t norm_off, int32_t n_norm, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { if (out_f32) { using TO_ = float; return head_act_fwd_impl<float, float>((const float*)x, (TO_*)y, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream); } else { using TO_ = float; return head_act_fwd_impl<float, float>((const float*)x, (TO_*)y, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream); } }
    case EMSA_DT_BF16: { if (out_f32) { using TO_ = float; return head_act_fwd_impl<emsa_bf16, float>((const emsa_bf16*)x, (TO_*)y, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream); } else { using TO_ = emsa_bf16; return head_act_fwd_impl<emsa_bf16, emsa_bf16>((const emsa_bf16*)x, (TO_*)y, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream); } }
    case EMSA_DT_F16: { if (out_f32) { using TO_ = float; return head_act_fwd_impl<emsa_f16, float>((const emsa_f16*)x, (TO_*)y, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream); } else { using TO_ = emsa_f16; return head_act_fwd_impl<emsa_f16, emsa_f16>((const emsa_f16*)x, (TO_*)y, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream); } }
    default: return EMSA_E_ARG;
  }
}
template <typename T, typename TO>
static int head_act_bwd_impl(const TO* dy, const TO* y, const T* x, T* dx, int64_t pixels, int32_t c, int32_t n_sig, int32_t n_tanh, int32_t norm_off, int32_t n_norm, void* stream) {
  if (!dy || !y || !dx || (n_norm && !x)) return EMSA_E_ARG;
  if ((n_norm != 0 && n_norm != 2) || n_sig < 0 || n_tanh < 0 || n_sig + n_tanh > c ||
      (n_norm && (norm_off < n_sig + n_tanh || norm_off + n_norm > c)))
    return EMSA_E_SHAPE;
  const long total = (long)pixels * c;
  hipLaunchKernelGGL((head_act_bwd_kernel<T, TO>), dim3(grid_for(total)), dim3(kThreads), 0,
                     (hipStream_t)stream, dy, y, x, dx, total, c, n_sig, n_tanh, norm_off, n_norm);
  return emsa_launch_status();
}
// head_act_bwd over an 8-channel (padded) head with the task gradients gathered on the fly: g0 / g1 /
// g2 = fp32 gradients of the leading c0 / c1 / c2 channels (NHWC, pixel strides ld0..2; NULL = no
// gradient for that task), y fp32 [pixels][8], x (only for n_norm) and dx in `dtype` [pixels][8];
// channels behind c0 + c1 + c2 are padding and get a zero gradient.
extern "C" int emsa_head_act_bwd_gather_t(int32_t dtype, const float* g0, int32_t ld0, int32_t c0,
                                          const float* g1, int32_t ld1, int32_t c1, const float* g2,
                                          int32_t ld2, int32_t c2, const float* y, const void* x,
                                          void* dx, int64_t pixels, int32_t c, int32_t n_sig,
                                          int32_t n_tanh, int32_t norm_off, int32_t n_norm,
                                          void* stream) {
  if (!y || !dx || (n_norm && !x)) return EMSA_E_ARG;
  if (c != 8 || c0 < 0 || c1 < 0 || c2 < 0 || c0 + c1 + c2 > c || pixels < 1) return EMSA_E_SHAPE;
  if ((g0 && ld0 < c0) || (g1 && ld1 < c1) || (g2 && ld2 < c2)) return EMSA_E_SHAPE;
  if ((n_norm != 0 && n_norm != 2) || n_sig < 0 || n_tanh < 0 || n_sig + n_tanh > c ||
      (n_norm && (norm_off < n_sig + n_tanh || norm_off + n_norm > c)))
    return EMSA_E_SHAPE;
  if ((((uintptr_t)y) | ((uintptr_t)dx)) & 15) return EMSA_E_SHAPE;
  HeadGatherArgs a;
  a.g[0] = g0; a.g[1] = g1; a.g[2] = g2;
  a.cs[0] = c0; a.cs[1] = c1; a.cs[2] = c2;
  a.ld[0] = ld0; a.ld[1] = ld1; a.ld[2] = ld2;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(grid_for((long)pixels)), block(kThreads);
  switch (dtype) {
    case EMSA_DT_F32:
      hipLaunchKernelGGL(head_act_bwd_gather_kernel<float>, grid, block, 0, st, a, y, (const float*)x,
                         (float*)dx, (long)pixels, n_sig, n_tanh, norm_off, n_norm);
      break;
    case EMSA_DT_BF16:
      hipLaunchKernelGGL(head_act_bwd_gather_kernel<emsa_bf16>, grid, block, 0, st, a, y,
                         (const emsa_bf16*)x, (emsa_bf16*)dx, (long)pixels, n_sig, n_tanh, norm_off, n_norm);
      break;
    case EMSA_DT_F16:
      hipLaunchKernelGGL(head_act_bwd_gather_kernel<emsa_f16>, grid, block, 0, st, a, y,
                         (const emsa_f16*)x, (emsa_f16*)dx, (long)pixels, n_sig, n_tanh, norm_off, n_norm);
      break;
    default: return EMSA_E_ARG;
  }
  return emsa_launch_status();
}
extern "C" int emsa_head_act_bwd(const float* dy, const float* y, const float* x, float* dx, int64_t pixels, int32_t c, int32_t n_sig, int32_t n_tanh, int32_t norm_off, int32_t n_norm, void* stream) {
  return head_act_bwd_impl<float, float>(dy, y, x, dx, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream);
}
extern "C" int emsa_head_act_bwd_t(int32_t dtype, int32_t out_f32, const void* dy, const void* y, const void* x, void* dx, int64_t pixels, int32_t c, int32_t n_sig, int32_t n_tanh, int32_t norm_off, int32_t n_norm, void* stream) {
  switch (dtype) {
    case EMSA_DT_F32: { if (out_f32) { using TO_ = float; return head_act_bwd_impl<float, float>((const TO_*)dy, (const TO_*)y, (const float*)x, (float*)dx, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream); } else { using TO_ = float; return head_act_bwd_impl<float, float>((const TO_*)dy, (const TO_*)y, (const float*)x, (float*)dx, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream); } }
    case EMSA_DT_BF16: { if (out_f32) { using TO_ = float; return head_act_bwd_impl<emsa_bf16, float>((const TO_*)dy, (const TO_*)y, (const emsa_bf16*)x, (emsa_bf16*)dx, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream); } else { using TO_ = emsa_bf16; return head_act_bwd_impl<emsa_bf16, emsa_bf16>((const TO_*)dy, (const TO_*)y, (const emsa_bf16*)x, (emsa_bf16*)dx, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream); } }
    case EMSA_DT_F16: { if (out_f32) { using TO_ = float; return head_act_bwd_impl<emsa_f16, float>((const TO_*)dy, (const TO_*)y, (const emsa_f16*)x, (emsa_f16*)dx, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream); } else { using TO_ = emsa_f16; return head_act_bwd_impl<emsa_f16, emsa_f16>((const TO_*)dy, (const TO_*)y, (const emsa_f16*)x, (emsa_f16*)dx, pixels, c, n_sig, n_tanh, norm_off, n_norm, stream); } }
    default: return EMSA_E_ARG;
  }
}

template <typename TS, typename TD>
static int copy_channels_impl(const TS* x, int32_t ld_x, TD* y, int32_t ld_y, int64_t pixels,
                              int32_t c, void* stream) {
  if (!x || !y) return EMSA_E_ARG;
  hipLaunchKernelGGL((copy_channels_kernel<TS, TD>), dim3(grid_for((long)pixels * c)),
                     dim3(kThreads), 0, (hipStream_t)stream, x, ld_x, y, ld_y, (long)pixels, c);
  return emsa_launch_status();
}
extern "C" int emsa_copy_channels(const float* x, int32_t ld_x, float* y, int32_t ld_y,
                                  int64_t pixels, int32_t c, void* stream) {
  return copy_channels_impl<float, float>(x, ld_x, y, ld_y, pixels, c, stream);
}
// strided channel copy with conversion between storage types (fp32 <-> 16-bit at the model
// boundary: side outputs, scene logits and their cotangents)
extern "C" int emsa_cast_channels(int32_t src_dtype, const void* x, int32_t ld_x, int32_t dst_dtype,
                                  void* y, int32_t ld_y, int64_t pixels, int32_t c, void* stream) {
  EMSA_DISPATCH_DTYPE(src_dtype, TS, {
    EMSA_DISPATCH_DTYPE(dst_dtype, TD, return copy_channels_impl<TS, TD>(
                                           (const TS*)x, ld_x, (TD*)y, ld_y, pixels, c, stream));
  });
  return EMSA_E_ARG;
}
// dense NHWC copy of a logical (n, c, h, w) tensor with ELEMENT strides (s_n, s_c, s_h, s_w) in the
// storage type `dtype` (see nchw_to_nhwc_tile_kernel); strides may be 0 (expanded tensors)
extern "C" int emsa_to_nhwc_t(int32_t dtype, const void* x, void* y, int32_t n, int32_t c,
                              int32_t h, int32_t w, int64_t s_n, int64_t s_c, int64_t s_h,
                              int64_t s_w, void* stream) {
  if (!x || !y) return EMSA_E_ARG;
  if (n < 1 || c < 1 || h < 1 || w < 1 || s_n < 0 || s_c < 0 || s_h < 0 || s_w < 0)
    return EMSA_E_SHAPE;
  const long hw = (long)h * w, total = (long)n * hw * c;
  const bool planes = (s_w == 1 || w == 1) && (s_h == w || h == 1) && (s_c == hw || c == 1);
  hipStream_t st = (hipStream_t)stream;
  if (planes) {
    const int tiles_c = (c + 63) / 64;
    const long tiles_hw = (hw + 63) / 64;
    const long blocks = (long)n * tiles_c * tiles_hw;
    if (blocks >= (1L << 31)) return EMSA_E_SHAPE;
    EMSA_DISPATCH_DTYPE(dtype, T,
                        hipLaunchKernelGGL((nchw_to_nhwc_tile_kernel<T>), dim3((unsigned)blocks),
                                           dim3(256), 0, st, (const T*)x, (T*)y, c, hw, (long)s_n,
                                           tiles_c, tiles_hw));
  } else {
    EMSA_DISPATCH_DTYPE(dtype, T,
                        hipLaunchKernelGGL((strided_to_nhwc_kernel<T>), dim3(grid_for(total)),
                                           dim3(kThreads), 0, st, (const T*)x, (T*)y, c, h, w, total,
                                           (long)s_n, (long)s_c, (long)s_h, (long)s_w));
  }
  return emsa_launch_status();
}

extern "C" int emsa_axpy(const float* x, float* y, int64_t n, float alpha, void* stream) {
  if (!x || !y) return EMSA_E_ARG;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for((long)n / 4 + 1)), dim3(kThreads), 0,
                     (hipStream_t)stream, x, y, (long)n, alpha);
  return emsa_launch_status();
}

extern "C" int emsa_sgd_nesterov(float* param, const float* grad, float* momentum_buf, int64_t n,
                                 float lr, float momentum, float weight_decay, float grad_scale,
                                 int32_t first_step, void* stream) {
  if (!param || !grad || !momentum_buf) return EMSA_E_ARG;
  if (n < 1) return EMSA_OK;
  if ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)momentum_buf)) & 15) return EMSA_E_SHAPE;
  const long n4 = (long)n / 4;
  hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(kThreads), 0,
                     (hipStream_t)stream, param, grad, momentum_buf, n4, (long)n, lr, momentum,
                     weight_decay, grad_scale, first_step ? 1 : 0, (const float*)nullptr);
  return emsa_launch_status();
}
extern "C" int emsa_sgd_nesterov_dev(float* param, const float* grad, float* momentum_buf,
                                     int64_t n, const float* hyper, void* stream) {
  if (!param || !grad || !momentum_buf || !hyper) return EMSA_E_ARG;
  if (n < 1) return EMSA_OK;
  if ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)momentum_buf)) & 15) return EMSA_E_SHAPE;
  const long n4 = (long)n / 4;
  hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(kThreads), 0,
                     (hipStream_t)stream, param, grad, momentum_buf, n4, (long)n, 0.f, 0.f, 0.f,
                     0.f, 0, hyper);
  return emsa_launch_status();
}
extern "C" int emsa_adam_advance(const double* hyper, double* state, void* stream) {
  if (!hyper || !state) return EMSA_E_ARG;
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper, state);
  return emsa_launch_status();
}
extern "C" int emsa_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                              int64_t n, const double* hyper, const double* state, void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || !hyper || !state) return EMSA_E_ARG;
  if (n < 1) return EMSA_OK;
  if ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq)) & 15)
    return EMSA_E_SHAPE;
  const long n4 = (long)n / 4;
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(kThreads), 0,
                     (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n4, (long)n, hyper, state);
  return emsa_launch_status();
}
