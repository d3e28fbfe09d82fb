// Implicit-GEMM convolution on CDNA4 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
//   emsa_conv_igemm : out[m][n] = epi( sum_{tap,c} in[gather(m,tap)][c] * w[tap][n][c] )
//                     forward conv and data gradient (same kernel, different gather geometry)
//   emsa_conv_wgrad : dw[tap][n][c] += sum_m dout[m][n] * in[gather(m,tap)][c]
//
// Stands in for every nn.Conv2d / nn.Linear the reference model composes
// (/root/reference/emsanet/model.py:47-119, /root/reference/emsanet/decoder.py:61-199), which
// the reference runs as cuDNN eager kernels (/root/reference/main.py:23-24).
//
// Design (MI355X_MICROARCH.md / cdna_hip_programming.md):
//   * NHWC activations: one GEMM row = one pixel = C contiguous floats -> every global access is
//     a full 128-B line (8 lanes x float4), no im2col buffer in HBM.
//   * K order inside a BK=32 chunk is permuted so that both MFMA operands are fetched from LDS
//     with ds_read_b128: lane (i = l&31, h = l>>5) reads 4 consecutive k at 8t+4h and feeds
//     them to 4 consecutive MFMAs; A and B use the same permutation so the product is exact.
//   * LDS rows padded to 36 floats: the 16-lane groups of ds_read_b128 then cover all 64 banks
//     (9*i mod 16 is a bijection) -> conflict-free reads; writes are 8 contiguous lanes per row.
//   * register-staged double buffering, one barrier per K step; 2 workgroups per CU so one
//     group's MFMA phase overlaps the other's global/LDS traffic.
//   * epilogue fuses bias, BatchNorm batch-statistics partials (deterministic, no atomics),
//     folded eval BatchNorm, residual add, ReLU and the ReLU-backward mask.
#include "common.h"

#include <type_traits>
#include <vector>

#include "prof.h"

namespace {

// ---- optional per-launch timing (bench.py roofline): HIP events on the launch stream ---------
// (prof.h; shared with conv_wino.hip)
struct ProfSlotImpl {
  hipEvent_t a, b;
  int cls;
  double flops, bytes;
};
std::vector<ProfSlotImpl> g_prof_pool;
size_t g_prof_used = 0;
int g_prof_every = 0;          // 0 = off, n = bracket every n-th launch of each kernel class
int g_prof_seen[kProfClasses] = {0};
double g_prof_next_flops = 0.0;   // one-shot override of the next sampled launch's FLOP count
const char* const kProfNames[kProfClasses] = {
    "conv_igemm_kernel<128,128,2,2>", "conv_igemm_kernel<128,64,4,2>",
    "conv_igemm_kernel<64,64,2,2>",   "conv_igemm_kernel<128,32,4,1>",
    "conv_wgrad_kernel<64,32,7,2,1,2>", "conv_wgrad_kernel<128,128,1,2,2,1> (unused)",
    "conv_wgrad_kernel<64,64,1,2,2,1>", "conv_wgrad_kernel<64,64,3,2,2,1>|conv_wgrad1d_wino_kernel<64,64> (1-D convs)",
    "conv1d_wino_kernel (1-D 3x1/1x3 convs)", "conv_h_kernel (16-bit igemm)",
    "wgrad kernels on 16-bit activations", "conv1d_wino_kernel (dense 3x3 convs, row-Winograd)",
    "conv_wgrad1d_wino_kernel<64,64> (dense 3x3 convs)",
    "conv_rs_kernel (16-bit 1-D 3x1/1x3 convs, register-stationary)"};
}  // namespace

int emsa_prof_begin(int cls, double flops, hipStream_t st, double bytes) {
  if (g_prof_every <= 0) return -1;
  struct ClearOverride {            // the override applies to exactly one launch, sampled or not
    ~ClearOverride() { g_prof_next_flops = 0.0; }
  } clear_override;
  if ((g_prof_seen[cls]++ % g_prof_every) != 0) return -1;
  if (g_prof_used == g_prof_pool.size()) {
    ProfSlotImpl s;
    if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return -1;
    g_prof_pool.push_back(s);
  }
  const int id = (int)g_prof_used++;
  ProfSlotImpl* s = &g_prof_pool[id];
  s->cls = cls;
  s->flops = g_prof_next_flops > 0.0 ? g_prof_next_flops : flops;
  s->bytes = bytes;
  (void)hipEventRecord(s->a, st);
  return id;
}
void emsa_prof_end(int slot, hipStream_t st) {
  if (slot >= 0) (void)hipEventRecord(g_prof_pool[slot].b, st);
}

namespace {
using ProfSlot = int;
inline int prof_begin(int cls, double flops, hipStream_t st, double bytes = 0.0) {
  return emsa_prof_begin(cls, flops, st, bytes);
}
inline void prof_end(int s, hipStream_t st) { emsa_prof_end(s, st); }

// algorithmic (direct-convolution) FLOPs of one launch: 2 * pixels * k_ch * n_ch * taps, pixels
// counted on the grid of the FORWARD conv's output (for a strided data gradient that is the
// gathered grid), no discount for padding taps
double algo_flops(const EmsaConvGeom& g) {
  const bool transposed = g.div_h > 1 || g.div_w > 1;
  const double px = (double)g.n_img * (transposed ? (double)g.in_h * g.in_w
                                                  : (double)g.out_h * g.out_w);
  return 2.0 * px * g.k_ch * g.n_ch * g.kh * g.kw;
}

#ifndef EMSA_ABL
#define EMSA_ABL 0   // tuning only (tools/conv_bench.py): 1 = no global loads, 2 = no stores, 4 = no MFMA,
                     // 8 = one K step per workgroup (fixed cost of a weight-gradient launch)
#endif
// Single LDS buffer + register prefetch (two barriers per step) beats double buffering on
// MI355X for both kernels (profiles/r01_c_*): half the LDS -> twice the resident workgroups, and
// with the 64-cycle fp32 MFMA it is resident waves, not barrier count, that keeps the matrix
// pipe fed.  EMSA_SB: bit0 = igemm single buffer, bit1 = wgrad single buffer.
#ifndef EMSA_SB
#define EMSA_SB 3
#endif
#ifndef EMSA_BK
#define EMSA_BK 32
#endif
#ifndef EMSA_SCHED
#define EMSA_SCHED 1   // tuning: bit0 = s_setprio(1) around the MFMA cluster, bit1 = issue the
#endif                 // next-step global loads after the first MFMA group
constexpr int kBK = EMSA_BK;      // K chunk (channels) per step
constexpr int kLD = kBK + 4;      // padded LDS row (floats): conflict-free ds_read_b128
constexpr int kRowLanes = kBK / 4;          // lanes (float4) per staged row


// unsigned division by a launch-time constant (n < 2^31): q = (mulhi(n, mul) + n) >> shift
struct FastDiv {
  uint32_t mul, shift, d;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  f.shift = s;
  f.mul = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv& f) {
  return (__umulhi(n, f.mul) + n) >> f.shift;
}

struct ConvArgs {
  EmsaConvGeom g;
  const float* in;
  const float* w;
  float* out;
  const float* bias;
  float* stats;
  const float* scale;
  const float* shift;
  const float* residual;
  const float* mask_src;
  int ld_res, ld_mask, act;
  int M, tiles_m, tiles_n, kchunks;
  uint32_t in_bytes, w_bytes;
  int vec_epilogue;          // float4 epilogue legal (n_ch, ld's, pointers 16-B aligned)
  int mapped;                // output pixel map in use (EmsaConvGeom::out_pix_*)
  FastDiv div_ohw, div_ow;
};

// Hardware bounds-checked 16-byte load: an offset >= num_records returns 0, which IS the zero
// padding / tail predication of the implicit GEMM -- no exec-mask branches around the loads.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kOOB = 0x80000000u;   // every buffer is < 2 GiB (checked by the launcher)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
// per-thread offset (VGPR) + wave-uniform offset in the instruction's scalar-offset operand: the
// K-loop advance costs no vector instruction; kOOB in `voff` still reads zero
__device__ __forceinline__ float4 buf_ld4s(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float4,
                            __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

// Four consecutive channels of an activation in storage type T (16 bytes for float, 8 for the
// 16-bit types: the weight-gradient kernels below read the bf16 activations of the mixed-precision
// training path through the same loaders).  The load returns the RAW registers; the conversion to
// fp32 happens where the prefetched registers are consumed (store_lds) -- converting right behind
// the load would put an s_waitcnt on the fresh data in front of the step's MFMAs.
typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));
template <typename T> struct Raw4 { typedef u32x2w type; };
template <> struct Raw4<float> { typedef float4 type; };
template <typename T>
__device__ __forceinline__ typename Raw4<T>::type buf_ld4raw(__amdgpu_buffer_rsrc_t r, uint32_t byte_off) {
  if constexpr (sizeof(T) == 4) {
    return buf_ld4(r, byte_off);
  } else {
    return __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
  }
}
__device__ __forceinline__ float4 raw_f4(const float4& v, float*) { return v; }
__device__ __forceinline__ float4 raw_f4(const u32x2w& raw, emsa_bf16*) {
  return make_float4(__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xFFFF0000u),
                     __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xFFFF0000u));
}
__device__ __forceinline__ float4 raw_f4(const u32x2w& raw, emsa_f16*) {
  const emsa_f16x4 h = __builtin_bit_cast(emsa_f16x4, raw);
  return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
}

// gather helper: BYTE offset of (row base, tap) or kOOB
struct Gather {
  int mul_h, off_h, step_h, div_h, mul_w, off_w, step_w, div_w, in_h, in_w;
  int row_stride, px_stride;
};

__device__ __forceinline__ uint32_t gather_offset(const Gather& q, int img_off, int bh, int bw,
                                                  int kh, int kw) {
  int hn = bh + kh * q.step_h, wn = bw + kw * q.step_w;
  bool ok = true;
  if (q.div_h > 1) {      // strided data gradient only (wave-uniform branch)
    ok = ok && (hn % q.div_h == 0);
    hn /= q.div_h;
  }
  if (q.div_w > 1) {
    ok = ok && (wn % q.div_w == 0);
    wn /= q.div_w;
  }
  ok = ok && hn >= 0 && hn < q.in_h && wn >= 0 && wn < q.in_w;
  return ok ? (uint32_t)(img_off + hn * q.row_stride + wn * q.px_stride) * 4u : kOOB;
}

#ifndef EMSA_WPE
#define EMSA_WPE 6     // min waves per SIMD requested for the 64x64 tile (register cap)
#endif
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64, (BM * BN <= 64 * 64) ? EMSA_WPE : 2) void
conv_igemm_kernel(const ConvArgs p) {
  constexpr int NT = WM * WN * 64;                 // threads per workgroup (4 or 8 waves)
  constexpr int kRowsPerPass = NT / kRowLanes;
  static_assert(BM % kRowsPerPass == 0 && BN % kRowsPerPass == 0, "loader mapping");
  constexpr int AR = BM / kRowsPerPass, BR = BN / kRowsPerPass;   // float4 per thread per tile
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NBUF = (EMSA_SB & 1) ? 1 : 2;
  float* const As = smem;                     // [NBUF][BM][kLD]
  float* const Bs = smem + NBUF * BM * kLD;   // [NBUF][BN][kLD]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int wg = emsa_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = wg % p.tiles_n, mt = wg / p.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  const EmsaConvGeom& g = p.g;
  Gather q;
  q.mul_h = g.mul_h; q.off_h = g.off_h; q.step_h = g.step_h; q.div_h = g.div_h;
  q.mul_w = g.mul_w; q.off_w = g.off_w; q.step_w = g.step_w; q.div_w = g.div_w;
  q.in_h = g.in_h; q.in_w = g.in_w;
  q.row_stride = (int)g.in_row_stride; q.px_stride = g.in_px_stride;

  // ---- loader state -------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes);
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, p.w_bytes);
  const int rl = tid / kRowLanes, c4 = (tid % kRowLanes) * 4;
  int a_bh[AR], a_bw[AR], a_img[AR];
  uint32_t a_off[AR], b_off[BR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    const int m = m0 + rl + kRowsPerPass * j;
    if (m < p.M) {
      const int img = (int)fast_div((uint32_t)m, p.div_ohw);
      const int rem = m - img * (int)p.div_ohw.d;
      const int oh = (int)fast_div((uint32_t)rem, p.div_ow), ow = rem - oh * g.out_w;
      a_bh[j] = oh * q.mul_h + q.off_h;
      a_bw[j] = ow * q.mul_w + q.off_w;
      a_img[j] = img * (int)g.in_img_stride;
    } else {
      a_bh[j] = -(1 << 29);
      a_bw[j] = 0;
      a_img[j] = 0;
    }
  }
#pragma unroll
  for (int j = 0; j < BR; ++j) {
    const int n = n0 + rl + kRowsPerPass * j;
    b_off[j] = n < g.n_ch ? (uint32_t)(n * g.k_ch + c4) * 4u : kOOB;
  }
  const int taps = g.kh * g.kw;
  const int steps = taps * p.kchunks;
  const uint32_t w_tap_bytes = (uint32_t)g.n_ch * g.k_ch * 4u;
  int tap_n = 0, kc_n = 0;
  float4 ra[AR], rb[BR];
  // kOOB for the lanes whose channels lie beyond k_ch in the last chunk of a tap
  const uint32_t last_oob = (p.kchunks - 1) * kBK + c4 < g.k_ch ? 0u : kOOB;

  auto load_regs = [&]() {
    // (tap, channel chunk) are wave-uniform: the gathered pixel offsets change once per tap, the
    // channel advance travels in the scalar offset operand
    if (kc_n == 0) {
      const int kh = tap_n / g.kw, kw = tap_n - kh * g.kw;
#pragma unroll
      for (int j = 0; j < AR; ++j) {
        const uint32_t o = gather_offset(q, a_img[j], a_bh[j], a_bw[j], kh, kw);
        a_off[j] = (o & kOOB) ? kOOB : o + (uint32_t)c4 * 4u;
      }
    }
    const int k0 = kc_n * kBK;
    const uint32_t sa = (uint32_t)k0 * 4u, sb = (uint32_t)tap_n * w_tap_bytes + (uint32_t)k0 * 4u;
#if EMSA_ABL & 1
#pragma unroll
    for (int j = 0; j < AR; ++j) ra[j] = make_float4(a_off[j], sa, 1.f, 2.f);
#pragma unroll
    for (int j = 0; j < BR; ++j) rb[j] = make_float4(b_off[j], sb, 2.f, 1.f);
#else
    // partial last channel chunk: lanes beyond k_ch go out of range through a wave-uniform mask
    // (no branch around the loads: a join behind them costs an s_waitcnt on the fresh data)
    const uint32_t pm = k0 + kBK > g.k_ch ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int j = 0; j < AR; ++j) ra[j] = buf_ld4s(rs_in, a_off[j] | (last_oob & pm), sa);
#pragma unroll
    for (int j = 0; j < BR; ++j) rb[j] = buf_ld4s(rs_w, b_off[j] | (last_oob & pm), sb);
#endif
    if (++kc_n == p.kchunks) {
      kc_n = 0;
      ++tap_n;
    }
  };
  auto store_lds = [&](int buf) {
    float* a = As + buf * BM * kLD;
    float* b = Bs + buf * BN * kLD;
#pragma unroll
    for (int j = 0; j < AR; ++j) emsa_st4(a + (rl + kRowsPerPass * j) * kLD + c4, ra[j]);
#pragma unroll
    for (int j = 0; j < BR; ++j) emsa_st4(b + (rl + kRowsPerPass * j) * kLD + c4, rb[j]);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_regs();
  store_lds(0);
  __syncthreads();

  int cur = 0;
  for (int s = 0; s < steps; ++s) {
    const bool has_next = s + 1 < steps;
    if (!(EMSA_SCHED & 2) && has_next) load_regs();

    const float* a = As + cur * BM * kLD + (wm * TM * 32 + l31) * kLD + lh * 4;
    const float* b = Bs + cur * BN * kLD + (wn * TN * 32 + l31) * kLD + lh * 4;
#pragma unroll
    for (int t = 0; t < kBK / 8; ++t) {
      float4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = emsa_ld4(a + i * 32 * kLD + t * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = emsa_ld4(b + j * 32 * kLD + t * 8);
      if (EMSA_SCHED & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float av = kk == 0 ? fa[i].x : kk == 1 ? fa[i].y : kk == 2 ? fa[i].z : fa[i].w;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const float bv = kk == 0 ? fb[j].x : kk == 1 ? fb[j].y : kk == 2 ? fb[j].z : fb[j].w;
#if EMSA_ABL & 4
            acc[i][j][kk] += av * bv;
#else
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
#endif
          }
        }
      }
      if (EMSA_SCHED & 1) __builtin_amdgcn_s_setprio(0);
      if ((EMSA_SCHED & 2) && t == 0 && has_next) load_regs();
    }
    if (NBUF == 2) {
      if (has_next) store_lds(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    } else {
      __syncthreads();                 // every wave is done reading the buffer
      if (has_next) store_lds(0);
      __syncthreads();
    }
  }

  // ---- epilogue -------------------------------------------------------------------------
  const bool want_stats = p.stats != nullptr;
  float s1[TN], s2[TN], bvv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + l31;
    bvv[j] = (n < g.n_ch && p.bias) ? p.bias[n] : 0.f;
    s1[j] = s2[j] = 0.f;
  }
  if (want_stats) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m0 + row < p.M) s1[j] += acc[i][j][r] + bvv[j];
        }
  }

  if (want_stats) {
    // Per-tile (count, sum, M2 about the tile mean) -> merged with Chan's formula in
    // emsa_bn_finalize: no E[x^2]-mean^2 cancellation.  Deterministic (no atomics).
    float* red = smem;            // [WM][BN]; main-loop LDS reads are all behind a barrier
    float* tmean = smem + WM * BN;  // [BN]
    const int cnt = min(BM, p.M - m0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      s1[j] += __shfl_xor(s1[j], 32);
      if (lh == 0) red[wm * BN + (wn * TN + j) * 32 + l31] = s1[j];
    }
    __syncthreads();
    for (int col = tid; col < BN; col += NT) {
      float a1 = 0.f;
#pragma unroll
      for (int w_ = 0; w_ < WM; ++w_) a1 += red[w_ * BN + col];
      tmean[col] = a1 / (float)cnt;
      const int n = n0 + col;
      if (n < g.n_ch) {
        p.stats[((size_t)0 * p.tiles_m + mt) * g.n_ch + n] = a1;
        p.stats[((size_t)2 * p.tiles_m + mt) * g.n_ch + n] = (float)cnt;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = (wn * TN + j) * 32 + l31;
      const int n = n0 + col;
      const float bv = (n < g.n_ch && p.bias) ? p.bias[n] : 0.f;
      const float mu = tmean[col];
      float q2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m0 + row < p.M) {
            const float d = acc[i][j][r] + bv - mu;
            q2 += d * d;
          }
        }
      s2[j] = q2 + __shfl_xor(q2, 32);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j)
      if (lh == 0) red[wm * BN + (wn * TN + j) * 32 + l31] = s2[j];
    __syncthreads();
    for (int col = tid; col < BN; col += NT) {
      const int n = n0 + col;
      if (n < g.n_ch) {
        float a2 = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < WM; ++w_) a2 += red[w_ * BN + col];
        p.stats[((size_t)1 * p.tiles_m + mt) * g.n_ch + n] = a2;
      }
    }
  }

  // ---- output pass --------------------------------------------------------------------------
  if (p.vec_epilogue) {
    // stage (acc + bias) through LDS so that global stores, residual and mask accesses are
    // 16 B per lane along channels (whole 128-B lines per 8 lanes) instead of 4-B scalars
    constexpr int SLD = BN + 4;
    float* stage = smem;                 // [BM][SLD] <= LDS of the main loop
    __syncthreads();                     // statistics (if any) are done with LDS
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          stage[row * SLD + (wn * TN + j) * 32 + l31] = acc[i][j][r] + bvv[j];
        }
    __syncthreads();
    constexpr int C4 = BN / 4, RPP = NT / C4;
    const int col4 = tid % C4, row0 = tid / C4;
    const int n = n0 + col4 * 4;
    if (n < g.n_ch) {
      float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = emsa_zero4();
      if (p.scale) {
        sc = emsa_ld4(p.scale + n);
        sh = emsa_ld4(p.shift + n);
      }
#pragma unroll 4
      for (int row = row0; row < BM; row += RPP) {
        const int mrow = m0 + row;
        if (mrow >= p.M) break;
        // pixel of the produced tensor: row index itself, or through the output map (one phase
        // of a strided data gradient writes every s-th row / column)
        int m = mrow;
        if (p.mapped) {
          const int img = (int)fast_div((uint32_t)mrow, p.div_ohw);
          const int rem = mrow - img * (int)p.div_ohw.d;
          const int oh = (int)fast_div((uint32_t)rem, p.div_ow), ow = rem - oh * g.out_w;
          m = img * g.out_pix_img + oh * g.out_pix_row + ow * g.out_pix_px + g.out_pix_off;
        }
        float4 v = emsa_ld4(stage + row * SLD + col4 * 4);
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y;
        v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        if (p.residual) {
          const float4 rr = emsa_ld4(p.residual + (size_t)m * p.ld_res + n);
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (p.mask_src) {
          const float4 mm = emsa_ld4(p.mask_src + (size_t)m * p.ld_mask + n);
          v.x = mm.x > 0.f ? v.x : 0.f; v.y = mm.y > 0.f ? v.y : 0.f;
          v.z = mm.z > 0.f ? v.z : 0.f; v.w = mm.w > 0.f ? v.w : 0.f;
        }
        if (p.act == EMSA_ACT_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
          v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
#if EMSA_ABL & 2
        if (v.x == 1.2345e30f) emsa_st4(p.out + (size_t)m * g.ld_out + n, v);
#else
        emsa_st4(p.out + (size_t)m * g.ld_out + n, v);
#endif
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + l31;
      const bool nok = n < g.n_ch;
      const float sc = (nok && p.scale) ? p.scale[n] : 1.f;
      const float sh = (nok && p.shift) ? p.shift[n] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int mrow = m0 + row;
          if (mrow < p.M && nok) {
            int m = mrow;
            if (p.mapped) {
              const int img = (int)fast_div((uint32_t)mrow, p.div_ohw);
              const int rem = mrow - img * (int)p.div_ohw.d;
              const int oh = (int)fast_div((uint32_t)rem, p.div_ow), ow = rem - oh * g.out_w;
              m = img * g.out_pix_img + oh * g.out_pix_row + ow * g.out_pix_px + g.out_pix_off;
            }
            float v = (acc[i][j][r] + bvv[j]) * sc + sh;
            if (p.residual) v += p.residual[(size_t)m * p.ld_res + n];
            if (p.mask_src) v = p.mask_src[(size_t)m * p.ld_mask + n] > 0.f ? v : 0.f;
            if (p.act == EMSA_ACT_RELU) v = fmaxf(v, 0.f);
            p.out[(size_t)m * g.ld_out + n] = v;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------
struct WgradArgs {
  EmsaConvGeom g;
  const float* in;
  const float* dout;
  float* dw;
  float* dbias;
  int M, steps_total, steps_per_split;
  int n_co_tiles, n_ci_tiles, n_tap_groups, n_tiles;
  int dout_aligned;   // float4 loads of dout are legal
  uint32_t in_bytes, dout_bytes;
  FastDiv div_ohw, div_ow;
};

// ALIGNED: float4 loads of dy are legal.  A compile-time split, not a branch around the loads: a
// control-flow join behind a load makes the compiler wait for the data at the join, i.e. before
// the step's MFMAs.
template <int BCO, int BCI, int TT, int WCO, int WCI, int WT, bool ALIGNED, typename T = float>
__device__ __forceinline__ void conv_wgrad_body(const WgradArgs& p) {
  constexpr uint32_t ES = sizeof(T);            // bytes per activation element
  static_assert(WCO * WCI * WT == 4, "4 waves");
  constexpr int PK = 32;                       // pixels (GEMM K) per step
  constexpr int TCO = BCO / 32 / WCO, TCI = BCI / 32 / WCI, TTW = (TT + WT - 1) / WT;
  constexpr int DR = BCO / 32, XR = BCI / 32;   // float4 per thread per tile
  constexpr int DTPR = BCO / 4, XTPR = BCI / 4; // threads per pixel row
  constexpr int BUF = PK * BCO + TT * PK * BCI; // floats per LDS buffer
  constexpr int NBUF = (EMSA_SB & 2) ? 1 : 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wco = wave % WCO, wci = (wave / WCO) % WCI, wt = wave / (WCO * WCI);

  const int tile = blockIdx.x % p.n_tiles, ks = blockIdx.x / p.n_tiles;
  const int tg = tile % p.n_tap_groups;
  const int ci_t = (tile / p.n_tap_groups) % p.n_ci_tiles;
  const int co_t = tile / (p.n_tap_groups * p.n_ci_tiles);
  const int co0 = co_t * BCO, ci0 = ci_t * BCI, tap0 = tg * TT;

  const EmsaConvGeom& g = p.g;
  const int taps = g.kh * g.kw;
  Gather q;
  q.mul_h = g.mul_h; q.off_h = g.off_h; q.step_h = g.step_h; q.div_h = g.div_h;
  q.mul_w = g.mul_w; q.off_w = g.off_w; q.step_w = g.step_w; q.div_w = g.div_w;
  q.in_h = g.in_h; q.in_w = g.in_w;
  q.row_stride = (int)g.in_row_stride; q.px_stride = g.in_px_stride;

  const int s_begin = ks * p.steps_per_split;
  const int s_end = min(s_begin + p.steps_per_split, p.steps_total);

  const int d_c4 = (tid % DTPR) * 4, d_r = tid / DTPR;
  const int x_c4 = (tid % XTPR) * 4, x_r = tid / XTPR;
  const bool do_bias = p.dbias != nullptr && ci_t == 0 && tg == 0;

  typename Raw4<T>::type rd[DR], rx[TT][XR];
  float4 bsum = emsa_zero4();

  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes);
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(p.dout, p.dout_bytes);
  // (kh, kw) of this workgroup's taps: uniform, computed once
  int tkh[TT], tkw[TT];
  bool tok[TT];
#pragma unroll
  for (int t = 0; t < TT; ++t) {
    const int tap = tap0 + t;
    tok[t] = tap < taps;
    tkh[t] = tap / g.kw;
    tkw[t] = tap - tkh[t] * g.kw;
  }
  // channel part of the x addresses; a column beyond the tensor's channels stays out of range
  const bool xok = ci0 + x_c4 < g.k_ch;
  const uint32_t xadd = xok ? (uint32_t)(ci0 + x_c4) * ES : 0u, xmask = xok ? 0u : kOOB;
  constexpr int TH = (TT + 1) / 2;              // taps per lane half
  auto load_regs = [&](int s) {
    const int mb = s * PK;
#pragma unroll
    for (int j = 0; j < DR; ++j) {
      const int m = mb + d_r + j * (256 / DTPR);
      const int co = co0 + d_c4;
      if constexpr (ALIGNED) {
        rd[j] = buf_ld4raw<T>(rs_dy, (m < p.M && co < g.n_ch) ? (uint32_t)(m * g.ld_out + co) * ES : kOOB);
      } else if constexpr (sizeof(T) == 4) {
        float4 v = emsa_zero4();
        if (m < p.M && co < g.n_ch) {
          const float* src = p.dout + (size_t)m * g.ld_out + co;
          v.x = src[0];
          if (co + 1 < g.n_ch) v.y = src[1];
          if (co + 2 < g.n_ch) v.z = src[2];
          if (co + 3 < g.n_ch) v.w = src[3];
        }
        rd[j] = v;
      }
    }
    // x: the 16+ lanes that load one pixel row share its gathered address, so lane (r, half) of
    // every wave computes the byte offsets of pixel mb + r for the taps t with (t & 1) == half
    // ONCE and the loading threads fetch them with wave shuffles (per-thread decomposition and
    // gather were ~900 VALU instructions per K step -- more issue time than the step's MFMAs)
    const int m = mb + l31;
    int bh = -(1 << 29), bw = 0, img_off = 0;
    if (m < p.M) {
      const int img = (int)fast_div((uint32_t)m, p.div_ohw);
      const int rem = m - img * (int)p.div_ohw.d;
      const int oh = (int)fast_div((uint32_t)rem, p.div_ow), ow = rem - oh * g.out_w;
      bh = oh * q.mul_h + q.off_h;
      bw = ow * q.mul_w + q.off_w;
      img_off = img * (int)g.in_img_stride;
    }
    uint32_t goff[TH];
#pragma unroll
    for (int u = 0; u < TH; ++u) {
      const int t0 = 2 * u, t1 = 2 * u + 1 < TT ? 2 * u + 1 : 2 * u;
      const bool two = 2 * u + 1 < TT;
      const int kh = lh ? tkh[t1] : tkh[t0], kw = lh ? tkw[t1] : tkw[t0];
      const bool ok_t = lh ? (two && tok[t1]) : tok[t0];
      const int hn = bh + kh * q.step_h, wn = bw + kw * q.step_w;    // (div_h = div_w = 1 here)
      const bool ok = ok_t && hn >= 0 && hn < q.in_h && wn >= 0 && wn < q.in_w;
      goff[u] = ok ? (uint32_t)(img_off + hn * q.row_stride + wn * q.px_stride) * ES : kOOB;
    }
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int r = x_r + j * (256 / XTPR);
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const uint32_t o =
            (uint32_t)__builtin_amdgcn_ds_bpermute((r + 32 * (t & 1)) * 4, (int)goff[t >> 1]);
        rx[t][j] = buf_ld4raw<T>(rs_in, (o + xadd) | xmask);
      }
    }
  };
  auto store_lds = [&](int buf) {
    float* d = smem + buf * BUF;
    float* x = d + PK * BCO;
#pragma unroll
    for (int j = 0; j < DR; ++j) {
      // (the bias gradient = column sums of dy is accumulated HERE, where the prefetched
      //  registers are consumed anyway: summing right behind the load puts the wave to sleep
      //  on s_waitcnt until the next step's data has arrived, before this step's MFMAs)
      const float4 dv = raw_f4(rd[j], (T*)nullptr);
      bsum.x += dv.x; bsum.y += dv.y; bsum.z += dv.z; bsum.w += dv.w;
      emsa_st4(d + (d_r + j * (256 / DTPR)) * BCO + d_c4, dv);
    }
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
      for (int j = 0; j < XR; ++j)
        emsa_st4(x + t * PK * BCI + (x_r + j * (256 / XTPR)) * BCI + x_c4,
                 raw_f4(rx[t][j], (T*)nullptr));
  };

  f32x16 acc[TCO][TCI][TTW];
#pragma unroll
  for (int i = 0; i < TCO; ++i)
#pragma unroll
    for (int j = 0; j < TCI; ++j)
#pragma unroll
      for (int t = 0; t < TTW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][t][r] = 0.f;

  if (s_begin < s_end) {
    load_regs(s_begin);
    store_lds(0);
  }
  __syncthreads();

  int cur = 0;
  for (int s = s_begin; s < s_end; ++s) {
    const bool has_next = s + 1 < s_end;
    if (has_next) load_regs(s + 1);

    const float* d = smem + cur * BUF + lh * BCO + wco * TCO * 32 + l31;
    const float* x = smem + cur * BUF + PK * BCO + lh * BCI + wci * TCI * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < PK / 2; ++kk) {
      float fa[TCO];
#pragma unroll
      for (int i = 0; i < TCO; ++i) fa[i] = d[kk * 2 * BCO + i * 32];
#pragma unroll
      for (int t = 0; t < TTW; ++t) {
        const int tl = wt * TTW + t;
        if (tl < TT) {
#pragma unroll
          for (int j = 0; j < TCI; ++j) {
            const float fb = x[tl * PK * BCI + kk * 2 * BCI + j * 32];
#pragma unroll
            for (int i = 0; i < TCO; ++i)
              acc[i][j][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb, acc[i][j][t], 0, 0, 0);
          }
        }
      }
    }
    if (NBUF == 2) {
      if (has_next) store_lds(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    } else {
      __syncthreads();                 // every wave is done reading the buffer
      if (has_next) store_lds(0);
      __syncthreads();
    }
  }

  // ---- epilogue: split-K partial -> atomic add ------------------------------------------
#pragma unroll
  for (int t = 0; t < TTW; ++t) {
    const int tl = wt * TTW + t;
    const int tap = tap0 + tl;
    if (tl < TT && tap < taps) {
#pragma unroll
      for (int i = 0; i < TCO; ++i)
#pragma unroll
        for (int j = 0; j < TCI; ++j) {
          const int ci = ci0 + (wci * TCI + j) * 32 + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int co = co0 + (wco * TCO + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
#if EMSA_ABL & 2
            if (co < g.n_ch && ci < g.k_ch && acc[i][j][t][r] == 1.2345e30f)
#else
            if (co < g.n_ch && ci < g.k_ch)
#endif
              unsafeAtomicAdd(p.dw + ((size_t)tap * g.n_ch + co) * g.k_ch + ci, acc[i][j][t][r]);
          }
        }
    }
  }

  if (do_bias) {
    float* red = smem;   // [256/DTPR][BCO]
    red[d_r * BCO + d_c4 + 0] = bsum.x;
    red[d_r * BCO + d_c4 + 1] = bsum.y;
    red[d_r * BCO + d_c4 + 2] = bsum.z;
    red[d_r * BCO + d_c4 + 3] = bsum.w;
    __syncthreads();
    if (tid < BCO && co0 + tid < g.n_ch) {
      float a = 0.f;
      for (int r = 0; r < 256 / DTPR; ++r) a += red[r * BCO + tid];
      unsafeAtomicAdd(p.dbias + co0 + tid, a);
    }
  }
}

template <int BCO, int BCI, int TT, int WCO, int WCI, int WT, typename T = float>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs p) {
  if constexpr (sizeof(T) == 4) {
    if (p.dout_aligned)
      conv_wgrad_body<BCO, BCI, TT, WCO, WCI, WT, true, T>(p);
    else
      conv_wgrad_body<BCO, BCI, TT, WCO, WCI, WT, false, T>(p);
  } else {
    conv_wgrad_body<BCO, BCI, TT, WCO, WCI, WT, true, T>(p);    // (the launcher requires alignment)
  }
}

// ------------------------------------------------------------------------------------------
// weight gradient of the stride-1 3-tap 1-D convolutions (3x1 / 1x3: 79 % of the model's MACs)
// ------------------------------------------------------------------------------------------
// Pixels (GEMM K) are enumerated ALONG the convolution direction (b fastest, length L), so the
// three taps of output pixel k read input pixels k-1, k, k+1: one halo'd x tile [PK+2][BCI] in
// LDS serves all taps as row shifts (the generic kernel loads one x tile per tap).  Taps that
// would cross a line end are masked per k with two ballot masks.  Half the global loads, LDS
// writes and LDS footprint of conv_wgrad_kernel<64,64,3>.
struct Wgrad1dArgs {
  const float* in;
  const float* dout;
  float* dw;
  float* dbias;
  int M, L, n_ch, k_ch;              // M = n_img*A*L pixels, L = line length along the conv
  int Lr;                            // real line length: L = Lr, or Lr + 1 (an odd line gets one
                                     // virtual zero pixel so that the Winograd pairs never
                                     // straddle lines; positions >= Lr read as zero)
  int in_sa, in_sb, in_simg;         // element strides of x for (a, b, img)
  int dy_sa, dy_sb, dy_simg;         // element strides of dy
  int steps_total, steps_per_split, n_co_tiles, n_ci_tiles, n_tiles;
  int R, A;                          // kernel rows (1, or 3 for a 3x3 conv: one per workgroup), lines per image
  float* ws;                         // NULL: atomics into dw; else split-K partial tiles (see
                                     // wgrad1d_reduce_kernel), ws_bias behind them
  float* ws_bias;
  uint32_t in_bytes, dout_bytes;
  FastDiv div_al, div_l;
  // 16-bit direct kernel, MODE 2 (stride-2 3-tap conv): x element stride of ONE pixel along the
  // line (in_sb is then twice that) and the line length of x (the odd pixels 2b+1 may run past it)
  int in_sb1, in_Lx;
  int taps;                          // accumulator tiles per workgroup: 3, or 1 (1x1 convs, MODE 1)
  // 16-bit direct kernel: x line of output line a and kernel row kr = a * x_mul_a + kr * dl_mul +
  // dl_off, valid inside [0, in_Ax); its element stride is in_sa1 (the 7x7 / 2 stem as seven
  // single-tap rows over the packed input; 3x3: x_mul_a 1, dl_mul 1, dl_off -1)
  int x_mul_a, in_Ax, in_sa1, dl_mul, dl_off;
  // XBN (emsa_conv_wgrad_inbn): the x operand is a = relu(in * in_scale[c] + in_shift[c]) formed
  // in the loader -- the weight gradient of a conv whose forward ran with the BatchNorm + ReLU of
  // its input folded into ITS loader (emsa_conv1d_wino_inbn); `in` is the BatchNorm's input
  const float* in_scale;
  const float* in_shift;
};

// (tried: A operand by ds_read_b64 over even/odd channel tiles, waves splitting K -- halves the LDS
//  read instructions but needs 96 accumulator registers -> 2 waves/SIMD -> 57 vs 77 TFLOP/s;
//  occupancy beats instruction count with the 64-cycle fp32 MFMA.)
#ifndef EMSA_W1D_WPE
#define EMSA_W1D_WPE 4
#endif
#ifndef EMSA_W1DW_WPE
#define EMSA_W1DW_WPE 4   // waves per SIMD the Winograd weight-gradient kernel is compiled for
#endif
#ifndef EMSA_W1D_XCD
#define EMSA_W1D_XCD 1
#endif
// direct form (EMSA_WGRAD_WINO=0): three MFMAs per pixel and accumulator tile, taps masked at the
// line ends with two ballot masks.  The default is the Winograd form below.
template <int BCO, int BCI>
__global__ __launch_bounds__(256, EMSA_W1D_WPE) void conv_wgrad1d_kernel(
    const Wgrad1dArgs p) {
  static_assert(BCO == 64 && BCI == 64, "wave layout below is for a 64x64 (co x ci) tile");
  constexpr int PK = 32, XROWS = PK + 2;   // pixels per K step (64 spills: measured slower)
  constexpr int DTPR = BCO / 4, XTPR = BCI / 4;
  constexpr int DR = PK * DTPR / 256;
  constexpr int XR = (XROWS * XTPR + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const dS = smem;                  // [PK][BCO]
  float* const xS = smem + PK * BCO;       // [XROWS][BCI]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wco = wave & 1, wci = wave >> 1;
  constexpr int NQ = 1;
  // blockIdx = (split * R + kernel row) * n_tiles + tile
  const int tile = blockIdx.x % p.n_tiles, kr = (blockIdx.x / p.n_tiles) % p.R;
  const int ks = blockIdx.x / (p.n_tiles * p.R);
  const int dline = p.R == 3 ? kr - 1 : 0;         // x is read `dline` lines away (3x3 row tap)
  const int ci_t = tile % p.n_ci_tiles, co_t = tile / p.n_ci_tiles;
  const int co0 = co_t * BCO, ci0 = ci_t * BCI;
  const int s_begin = ks * p.steps_per_split;
  const int s_end = min(s_begin + p.steps_per_split, p.steps_total);
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes);
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(p.dout, p.dout_bytes);
  const int d_c4 = (tid % DTPR) * 4, d_r = tid / DTPR;
  const int x_c4 = (tid % XTPR) * 4, x_r = tid / XTPR;
  const bool do_bias = p.dbias != nullptr && ci_t == 0 && kr == 0;

  // Addresses of the step's pixels k0 - 1 .. k0 + 32: every row of the two tiles is one pixel,
  // shared by the 16 lanes that load its channels, so lane l of each wave decomposes pixel
  // k0 + l - 1 ONCE (image, line, position -> byte offsets into x and dy, or kOOB) and the
  // loading threads fetch their rows' offsets with wave shuffles (ds_bpermute).  ~45 VALU
  // instructions per step; per-thread decomposition of each of the five loads was ~350, as much
  // issue time as the step's 32 MFMAs.
  // x is read `dline` lines away (3x3 row tap; a shifted line outside the image is zero padding)
  uint32_t mleft = 0, mright = 0, mleft_n = 0, mright_n = 0;
  float4 rd[DR], rx[XR];
  float4 bsum = emsa_zero4();
  auto load_regs = [&](int s) {
    const int k = s * PK + lane - 1;
    const bool valid = lane < XROWS && k >= 0 && k < p.M;
    const uint32_t ku = valid ? (uint32_t)k : 0u;
    const uint32_t img = fast_div(ku, p.div_al);
    const uint32_t line = fast_div(ku, p.div_l);                 // = img * A + a
    const int b_ = (int)(ku - __umul24(line, (uint32_t)p.L));
    const int a_ = (int)(line - __umul24(img, (uint32_t)p.A));
    const int a2 = a_ + dline;
    const bool in_line = valid && b_ < p.Lr;
    const uint32_t off_d = in_line
        ? (__umul24(img, (uint32_t)p.dy_simg) + __umul24((uint32_t)a_, (uint32_t)p.dy_sa) +
           __umul24((uint32_t)b_, (uint32_t)p.dy_sb)) * 4u : kOOB;
    const uint32_t off_x = (in_line && a2 >= 0 && a2 < p.A)
        ? (__umul24(img, (uint32_t)p.in_simg) + __umul24((uint32_t)a2, (uint32_t)p.in_sa) +
           __umul24((uint32_t)b_, (uint32_t)p.in_sb)) * 4u : kOOB;
    // line-end masks of the step's 32 pixels (lanes 1..32): bit r set = tap 0 (left) / tap 2
    // (right) of pixel k0 + r crosses a line end
    const bool px = lane >= 1 && lane <= PK;
    mleft_n = (uint32_t)(__ballot(px && b_ == 0) >> 1);
    mright_n = (uint32_t)(__ballot(px && b_ == p.L - 1) >> 1);
    const uint32_t cob = (co0 + d_c4) < p.n_ch ? (uint32_t)(co0 + d_c4) * 4u : kOOB;
    const uint32_t cib = (ci0 + x_c4) < p.k_ch ? (uint32_t)(ci0 + x_c4) * 4u : kOOB;
#pragma unroll
    for (int j = 0; j < DR; ++j) {
      const int r = d_r + j * (256 / DTPR);                      // dy row r = pixel k0 + r
      const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute((r + 1) * 4, (int)off_d);
#if EMSA_ABL & 1
      rd[j] = make_float4(o, cob, 1.f, 2.f);
#else
      rd[j] = buf_ld4(rs_dy, ((o | cob) & kOOB) ? kOOB : o + cob);
#endif
    }
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int r = x_r + j * (256 / XTPR);                      // x row r = pixel k0 + r - 1
      uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute((r & 63) * 4, (int)off_x);
      o = r < XROWS ? o : kOOB;
#if EMSA_ABL & 1
      rx[j] = make_float4(o, cib, 1.f, 2.f);
#else
      rx[j] = buf_ld4(rs_in, ((o | cib) & kOOB) ? kOOB : o + cib);
#endif
    }
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int j = 0; j < DR; ++j) {
      // (bias gradient: summed where the prefetched registers are consumed, never right behind
      //  the load -- that would stall this step's MFMAs on the NEXT step's data)
      bsum.x += rd[j].x; bsum.y += rd[j].y; bsum.z += rd[j].z; bsum.w += rd[j].w;
      emsa_st4(dS + (d_r + j * (256 / DTPR)) * BCO + d_c4, rd[j]);
    }
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int r = x_r + j * (256 / XTPR);
      if (r < XROWS) emsa_st4(xS + r * BCI + x_c4, rx[j]);
    }
  };

  constexpr int NACC = 3;
  f32x16 acc[NQ][NACC];
#pragma unroll
  for (int q = 0; q < NQ; ++q)
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][t][r] = 0.f;

  if (s_begin < s_end) {
    load_regs(s_begin);
    store_lds();
  }
  __syncthreads();

  for (int s = s_begin; s < s_end; ++s) {
    const bool has_next = s + 1 < s_end;
    mleft = mleft_n;                         // masks of THIS step (set when it was loaded)
    mright = mright_n;
    if (has_next) load_regs(s + 1);
    const float* d = dS + lh * BCO + wco * 32 + l31;
    const float* x = xS + lh * BCI + wci * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < PK / 2; ++kk) {
      const int r = 2 * kk + lh;
      const float fa = d[kk * 2 * BCO];
      const bool okl = !((mleft >> r) & 1u), okr = !((mright >> r) & 1u);
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        float fb = x[(kk * 2 + t) * BCI];
        if (t == 0) fb = okl ? fb : 0.f;
        if (t == 2) fb = okr ? fb : 0.f;
        acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[0][t], 0, 0, 0);
      }
    }
    __syncthreads();
    if (has_next) store_lds();
    __syncthreads();
  }

  if (p.ws != nullptr) {
    // deterministic split-K: this workgroup's partial tile [3][BCO][BCI] goes to the workspace
    // with plain (coalesced along ci) stores; wgrad1d_reduce_kernel sums the splits
    float* wt = p.ws + (((size_t)ks * p.R + kr) * p.n_tiles + tile) * (3 * BCO * BCI);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        wt[(t * BCO + wco * 32 + row) * BCI + wci * 32 + l31] = acc[0][t][r];
      }
  } else {
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int ci = ci0 + wci * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int co = co0 + wco * 32 + row;
#if EMSA_ABL & 2
        if (co < p.n_ch && ci < p.k_ch && acc[q][t][r] == 1.2345e30f)
#else
        if (co < p.n_ch && ci < p.k_ch)
#endif
          unsafeAtomicAdd(p.dw + ((size_t)(kr * 3 + t) * p.n_ch + co) * p.k_ch + ci, acc[q][t][r]);
      }
    }
  }
  if (do_bias) {
    float* red = smem;   // [256/DTPR][BCO]
    __syncthreads();     // (the K loop's LDS reads are done; red overlays dS)
    red[d_r * BCO + d_c4 + 0] = bsum.x;
    red[d_r * BCO + d_c4 + 1] = bsum.y;
    red[d_r * BCO + d_c4 + 2] = bsum.z;
    red[d_r * BCO + d_c4 + 3] = bsum.w;
    __syncthreads();
    if (tid < BCO) {
      float a = 0.f;
      for (int r = 0; r < 256 / DTPR; ++r) a += red[r * BCO + tid];
      if (p.ws_bias != nullptr)
        p.ws_bias[((size_t)ks * p.n_co_tiles + co_t) * BCO + tid] = a;
      else if (co0 + tid < p.n_ch)
        unsafeAtomicAdd(p.dbias + co0 + tid, a);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Winograd F(3,2) weight gradient with the operand transforms done ONCE, when the tile enters LDS
// ------------------------------------------------------------------------------------------
// Over pixel PAIRS along the line (even line length; an odd line gets one virtual zero pixel):
//   dW_t = sum_pairs sum_i e_i d_(i+t),  e_i = dy(2p+i), d_r = x(2p-1+r)   ->   4 products per pair
//   E = (e0, e0+e1, e0-e1, -e1)   D = (d0-d2, d1+d2, d2-d1, d1-d3)   M_k = sum_pairs E_k (x) D_k
//   dW_0 = M0 + (M1+M2)/2    dW_1 = (M1-M2)/2    dW_2 = (M1+M2)/2 + M3
// i.e. 4 MFMAs per pixel pair instead of 6 (the transposed form of the forward F(2,3) kernel in
// conv_wino.hip); the output transform is applied to the accumulators in registers.
// On gfx950 the fp32 MFMAs and the vector ALU are the same hardware: every VALU instruction takes
// its four cycles away from the matrix pipe, whichever wave issues it (tools/mfma_peak.hip:
// 156 TFLOP/s with none, 125 with four FMAs per MFMA).  An earlier version formed
// E and D in every consuming wave and
// masked the line ends per pixel pair: ~14 VALU instructions per 4 MFMAs plus ~100 per K step of
// addressing -> 69 % of the MFMA peak at best.  Here each loader thread owns one pixel PAIR and
// four channels: it loads the pair's two dy rows and four x rows, transforms them with packed
// math and stores E and D; the MFMA loop only reads LDS.  Line ends need no masks: the loads of
// d0 / d3 across a line end are sent out of range and read as zero.  ~80 VALU per K step of 32
// MFMAs.
// BF16 (EMSA_BF16_MFMA=1, NOT the default and never the headline: BASELINE config 3's mixed
// precision): E and D are rounded to bf16 as they enter LDS ([comp][channel][pair], pair fastest)
// and one v_mfma_f32_32x32x16_bf16 per component covers the step's 16 pixel pairs; accumulation,
// the transforms and everything in HBM stay fp32.
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
#ifndef EMSA_W1DW_XBN_WPE
#define EMSA_W1DW_XBN_WPE 3   // XBN needs ~10 registers more than the 128 of four waves per SIMD
#endif
template <int BCO, int BCI, bool BF16 = false, typename T = float, bool XBN = false>
__global__ __launch_bounds__(256, XBN ? EMSA_W1DW_XBN_WPE : EMSA_W1DW_WPE) void conv_wgrad1d_wino_kernel(const Wgrad1dArgs p) {
  static_assert(BCO == 64 && BCI == 64, "wave layout below is for a 64x64 (co x ci) tile");
  static_assert(!XBN || (!BF16 && sizeof(T) == 4), "the folded input BatchNorm is an fp32 form");
  constexpr uint32_t ES = sizeof(T);            // bytes per activation element
  constexpr int PK = 32, NP = PK / 2, XROWS = PK + 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const eS = smem;                      // [NP pairs][4: e0, e0+e1, e0-e1, e1][BCO]
  float* const dS = smem + NP * 4 * BCO;       // [NP pairs][4: D0..D3][BCI]
  // BF16: [4 comps][64 channels][kBL = 20 (16 pairs + pad: 8-byte aligned rows)] bf16 each
  constexpr int kBL = 20;
  __bf16* const eH = reinterpret_cast<__bf16*>(smem);
  __bf16* const dH = eH + 4 * BCO * kBL;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wco = wave & 1, wci = wave >> 1;
  // work item = (split * R + kernel row) * n_tiles + tile; every XCD (own L2) gets a contiguous
  // range of items, i.e. ALL the tiles of a few pixel ranges: x and dy of a range are then
  // fetched into one L2 once instead of into all eight (PMC FETCH_SIZE was 2x the algorithmic
  // bytes with the round-robin order)
#if EMSA_W1D_XCD
  const int wg = emsa_xcd_remap(blockIdx.x, gridDim.x);
#else
  const int wg = blockIdx.x;
#endif
  const int tile = wg % p.n_tiles, kr = (wg / p.n_tiles) % p.R;
  const int ks = wg / (p.n_tiles * p.R);
  const int dline = p.R == 3 ? kr - 1 : 0;         // x is read `dline` lines away (3x3 row tap)
  const int ci_t = tile % p.n_ci_tiles, co_t = tile / p.n_ci_tiles;
  const int co0 = co_t * BCO, ci0 = ci_t * BCI;
  const int s_begin = ks * p.steps_per_split;
  const int s_end = min(s_begin + p.steps_per_split, p.steps_total);
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes);
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(p.dout, p.dout_bytes);
  const int col4 = (tid & 15) * 4, pr = tid >> 4;  // this thread's float4 column and pixel pair
  const bool do_bias = p.dbias != nullptr && ci_t == 0 && kr == 0;
  // channel part of the addresses; a column beyond the tensor's channels stays out of range
  const bool dok = co0 + col4 < p.n_ch, xok = ci0 + col4 < p.k_ch;
  const uint32_t dadd = dok ? (uint32_t)(co0 + col4) * ES : 0u, dmask = dok ? 0u : kOOB;
  const uint32_t xadd = xok ? (uint32_t)(ci0 + col4) * ES : 0u, xmask = xok ? 0u : kOOB;

  typename Raw4<T>::type re_raw[2], rx_raw[4];
  float4 bsum = emsa_zero4();
  // XBN: bit r set = x row d_r of the prefetched step is a real pixel (out-of-range rows -- line
  // ends, image border, the virtual pixel of an odd line -- are ZERO after the BatchNorm + ReLU too)
  uint32_t x_real = 0;
  float* const xbnS = smem + NP * 4 * (BCO + BCI);   // XBN: [2: scale, shift][BCI]
  if constexpr (XBN) {
    if (tid < 2 * BCI) {
      const int ch = ci0 + (tid & (BCI - 1));
      const float* src = tid < BCI ? p.in_scale : p.in_shift;
      xbnS[tid] = ch < p.k_ch ? src[ch] : 0.f;
    }
  }
  auto load_regs = [&](int s) {
    // lane l of every wave decomposes pixel k0 + l - 1 (l < 34) once; the loading threads fetch
    // their rows' byte offsets with wave shuffles
    const int k = s * PK + lane - 1;
    const bool valid = lane < XROWS && k >= 0 && k < p.M;
    const uint32_t ku = valid ? (uint32_t)k : 0u;
    const uint32_t img = fast_div(ku, p.div_al);
    const uint32_t line = fast_div(ku, p.div_l);                 // = img * A + a
    const int b_ = (int)(ku - __umul24(line, (uint32_t)p.L));
    const int a_ = (int)(line - __umul24(img, (uint32_t)p.A));
    const int a2 = a_ + dline;
    const bool in_line = valid && b_ < p.Lr;
    const uint32_t off_d = in_line
        ? (__umul24(img, (uint32_t)p.dy_simg) + __umul24((uint32_t)a_, (uint32_t)p.dy_sa) +
           __umul24((uint32_t)b_, (uint32_t)p.dy_sb)) * ES : kOOB;
    const uint32_t off_x = (in_line && a2 >= 0 && a2 < p.A)
        ? (__umul24(img, (uint32_t)p.in_simg) + __umul24((uint32_t)a2, (uint32_t)p.in_sa) +
           __umul24((uint32_t)b_, (uint32_t)p.in_sb)) * ES : kOOB;
    // as the LEFT neighbour (d0) of the next pixel a line's last pixel reads zero, as the RIGHT
    // neighbour (d3) of the previous pixel a line's first pixel does: the tap crosses a line end
    const uint32_t off_xl = b_ == p.L - 1 ? kOOB : off_x;
    const uint32_t off_xr = b_ == 0 ? kOOB : off_x;
    const int l0 = (2 * pr) * 4;                  // lane of pixel k0 + 2*pr - 1, times 4
    const uint32_t o_e0 = (uint32_t)__builtin_amdgcn_ds_bpermute(l0 + 4, (int)off_d);
    const uint32_t o_e1 = (uint32_t)__builtin_amdgcn_ds_bpermute(l0 + 8, (int)off_d);
    const uint32_t o_d0 = (uint32_t)__builtin_amdgcn_ds_bpermute(l0, (int)off_xl);
    const uint32_t o_d1 = (uint32_t)__builtin_amdgcn_ds_bpermute(l0 + 4, (int)off_x);
    const uint32_t o_d2 = (uint32_t)__builtin_amdgcn_ds_bpermute(l0 + 8, (int)off_x);
    const uint32_t o_d3 = (uint32_t)__builtin_amdgcn_ds_bpermute(l0 + 12, (int)off_xr);
    re_raw[0] = buf_ld4raw<T>(rs_dy, (o_e0 + dadd) | dmask);
    re_raw[1] = buf_ld4raw<T>(rs_dy, (o_e1 + dadd) | dmask);
    rx_raw[0] = buf_ld4raw<T>(rs_in, (o_d0 + xadd) | xmask);
    rx_raw[1] = buf_ld4raw<T>(rs_in, (o_d1 + xadd) | xmask);
    rx_raw[2] = buf_ld4raw<T>(rs_in, (o_d2 + xadd) | xmask);
    rx_raw[3] = buf_ld4raw<T>(rs_in, (o_d3 + xadd) | xmask);
    if constexpr (XBN)
      x_real = (~o_d0 >> 31) | ((~o_d1 >> 31) << 1) | ((~o_d2 >> 31) << 2) | ((~o_d3 >> 31) << 3);
  };
  auto add4 = [](const float4& a, const float4& b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  };
  auto sub4 = [](const float4& a, const float4& b) {
    return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
  };
  auto store_lds = [&]() {
    const float4 re[2] = {raw_f4(re_raw[0], (T*)nullptr), raw_f4(re_raw[1], (T*)nullptr)};
    float4 rx[4] = {raw_f4(rx_raw[0], (T*)nullptr), raw_f4(rx_raw[1], (T*)nullptr),
                    raw_f4(rx_raw[2], (T*)nullptr), raw_f4(rx_raw[3], (T*)nullptr)};
    if constexpr (XBN) {
      // a = relu(x * scale + shift) exactly as the forward loader formed it (one FMA, then
      // v_med3(v, 0, cap): max(v, 0) for cap = +inf, 0 for an out-of-range row)
      typedef float f2 __attribute__((ext_vector_type(2)));
      const float4 sc = emsa_ld4(xbnS + col4), sh = emsa_ld4(xbnS + BCI + col4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float cap = (x_real >> r) & 1u ? __builtin_inff() : 0.f;
        const f2 lo = __builtin_elementwise_fma(f2{rx[r].x, rx[r].y}, f2{sc.x, sc.y}, f2{sh.x, sh.y});
        const f2 hi = __builtin_elementwise_fma(f2{rx[r].z, rx[r].w}, f2{sc.z, sc.w}, f2{sh.z, sh.w});
        rx[r] = make_float4(__builtin_amdgcn_fmed3f(lo.x, 0.f, cap), __builtin_amdgcn_fmed3f(lo.y, 0.f, cap),
                            __builtin_amdgcn_fmed3f(hi.x, 0.f, cap), __builtin_amdgcn_fmed3f(hi.y, 0.f, cap));
      }
    }
    float* e = eS + pr * 4 * BCO + col4;
    float* d = dS + pr * 4 * BCI + col4;
    const float4 es = add4(re[0], re[1]);
    bsum = add4(bsum, es);                       // bias gradient = column sums of dy
    if constexpr (BF16) {
      auto st4h = [&](__bf16* o, const float4& v) {
        o[0] = (__bf16)v.x; o[kBL] = (__bf16)v.y; o[2 * kBL] = (__bf16)v.z; o[3 * kBL] = (__bf16)v.w;
      };
      __bf16* eo = eH + col4 * kBL + pr;
      __bf16* dq = dH + col4 * kBL + pr;
      st4h(eo, re[0]);
      st4h(eo + BCO * kBL, es);
      st4h(eo + 2 * BCO * kBL, sub4(re[0], re[1]));
      st4h(eo + 3 * BCO * kBL, re[1]);
      st4h(dq, sub4(rx[0], rx[2]));
      st4h(dq + BCI * kBL, add4(rx[1], rx[2]));
      st4h(dq + 2 * BCI * kBL, sub4(rx[2], rx[1]));
      st4h(dq + 3 * BCI * kBL, sub4(rx[1], rx[3]));
      return;
    }
    emsa_st4(e, re[0]);
    emsa_st4(e + BCO, es);
    emsa_st4(e + 2 * BCO, sub4(re[0], re[1]));
    emsa_st4(e + 3 * BCO, re[1]);
    emsa_st4(d, sub4(rx[0], rx[2]));
    emsa_st4(d + BCI, add4(rx[1], rx[2]));
    emsa_st4(d + 2 * BCI, sub4(rx[2], rx[1]));
    emsa_st4(d + 3 * BCI, sub4(rx[1], rx[3]));
  };

  f32x16 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (s_begin < s_end) {
    load_regs(s_begin);
    if constexpr (XBN) __syncthreads();            // the affine table is in LDS
    store_lds();
  }
  __syncthreads();

#if EMSA_ABL & 8
  for (int s = s_begin; s < s_begin + 1 && s < s_end; ++s) {   // tuning: one K step only
#else
  for (int s = s_begin; s < s_end; ++s) {
#endif
    const bool has_next = s + 1 < s_end;
    if (has_next) load_regs(s + 1);
    // K index of the MFMA = pixel pair: lane half lh takes pair 2*kk + lh of the step's 16
    if constexpr (BF16) {
      // lane (l31, lh): channel row l31 of its 32x32 tile, pairs 8*lh .. 8*lh + 7 (16 bytes)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint2* ea = reinterpret_cast<const uint2*>(eH + (c * BCO + wco * 32 + l31) * kBL + 8 * lh);
        const uint2* da = reinterpret_cast<const uint2*>(dH + (c * BCI + wci * 32 + l31) * kBL + 8 * lh);
        const uint2 e0 = ea[0], e1 = ea[1], d0 = da[0], d1 = da[1];
        const uint4 eu = make_uint4(e0.x, e0.y, e1.x, e1.y), du = make_uint4(d0.x, d0.y, d1.x, d1.y);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gbf16x8, eu),
                                                         __builtin_bit_cast(gbf16x8, du), acc[c],
                                                         0, 0, 0);
      }
    } else {
    const float* e = eS + lh * 4 * BCO + wco * 32 + l31;
    const float* d = dS + lh * 4 * BCI + wci * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < NP / 2; ++kk) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(e[(kk * 8 + c) * BCO], d[(kk * 8 + c) * BCI],
                                                      acc[c], 0, 0, 0);
    }
    }
    __syncthreads();
    if (has_next) store_lds();
    __syncthreads();
  }

  // output transform G^T on the accumulators (M3 was accumulated with +e1, i.e. negated):
  //   dW_0 = M0 + (M1+M2)/2    dW_1 = (M1-M2)/2    dW_2 = (M1+M2)/2 - M3'
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float m0 = acc[0][r], m1 = acc[1][r], m2 = acc[2][r], m3 = acc[3][r];
    const float hs = 0.5f * (m1 + m2);
    acc[0][r] = m0 + hs;
    acc[1][r] = 0.5f * (m1 - m2);
    acc[2][r] = hs - m3;
  }

  if (p.ws != nullptr) {
    // deterministic split-K: this workgroup's partial tile [3][BCO][BCI] goes to the workspace
    float* wt = p.ws + (((size_t)ks * p.R + kr) * p.n_tiles + tile) * (3 * BCO * BCI);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        wt[(t * BCO + wco * 32 + row) * BCI + wci * 32 + l31] = acc[t][r];
      }
  } else {
    const int ci = ci0 + wci * 32 + l31;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int co = co0 + wco * 32 + row;
#if EMSA_ABL & 2
        if (co < p.n_ch && ci < p.k_ch && acc[t][r] == 1.2345e30f)
#else
        if (co < p.n_ch && ci < p.k_ch)
#endif
          unsafeAtomicAdd(p.dw + ((size_t)(kr * 3 + t) * p.n_ch + co) * p.k_ch + ci, acc[t][r]);
      }
  }
  if (do_bias) {
    float* red = smem;   // [NP][BCO]
    __syncthreads();     // (the K loop's LDS reads are done; red overlays eS)
    red[pr * BCO + col4 + 0] = bsum.x;
    red[pr * BCO + col4 + 1] = bsum.y;
    red[pr * BCO + col4 + 2] = bsum.z;
    red[pr * BCO + col4 + 3] = bsum.w;
    __syncthreads();
    if (tid < BCO) {
      float a = 0.f;
      for (int r = 0; r < NP; ++r) a += red[r * BCO + tid];
      if (p.ws_bias != nullptr)
        p.ws_bias[((size_t)ks * p.n_co_tiles + co_t) * BCO + tid] = a;
      else if (co0 + tid < p.n_ch)
        unsafeAtomicAdd(p.dbias + co0 + tid, a);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the stride-1 3-tap convolutions from 16-bit activations (bf16 training path)
// ------------------------------------------------------------------------------------------
// dW_t[co][ci] = sum_p dy[p][co] x[p + t - 1][ci] on v_mfma_f32_32x32x16_bf16: the MFMA K index
// is the PIXEL, so both operands must be K-contiguous per channel -- the transpose of the NHWC
// rows they are loaded as.  Each loader thread owns 4 consecutive pixels x 4 channels of BOTH
// tensors (four coalesced 8-byte loads each: 16 lanes cover one 128-byte pixel row), transposes the
// 4x4 blocks in registers (8 v_perm_b32 each) and writes four 8-byte rows of the [channel][pixel]
// LDS images.  One x image and one dy image serve all three taps: the left tap dy(p) x(p - 1) reads
// x shifted by one pixel, the right tap is summed as dy(p - 1) x(p) -- the same set of products
// over the padded enumeration -- and reads dy shifted by one pixel.  A shifted fragment is rebuilt
// from the centre registers with four v_alignbit_b32 and the dword of its partner lane (the other
// 8-pixel half of the row: v_permlane32_swap), so LDS holds every element once (the Winograd
// F(3,2) form stores four transformed copies of each and rounds them to bf16 once more).  Both
// shifts want the pixel in FRONT of a step: the last pixel of the previous step, written into the
// rows' halo dword from the registers that step was loaded into (only a split's first step loads it).
// Lines are enumerated with one virtual zero pixel behind every line (L = Lr + 1), so taps never
// cross a line end and no masks exist.  64 pixels per K step: 12 MFMAs per wave between barriers.
// Direct products of the stored bf16 values with fp32 accumulation: as exact as the fp32 kernel on
// the same inputs.
// Measured (bs=32 layer shapes, kernel trace): 32-36 us at EVERY channel count + 8 us of reduction
// pass (the Winograd bf16 kernel: 75-80 in all): a launch moves ~157 MB through the CUs' vector
// memory path whatever C is (HBM at C=64, L2 re-reads by the 64x64 tiles at C=512).  DESIGN.md 7
// (round 4) has the ablations, counters and phase timings: no single resource bounds the K loop;
// two K steps in flight, 128-pixel steps and (round 3) 16-byte loads were all measured no faster.
// EMSA_WH_DBG=1 (tools/wgrad_phases.py builds only): lane 0 of every wave accumulates the shader-clock
// time of each phase of its K loop; emsa_wgrad1d_h_dbg_read returns the table [wg][wave][8]
#ifndef EMSA_WH_DBG
#define EMSA_WH_DBG 0
#endif
#if EMSA_WH_DBG
__device__ long long g_wh_dbg[8 * 4 * 4096];
#define WH_MARK(ph)                                               \
  do {                                                            \
    const long long t_ = (long long)__builtin_readcyclecounter(); \
    dbg_t[ph] += t_ - dbg_prev;                                   \
    dbg_prev = t_;                                                \
  } while (0)
#else
#define WH_MARK(ph) do {} while (0)
#endif
// (the device pass only: the host pass does not know the feature)
#if defined(__HIP_DEVICE_COMPILE__)
#define EMSA_NO_LSOPT __attribute__((target("no-load-store-opt")))
#else
#define EMSA_NO_LSOPT
#endif
#ifndef EMSA_WH_PK
#define EMSA_WH_PK 64
#endif
constexpr int kWH_PK = EMSA_WH_PK;  // pixels per K step (64 or 128): 24 MFMAs per wave between barriers
constexpr int kWH_NH = kWH_PK / 64; // 64-pixel halves of a step (a thread loads 4 pixels of each)
constexpr int kWH_ROW = kWH_PK + 4; // LDS row (elements): the pixels + halo dword + pad = 136 / 264 B
                                    // = 2 banks mod 32: the transposing ds_write_b64 of a 16-lane group
                                    // (4 channel quads x 4 pixel groups, 32-bank rule) hit 16 distinct
                                    // bank pairs and the fragment reads, two ds_read_b64 per operand (rows
                                    // are 8-byte aligned; 64-bank rule, 32-lane groups), 32 distinct ones.
                                    // (Round 3 used 144 B rows and ds_read_b128: the stores were 2-way
                                    // conflicted and the dword reads of the shifted taps 4-way --
                                    // SQ_LDS_BANK_CONFLICT was 44 % of SQ_LDS_IDX_ACTIVE.)
// MODE 0: stride-1 3-tap convs (and the row taps of a 3x3).  MODE 1: ONE tap -- the 1x1 convs
// (skip fusions, pyramid pooling, strided down-sampling shortcuts: the stride lives in the x pixel
// strides, nothing else changes).  MODE 2: stride-2 3-tap convs (first convs of a down-sampling
// NBt1D block): output pixel b reads x(2b-1), x(2b), x(2b+1) -> TWO x images, the even pixels
// E(b) = x(2b) (centre tap) and the odd pixels O(b) = x(2b+1) (right tap; the left tap is O shifted
// by one element, O(b-1)).  Round 2 ran modes 1 / 2 on the fp32-MFMA tap-group kernel with a
// bf16 -> fp32 conversion at the LDS store (3.5 ms of the 51 ms bf16 step at 78-111 TFLOP/s).
template <typename T, int MODE = 0>
__global__ __launch_bounds__(256, 3) EMSA_NO_LSOPT void conv_wgrad1d_h_kernel(const Wgrad1dArgs p) {
  constexpr int BCO = 64, BCI = 64;
  constexpr uint32_t ES = sizeof(T);
  constexpr int NT_ = MODE == 1 ? 1 : 3;             // accumulator tiles (taps)
  typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  T* const dS = reinterpret_cast<T*>(smem);          // [64 co][kWH_ROW]: pixel k0 + e at e < 64
  T* const xS = dS + BCO * kWH_ROW;                  // [64 ci][kWH_ROW]: same; elements 64 / 65 =
                                                     // pixels k0 + 64 / k0 - 1 (the halo dword)
  T* const oS = xS + BCI * kWH_ROW;                  // MODE 2: the odd-pixel image (halo: k0 - 1)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int wco = wave & 1, wci = wave >> 1;
#if EMSA_W1D_XCD
  const int wg = emsa_xcd_remap(blockIdx.x, gridDim.x);
#else
  const int wg = blockIdx.x;
#endif
  const int tile = wg % p.n_tiles, kr = (wg / p.n_tiles) % p.R;
  const int ks = wg / (p.n_tiles * p.R);
  const int dline = kr * p.dl_mul + p.dl_off;      // x line offset of this kernel row
  const int ci_t = tile % p.n_ci_tiles, co_t = tile / p.n_ci_tiles;
  const int co0 = co_t * BCO, ci0 = ci_t * BCI;
  const int s_begin = ks * p.steps_per_split;
  const int s_end = min(s_begin + p.steps_per_split, p.steps_total);
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes);
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(p.dout, p.dout_bytes);
  // loader unit: pixel group pg (4 pixels) x channel quad q.  A 16-lane group holds 4 quads x 4
  // pixel groups: the ds_write_b64 of one group then hit 16 distinct bank pairs (rows 4q + c are
  // 16 banks apart, pixel groups 2 banks), while every load instruction of the wave still covers
  // whole 128-byte pixel rows (16 quads x 8 B).
  const int q = (lane & 3) | ((lane >> 4) << 2);       // 0..15
  const int pg = wave * 4 + ((lane >> 2) & 3);         // 0..15
  const bool do_bias = p.dbias != nullptr && ci_t == 0 && kr == 0;
  const bool dok = co0 + 4 * q < p.n_ch, xok = ci0 + 4 * q < p.k_ch;
  const uint32_t dadd = dok ? (uint32_t)(co0 + 4 * q) * ES : 0u, dmask = dok ? 0u : kOOB;
  const uint32_t xadd = xok ? (uint32_t)(ci0 + 4 * q) * ES : 0u, xmask = xok ? 0u : kOOB;

  // byte offsets (or kOOB) of pixel k of the padded enumeration in dy (od) and in x (ox); MODE 2:
  // ox = the even pixel x(2b), oo = the odd pixel x(2b+1) (out of range behind the line's end)
  uint32_t oo_last = kOOB;
  auto px_off = [&](int k, uint32_t& od, uint32_t& ox) {
    const bool valid = k >= 0 && k < p.M;
    const uint32_t ku = valid ? (uint32_t)k : 0u;
    const uint32_t img = fast_div(ku, p.div_al);
    const uint32_t line = fast_div(ku, p.div_l);                 // = img * A + a
    const int b_ = (int)(ku - __umul24(line, (uint32_t)p.L));
    const int a_ = (int)(line - __umul24(img, (uint32_t)p.A));
    const int a2 = a_ * p.x_mul_a + dline;
    const bool in_line = valid && b_ < p.Lr;                     // the virtual pixel reads zero
    od = in_line ? (__umul24(img, (uint32_t)p.dy_simg) + __umul24((uint32_t)a_, (uint32_t)p.dy_sa) +
                    __umul24((uint32_t)b_, (uint32_t)p.dy_sb)) * ES : kOOB;
    ox = (in_line && a2 >= 0 && a2 < p.in_Ax)
        ? (__umul24(img, (uint32_t)p.in_simg) + __umul24((uint32_t)a2, (uint32_t)p.in_sa1) +
           __umul24((uint32_t)b_, (uint32_t)p.in_sb)) * ES : kOOB;
    if constexpr (MODE == 2)
      oo_last = (ox != kOOB && 2 * b_ + 1 < p.in_Lx) ? ox + (uint32_t)p.in_sb1 * ES : kOOB;
  };

  u32x2w rd[kWH_NH][4], rx[kWH_NH][4], ro[kWH_NH][4];
  // the pixel in front of a step (its four channels of this thread's quad): the LAST pixel of the
  // previous step, kept from that step's registers -- only the split's first step loads it
  u32x2w sv_d = {0u, 0u}, sv_x = {0u, 0u}, sv_o = {0u, 0u};
  float4 bsum = emsa_zero4();
  auto load_regs = [&](int s) {
    const int k0 = s * kWH_PK;
    // a wave loads pixels k0 + 16 wave .. + 15 (of each 64-pixel half): every 16-lane row decomposes those 16 once, and a
    // thread's four pixels are the four lanes of its own quad (DPP quad broadcasts; the round-3
    // form fetched them with 8-12 ds_bpermute per step through the LDS pipe, the kernel's bottleneck)
#pragma unroll
    for (int hh = 0; hh < kWH_NH; ++hh) {
      uint32_t off_d, off_x;
      px_off(k0 + hh * 64 + wave * 16 + (lane & 15), off_d, off_x);
      const uint32_t off_o = oo_last;
#define EMSA_WH_LOADJ(j)                                                                          \
      {                                                                                           \
        const uint32_t od = (uint32_t)__builtin_amdgcn_mov_dpp((int)off_d, (j) * 0x55, 0xf, 0xf, false); \
        const uint32_t ox = (uint32_t)__builtin_amdgcn_mov_dpp((int)off_x, (j) * 0x55, 0xf, 0xf, false); \
        rd[hh][j] = __builtin_amdgcn_raw_buffer_load_b64(rs_dy, (int)((od + dadd) | dmask), 0, 0); \
        rx[hh][j] = __builtin_amdgcn_raw_buffer_load_b64(rs_in, (int)((ox + xadd) | xmask), 0, 0); \
        if constexpr (MODE == 2) {                                                                \
          const uint32_t oo = (uint32_t)__builtin_amdgcn_mov_dpp((int)off_o, (j) * 0x55, 0xf, 0xf, false); \
          ro[hh][j] = __builtin_amdgcn_raw_buffer_load_b64(rs_in, (int)((oo + xadd) | xmask), 0, 0); \
        }                                                                                         \
      }
      EMSA_WH_LOADJ(0) EMSA_WH_LOADJ(1) EMSA_WH_LOADJ(2) EMSA_WH_LOADJ(3)
#undef EMSA_WH_LOADJ
    }
  };
  // the pixel in front of the split's first step
  auto load_front = [&](int s) {
    uint32_t hd, hx;
    px_off(s * kWH_PK - 1, hd, hx);
    if constexpr (MODE == 0) {
      sv_d = __builtin_amdgcn_raw_buffer_load_b64(rs_dy, (int)((hd + dadd) | dmask), 0, 0);
      sv_x = __builtin_amdgcn_raw_buffer_load_b64(rs_in, (int)((hx + xadd) | xmask), 0, 0);
    } else if constexpr (MODE == 2) {
      sv_o = __builtin_amdgcn_raw_buffer_load_b64(rs_in, (int)((oo_last + xadd) | xmask), 0, 0);
    }
  };
  // 4 pixels x 4 channels (one u32x2 = 4 channels per pixel) -> per channel 4 pixels = 8 bytes
  auto tr_store = [&](T* base, const u32x2w (&r)[4]) {
    constexpr unsigned kLo = 0x05040100u, kHi = 0x07060302u;     // v_perm_b32 byte selectors
    u32x2w c0, c1, c2, c3;
    c0.x = __builtin_amdgcn_perm(r[1].x, r[0].x, kLo); c0.y = __builtin_amdgcn_perm(r[3].x, r[2].x, kLo);
    c1.x = __builtin_amdgcn_perm(r[1].x, r[0].x, kHi); c1.y = __builtin_amdgcn_perm(r[3].x, r[2].x, kHi);
    c2.x = __builtin_amdgcn_perm(r[1].y, r[0].y, kLo); c2.y = __builtin_amdgcn_perm(r[3].y, r[2].y, kLo);
    c3.x = __builtin_amdgcn_perm(r[1].y, r[0].y, kHi); c3.y = __builtin_amdgcn_perm(r[3].y, r[2].y, kHi);
    T* o = base + (4 * q) * kWH_ROW + 4 * pg;
    *reinterpret_cast<u32x2w*>(o) = c0;
    *reinterpret_cast<u32x2w*>(o + kWH_ROW) = c1;
    *reinterpret_cast<u32x2w*>(o + 2 * kWH_ROW) = c2;
    *reinterpret_cast<u32x2w*>(o + 3 * kWH_ROW) = c3;
  };
  auto store_lds = [&]() {
#pragma unroll
    for (int hh = 0; hh < kWH_NH; ++hh) {
      if (do_bias) {
        // bias gradient = column sums of dy, summed where the prefetched registers are consumed
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 v = raw_f4(rd[hh][j], (T*)nullptr);
          bsum.x += v.x; bsum.y += v.y; bsum.z += v.z; bsum.w += v.w;
        }
      }
      tr_store(dS + hh * 64, rd[hh]);
      tr_store(xS + hh * 64, rx[hh]);
      if constexpr (MODE == 2) tr_store(oS + hh * 64, ro[hh]);
    }
    if constexpr (MODE != 1) {
      if (pg == 15) {
        // halo dword of a row: its high half = pixel k0 - 1 (the low half is unused)
        auto put = [&](T* img, const u32x2w& v) {
          unsigned short* h = reinterpret_cast<unsigned short*>(img) + (4 * q) * kWH_ROW + kWH_PK + 1;
          h[0] = (unsigned short)(v.x & 0xFFFFu);
          h[kWH_ROW] = (unsigned short)(v.x >> 16);
          h[2 * kWH_ROW] = (unsigned short)(v.y & 0xFFFFu);
          h[3 * kWH_ROW] = (unsigned short)(v.y >> 16);
        };
        if constexpr (MODE == 0) {
          put(dS, sv_d);
          put(xS, sv_x);
        } else {
          put(oS, sv_o);
        }
      }
      sv_d = rd[kWH_NH - 1][3]; sv_x = rx[kWH_NH - 1][3];
      if constexpr (MODE == 2) sv_o = ro[kWH_NH - 1][3];
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

#if EMSA_WH_DBG
  long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long dbg_prev = (long long)__builtin_readcyclecounter();
  const long long dbg_t0 = dbg_prev;
#endif
  if (s_begin < s_end) {
    load_front(s_begin);
    load_regs(s_begin);
    store_lds();
  }
  __syncthreads();
  WH_MARK(5);

  typedef typename std::conditional<std::is_same<T, emsa_f16>::value, _Float16, __bf16>::type E;
  typedef E ev8 __attribute__((ext_vector_type(8)));
  auto mma = [&](f32x16& c_, const ev8& av, const u32x4h& bv) {
    if constexpr (std::is_same<T, emsa_f16>::value)
      c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(ev8, bv), c_, 0, 0, 0);
    else
      c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(ev8, bv), c_, 0, 0, 0);
  };
  for (int s = s_begin; s < s_end; ++s) {
    const bool has_next = s + 1 < s_end;
    if (has_next) load_regs(s + 1);
    WH_MARK(0);
    const T* drow0 = dS + (wco * 32 + l31) * kWH_ROW;
    const T* drow = drow0 + 8 * lh;
    const T* xrow = xS + (wci * 32 + l31) * kWH_ROW + 8 * lh;
    // the image whose shift by one pixel is read: x itself (MODE 0) or the odd pixels (MODE 2)
    const T* srow0 = (MODE == 2 ? oS : xS) + (wci * 32 + l31) * kWH_ROW;
    const T* srow = srow0 + 8 * lh;
    // a fragment = two ds_read_b64 (2 LDS cycles each, 64-bank rule).  The kernel is compiled
    // without the load-store optimizer (target attribute): it would merge them into ds_read2_b64
    // (8 cycles, 32-bank rule) -- and the eight ds_write_b64 of the transposes into ds_write2_b64
    // (and the second half through an offset the compiler cannot see through: the IR vectorizer
    //  would turn the pair into one 16-byte load of 8-byte alignment, which is selected as
    //  ds_read2_b64 again -- tools/check_wgrad16_isa.py)
    int h4 = 4;
    asm volatile("" : "+v"(h4));
    auto ld8 = [&](const T* row, int e) {
      const u32x2w a = *reinterpret_cast<const u32x2w*>(row + e);
      const u32x2w b = *reinterpret_cast<const u32x2w*>(row + e + h4);
      u32x4h v; v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y;
      return v;
    };
    // eight pixels shifted by one: v' = pixels [p - 1, p + 6] of v = [p, p + 7].  The pixel in front
    // lives in the partner lane (same row, the other 8-pixel half: lane ^ 32) -- of this K block
    // (lh = 1) or of the previous one (lh = 0) -- or, at the step's start, in the row's halo dword.
    // v_permlane32_swap exchanges the upper half of its first operand with the lower half of its
    // second: one swap serves both halves of the wave.
    auto shifted = [&](const u32x4h& v, unsigned& prev_w) {
      const auto r = __builtin_amdgcn_permlane32_swap(prev_w, v.w, false, false);
      const unsigned wl = lh ? (unsigned)r[0] : (unsigned)r[1];
      prev_w = v.w;
      u32x4h o;
      o.x = __builtin_amdgcn_alignbit(v.x, wl, 16);  o.y = __builtin_amdgcn_alignbit(v.y, v.x, 16);
      o.z = __builtin_amdgcn_alignbit(v.z, v.y, 16); o.w = __builtin_amdgcn_alignbit(v.w, v.z, 16);
      return o;
    };
    unsigned prev_s = 0, prev_d = 0;
    if constexpr (MODE != 1) prev_s = *reinterpret_cast<const unsigned*>(srow0 + kWH_PK);
    if constexpr (MODE == 0) prev_d = *reinterpret_cast<const unsigned*>(drow0 + kWH_PK);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int k16 = 0; k16 < kWH_PK / 16; ++k16) {
      const u32x4h a = ld8(drow, 16 * k16);
      const u32x4h c = ld8(xrow, 16 * k16);
      const ev8 av = __builtin_bit_cast(ev8, a);
      if constexpr (MODE == 1) {
        mma(acc[0], av, c);
      } else if constexpr (MODE == 2) {
        // taps: x(2b-1) = O(b-1) = O shifted, x(2b) = E, x(2b+1) = O
        const u32x4h o_ = ld8(srow, 16 * k16);
        const u32x4h lft = shifted(o_, prev_s);
        mma(acc[0], av, lft);
        mma(acc[1], av, c);
        mma(acc[2], av, o_);
      } else {
        // left tap: dy(p) x(p - 1).  Right tap: dy(p) x(p + 1) summed as dy(p - 1) x(p) -- over the
        // padded enumeration the same set of products (the virtual zero pixel behind every line
        // closes both ends), and BOTH shifted operands then want the pixel in FRONT of the step,
        // which the previous step's registers hold: no halo loads in the K loop (round 3 loaded
        // x(k0 - 1) and x(k0 + 64) per step in wave 0, and the other waves waited for it at the
        // barrier: 320 of a step's 3500 cycles).  A split's partial sums differ from the x-only
        // form by its first and last products; their total does not.
        const u32x4h lft = shifted(c, prev_s);
        const u32x4h a_sh = shifted(a, prev_d);
        mma(acc[0], av, lft);
        mma(acc[1], av, c);
        mma(acc[2], __builtin_bit_cast(ev8, a_sh), c);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    WH_MARK(1);
    __syncthreads();
    WH_MARK(2);
    if (has_next) store_lds();
    WH_MARK(3);
    __syncthreads();
    WH_MARK(4);
  }

  if (p.ws != nullptr) {
    // deterministic split-K: this workgroup's partial tile [taps][BCO][BCI] goes to the workspace
    float* wt = p.ws + (((size_t)ks * p.R + kr) * p.n_tiles + tile) * (NT_ * BCO * BCI);
#pragma unroll
    for (int t = 0; t < NT_; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        wt[(t * BCO + wco * 32 + row) * BCI + wci * 32 + l31] = acc[t][r];
      }
  } else {
    const int ci = ci0 + wci * 32 + l31;
#pragma unroll
    for (int t = 0; t < NT_; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int co = co0 + wco * 32 + row;
        if (co < p.n_ch && ci < p.k_ch)
          unsafeAtomicAdd(p.dw + ((size_t)(kr * NT_ + t) * p.n_ch + co) * p.k_ch + ci, acc[t][r]);
      }
  }
#if EMSA_WH_DBG
  WH_MARK(6);
  dbg_t[7] = dbg_prev - dbg_t0;
  if (lane == 0 && blockIdx.x < 4096)
    for (int i = 0; i < 8; ++i) g_wh_dbg[(blockIdx.x * 4 + wave) * 8 + i] = dbg_t[i];
#endif
  if (do_bias) {
    float* red = smem;   // [16 pixel groups][BCO]
    __syncthreads();     // (the K loop's LDS reads are done; red overlays dS)
    red[pg * BCO + 4 * q + 0] = bsum.x;
    red[pg * BCO + 4 * q + 1] = bsum.y;
    red[pg * BCO + 4 * q + 2] = bsum.z;
    red[pg * BCO + 4 * q + 3] = bsum.w;
    __syncthreads();
    if (tid < BCO) {
      float a = 0.f;
      for (int r = 0; r < 16; ++r) a += red[r * BCO + tid];
      if (p.ws_bias != nullptr)
        p.ws_bias[((size_t)ks * p.n_co_tiles + co_t) * BCO + tid] = a;
      else if (co0 + tid < p.n_ch)
        unsafeAtomicAdd(p.dbias + co0 + tid, a);
    }
  }
}

// ------------------------------------------------------------------------------------------
// The stride-1 3-tap weight gradient with hardware-transposed operand reads (round 4)
// ------------------------------------------------------------------------------------------
// Same tile (64 co x 64 ci x 3 taps, 4 waves of 32 x 32), same split-K workspace and reduction
// pass as conv_wgrad1d_h_kernel MODE 0, but NOTHING is transposed or shifted by the VALU: the NHWC
// rows go to LDS as they are loaded ([pixel][64 channels], 16-byte chunks: two
// buffer_load_dwordx4 + two ds_write_b128 per thread and tensor instead of four 8-byte loads, 16
// v_perm_b32 and four ds_write_b64), and ds_read_b64_tr_b16 delivers the K-contiguous MFMA
// fragments: a 16-lane group hands in the addresses of a [4 pixels][16 channels] block, lane i
// receives channel i of the four pixels.  A tap is a ROW offset of that read: the left tap reads x
// one pixel row up, the right tap -- summed as dy(p - 1) x(p) -- reads dy one row up; the pixel in
// front of a step sits in one of two front rows of each image, stored there by the quad that
// stored it as pixel 63 of the previous step (two rows: the current step still reads the other).
// Measured no faster: fragments of K block k + 1 read under the MFMAs of block k (+20 VGPRs), two
// steps of loads in flight (176 VGPRs: two workgroups per CU), 1024 workgroups.
// conv_wgrad1d_h_kernel spends ~170 VALU instructions per wave and K step on offsets, register
// transposes and v_alignbit shifts beside its 12 MFMAs; this form ~60.
// Rows are 192 bytes apart (64 channels + 64 bytes of padding): the four pixel rows x 64 bytes of
// a 32-lane half of a transposed read then cover the 64 banks once, and the 8 lanes of a
// ds_write_b128 group (two pixel rows x 64 bytes) the 32 banks of the store rule once.
constexpr int kWT_PK = 64;                          // pixels per K step
constexpr int kWT_RS = 96;                          // LDS row stride (elements) = 192 B
constexpr int kWT_IMG = (kWT_PK + 2) * kWT_RS;      // one image: two front rows + 64 pixel rows
// up to kWgradMultiMax independent weight gradients in ONE launch (grid.y = job): the four convs of
// an NBt1D block (same channel count, same pixel count; ref emsanet/model.py:47-58).  A single launch
// at the /16 and /32 stages is 64-256 output tiles split ~12x along K to fill the chip: every
// workgroup then runs ~10 K steps between a cold prologue and a 48 KB partial-tile epilogue, and the
// launch ends in a half-empty round.  Four jobs share the split budget (768 workgroups in total, not
// per job): a quarter of the splits = a quarter of the partial-tile traffic and of the reduction
// pass, four times the K steps per workgroup, one tail round instead of four.
constexpr int kWgradMultiMax = 4;
struct WgradMulti {
  Wgrad1dArgs j[kWgradMultiMax];
};

// XBN (round 6, emsa_conv_wgrad_inbn_t / emsa_conv_wgrad_multi_inbn_t): a job with in_scale != NULL
// takes x = relu(in * in_scale[c] + in_shift[c]) -- the BatchNorm + ReLU in front of the conv, whose
// forward ran with the same fold in ITS loader (conv_rs.hip INBN) -- formed where the prefetched
// registers go to LDS: per thread 16 channels, scale / shift from a 512-byte LDS table, pixels that
// were loaded out of range (line ends, the virtual pixel of an odd line, image borders) stay zero.
template <typename T, bool XBN = false>
__device__ __forceinline__ void wgrad1d_tr_body(const Wgrad1dArgs& p, const int bid, const int nblk) {
  constexpr int BCO = 64, BCI = 64;
  constexpr uint32_t ES = sizeof(T);
  typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // [66][kWT_RS]: row 2 + e = pixel k0 + e; the pixel in front of the step, k0 - 1, is in row
  // (step & 1): the quad that stores pixel 63 of a step also stores it as the NEXT step's front row,
  // which is not the one the current step reads
  T* const dS = reinterpret_cast<T*>(smem);
  T* const xS = dS + kWT_IMG;
  float* const xbnS = reinterpret_cast<float*>(xS + kWT_IMG);     // XBN: [2: scale, shift][BCI]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wco = wave & 1, wci = wave >> 1;
#if EMSA_W1D_XCD
  const int wg = emsa_xcd_remap(bid, nblk);
#else
  const int wg = bid;
#endif
  const int tile = wg % p.n_tiles, kr = (wg / p.n_tiles) % p.R;
  const int ks = wg / (p.n_tiles * p.R);
  const int dline = kr * p.dl_mul + p.dl_off;
  const int ci_t = tile % p.n_ci_tiles, co_t = tile / p.n_ci_tiles;
  const int co0 = co_t * BCO, ci0 = ci_t * BCI;
  const int s_begin = ks * p.steps_per_split;
  const int s_end = min(s_begin + p.steps_per_split, p.steps_total);
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes);
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(p.dout, p.dout_bytes);
  // loader unit: quad tid >> 2 owns pixel (tid >> 2) of the step, its four lanes the 16-byte
  // chunks (lane & 3) and 4 + (lane & 3) of the pixel's 128-byte row -- every lane decomposes its
  // own pixel, nothing crosses lanes
  const int lp = tid >> 2, ch0 = (lane & 3) * 8;
  const bool do_bias = p.dbias != nullptr && ci_t == 0 && kr == 0;
  uint32_t dadd[2], xadd[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    dadd[h] = co0 + ch0 + 32 * h < p.n_ch ? (uint32_t)(co0 + ch0 + 32 * h) * ES : kOOB;
    xadd[h] = ci0 + ch0 + 32 * h < p.k_ch ? (uint32_t)(ci0 + ch0 + 32 * h) * ES : kOOB;
  }
  auto px_off = [&](int k, uint32_t& od, uint32_t& ox) {
    const bool valid = k >= 0 && k < p.M;
    const uint32_t ku = valid ? (uint32_t)k : 0u;
    const uint32_t img = fast_div(ku, p.div_al);
    const uint32_t line = fast_div(ku, p.div_l);                 // = img * A + a
    const int b_ = (int)(ku - __umul24(line, (uint32_t)p.L));
    const int a_ = (int)(line - __umul24(img, (uint32_t)p.A));
    const int a2 = a_ * p.x_mul_a + dline;
    const bool in_line = valid && b_ < p.Lr;                     // the virtual pixel reads zero
    od = in_line ? (__umul24(img, (uint32_t)p.dy_simg) + __umul24((uint32_t)a_, (uint32_t)p.dy_sa) +
                    __umul24((uint32_t)b_, (uint32_t)p.dy_sb)) * ES : kOOB;
    ox = (in_line && a2 >= 0 && a2 < p.in_Ax)
        ? (__umul24(img, (uint32_t)p.in_simg) + __umul24((uint32_t)a2, (uint32_t)p.in_sa1) +
           __umul24((uint32_t)b_, (uint32_t)p.in_sb)) * ES : kOOB;
  };
  const bool xbn = XBN && p.in_scale != nullptr;      // (uniform per workgroup: one job of a launch)
  if constexpr (XBN) {
    if (xbn && tid < 2 * BCI) {
      const int ch = ci0 + (tid & (BCI - 1));
      const float* src = tid < BCI ? p.in_scale : p.in_shift;
      xbnS[tid] = ch < p.k_ch ? src[ch] : 0.f;
    }
  }
  u32x4h rd[2], rx[2];
  [[maybe_unused]] bool x_real = false;               // the prefetched x pixel is a real pixel
  auto load_px = [&](int k, u32x4h (&d)[2], u32x4h (&x)[2]) {
    uint32_t od, ox;
    px_off(k, od, ox);
    if constexpr (XBN) x_real = ox != kOOB;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      d[h] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)(od | dadd[h]) >= 0 ? (int)(od + dadd[h]) : (int)kOOB, 0, 0);
      x[h] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(ox | xadd[h]) >= 0 ? (int)(ox + xadd[h]) : (int)kOOB, 0, 0);
    }
  };
  // bias gradient (column sums of dy): 16 running sums per thread.  (Kept in LDS instead -- to
  // free registers for a second set of loads in flight -- they cost 3 us per launch as read-add-
  // write of the thread's own slots, and 190 us as ds_add_f32: ~64 cycles per wave instruction.)
  float bsum[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[h][e] = 0.f;
  // st: the step being stored (its pixel 63 becomes the front row of step st + 1)
  [[maybe_unused]] auto xbn_apply = [&](u32x4h (&x)[2]) {
    if constexpr (XBN) {
      if (xbn) {
        typedef typename std::conditional<std::is_same<T, emsa_f16>::value, _Float16, __bf16>::type EX;
        typedef EX ex8 __attribute__((ext_vector_type(8)));
        typedef float fx8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float4 s0 = emsa_ld4(xbnS + ch0 + 32 * h), s1 = emsa_ld4(xbnS + ch0 + 32 * h + 4);
          const float4 t0 = emsa_ld4(xbnS + BCI + ch0 + 32 * h), t1 = emsa_ld4(xbnS + BCI + ch0 + 32 * h + 4);
          const fx8 v = __builtin_convertvector(__builtin_bit_cast(ex8, x[h]), fx8);
          const bool ok = x_real && xadd[h] != kOOB;
          fx8 a;
          a[0] = ok ? fmaxf(__builtin_fmaf(v[0], s0.x, t0.x), 0.f) : 0.f;
          a[1] = ok ? fmaxf(__builtin_fmaf(v[1], s0.y, t0.y), 0.f) : 0.f;
          a[2] = ok ? fmaxf(__builtin_fmaf(v[2], s0.z, t0.z), 0.f) : 0.f;
          a[3] = ok ? fmaxf(__builtin_fmaf(v[3], s0.w, t0.w), 0.f) : 0.f;
          a[4] = ok ? fmaxf(__builtin_fmaf(v[4], s1.x, t1.x), 0.f) : 0.f;
          a[5] = ok ? fmaxf(__builtin_fmaf(v[5], s1.y, t1.y), 0.f) : 0.f;
          a[6] = ok ? fmaxf(__builtin_fmaf(v[6], s1.z, t1.z), 0.f) : 0.f;
          a[7] = ok ? fmaxf(__builtin_fmaf(v[7], s1.w, t1.w), 0.f) : 0.f;
          x[h] = __builtin_bit_cast(u32x4h, __builtin_convertvector(a, ex8));
        }
      }
    }
  };
  auto store_lds = [&](int st, const u32x4h (&rd)[2], u32x4h (&rx)[2]) {
    xbn_apply(rx);
    if (do_bias) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u32x2w lo, hi;
        lo.x = rd[h].x; lo.y = rd[h].y; hi.x = rd[h].z; hi.y = rd[h].w;
        const float4 v0 = raw_f4(lo, (T*)nullptr), v1 = raw_f4(hi, (T*)nullptr);
        bsum[h][0] += v0.x; bsum[h][1] += v0.y; bsum[h][2] += v0.z; bsum[h][3] += v0.w;
        bsum[h][4] += v1.x; bsum[h][5] += v1.y; bsum[h][6] += v1.z; bsum[h][7] += v1.w;
      }
    }
    // quad 63 stores its pixel twice: lp_row = row 2 + 63 and the front row of the next step
    const int fr = ((st + 1 - s_begin) & 1) * kWT_RS;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      *reinterpret_cast<u32x4h*>(dS + (2 + lp) * kWT_RS + ch0 + 32 * h) = rd[h];
      *reinterpret_cast<u32x4h*>(xS + (2 + lp) * kWT_RS + ch0 + 32 * h) = rx[h];
    }
    if (lp == kWT_PK - 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<u32x4h*>(dS + fr + ch0 + 32 * h) = rd[h];
        *reinterpret_cast<u32x4h*>(xS + fr + ch0 + 32 * h) = rx[h];
      }
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

#if EMSA_WH_DBG
  long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long dbg_prev = (long long)__builtin_readcyclecounter();
  const long long dbg_t0 = dbg_prev;
#endif
  if (s_begin < s_end) {
    // the pixel in front of the split: loaded by every quad, stored (as "pixel 63 of step
    // s_begin - 1") by quad 63 into front row 0
    if constexpr (XBN) __syncthreads();               // the affine table is in LDS
    load_px(s_begin * kWT_PK - 1, rd, rx);
    if constexpr (XBN) xbn_apply(rx);
    if (lp == kWT_PK - 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<u32x4h*>(dS + ch0 + 32 * h) = rd[h];
        *reinterpret_cast<u32x4h*>(xS + ch0 + 32 * h) = rx[h];
      }
    }
    load_px(s_begin * kWT_PK + lp, rd, rx);
    store_lds(s_begin, rd, rx);
  }
  __syncthreads();
  WH_MARK(5);

  typedef typename std::conditional<std::is_same<T, emsa_f16>::value, _Float16, __bf16>::type E;
  typedef E ev8 __attribute__((ext_vector_type(8)));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  auto mma = [&](f32x16& c_, const s16x8& av, const s16x8& bv) {
    if constexpr (std::is_same<T, emsa_f16>::value)
      c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ev8, av), __builtin_bit_cast(ev8, bv), c_, 0, 0, 0);
    else
      c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ev8, av), __builtin_bit_cast(ev8, bv), c_, 0, 0, 0);
  };
  // transposed fragment read: the 16-lane group lane >> 4 covers channel block (lane >> 4) & 1 of
  // the wave's 32 channels and K half lane >> 5; lane i of it hands in pixel row (i >> 2), 8-byte
  // piece (i & 3) and receives channel i of pixels 0..3 (first read) and 4..7 (second)
  const int gi = lane & 15, g16 = lane >> 4;
  const int frag_off = (8 * (g16 >> 1) + (gi >> 2)) * kWT_RS + ((g16 & 1) * 16 + 4 * (gi & 3));
  const T* const dfr = dS + 2 * kWT_RS + frag_off + wco * 32;   // pixel rows start at row 2
  const T* const xfr = xS + 2 * kWT_RS + frag_off + wci * 32;
  // the lanes whose "one row up" of the step's first K block is the front row (K half 0, row 0 of
  // the 4-row block) reach it through an offset that alternates with the step: row (step & 1) is
  // 2 - (step & 1) rows above pixel row 0, i.e. one more row up on even steps
  const bool at_front = (g16 >> 1) == 0 && (gi >> 2) == 0;
  auto tr4 = [&](const T* ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)const_cast<T*>(ptr));
  };
  auto join = [&](const s16x4& lo, const s16x4& hi) {
    s16x8 v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
    v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return v;
  };
  struct Frags { s16x8 a, a_up, c, c_up; };
  auto read_frags = [&](int k16, int up0) {
    // up0: element offset of the first "one row up" read of K block 0 (front row for some lanes)
    Frags f;
    const int r = 16 * k16 * kWT_RS;
    f.a = join(tr4(dfr + r), tr4(dfr + r + 4 * kWT_RS));
    f.c = join(tr4(xfr + r), tr4(xfr + r + 4 * kWT_RS));
    const int u = k16 == 0 ? up0 : r - kWT_RS;
    f.a_up = join(tr4(dfr + u), tr4(dfr + r + 3 * kWT_RS));
    f.c_up = join(tr4(xfr + u), tr4(xfr + r + 3 * kWT_RS));
    return f;
  };

  auto compute = [&](int st) {
    const int up0 = at_front ? -(2 - ((st - s_begin) & 1)) * kWT_RS : -kWT_RS;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int k16 = 0; k16 < kWT_PK / 16; ++k16) {
      const Frags g = read_frags(k16, up0);
      mma(acc[0], g.a, g.c_up);        // dy(p) x(p - 1)
      mma(acc[1], g.a, g.c);           // dy(p) x(p)
      mma(acc[2], g.a_up, g.c);        // dy(p - 1) x(p) = the right tap's products
    }
    __builtin_amdgcn_s_setprio(0);
  };
  for (int s = s_begin; s < s_end; ++s) {
    const bool has_next = s + 1 < s_end;
    if (has_next) load_px((s + 1) * kWT_PK + lp, rd, rx);
    WH_MARK(0);
    compute(s);
    WH_MARK(1);
    __syncthreads();
    WH_MARK(2);
    if (has_next) store_lds(s + 1, rd, rx);
    WH_MARK(3);
    __syncthreads();
    WH_MARK(4);
  }

  // (the epilogue's lane / thread indices are rebuilt from v_mbcnt: carried over from the prologue
  //  they would be live -- and spilled -- across the K loop)
  const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int tid_e = wave * 64 + lane_e;
  const int l31 = lane_e & 31, lh = lane_e >> 5;
  if (p.ws != nullptr) {
    float* wt = p.ws + (((size_t)ks * p.R + kr) * p.n_tiles + tile) * (3 * BCO * BCI);
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        wt[(t * BCO + wco * 32 + row) * BCI + wci * 32 + l31] = acc[t][r];
      }
  } else {
    const int ci = ci0 + wci * 32 + l31;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int co = co0 + wco * 32 + row;
        if (co < p.n_ch && ci < p.k_ch)
          unsafeAtomicAdd(p.dw + ((size_t)(kr * 3 + t) * p.n_ch + co) * p.k_ch + ci, acc[t][r]);
      }
  }
#if EMSA_WH_DBG
  WH_MARK(6);
  dbg_t[7] = dbg_prev - dbg_t0;
  if (lane == 0 && blockIdx.x < 4096)
    for (int i = 0; i < 8; ++i) g_wh_dbg[(blockIdx.x * 4 + wave) * 8 + i] = dbg_t[i];
#endif
  if (do_bias) {
    float* red = smem;   // [64 pixels][BCO]
    __syncthreads();     // (the K loop's LDS reads are done; red overlays dS)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[lp * BCO + ch0 + 32 * h + e] = bsum[h][e];
    __syncthreads();
    if (tid_e < BCO) {
      float a = 0.f;
      for (int r = 0; r < kWT_PK; ++r) a += red[r * BCO + tid_e];
      if (p.ws_bias != nullptr)
        p.ws_bias[((size_t)ks * p.n_co_tiles + co_t) * BCO + tid_e] = a;
      else if (co0 + tid_e < p.n_ch)
        unsafeAtomicAdd(p.dbias + co0 + tid_e, a);
    }
  }
}

template <typename T, bool XBN = false>
__global__ __launch_bounds__(256, 3) void conv_wgrad1d_tr_kernel(const Wgrad1dArgs p) {
  wgrad1d_tr_body<T, XBN>(p, blockIdx.x, gridDim.x);
}
template <typename T, bool XBN = false>
__global__ __launch_bounds__(256, 3) void conv_wgrad1d_tr_multi_kernel(const WgradMulti m) {
  wgrad1d_tr_body<T, XBN>(m.j[blockIdx.y], blockIdx.x, gridDim.x);
}

// second pass of the deterministic split-K: dw[co][ci][t] (OIHW of a 3-tap 1-D conv) =
// sum over splits of ws[split][tile][t][co_l][ci_l]; dbias[co] = sum of ws_bias[split][co].
// workgroup = one (tile, t, co_l) row of 64 ci (16 float4 columns) x 16 split groups; the
// workgroups behind the weight rows reduce the bias.  (A finer 8 x 32 split with 4 loads in
// flight was slower on every layer shape: most split groups idle when there are few splits.)
__device__ __forceinline__ void wgrad1d_reduce_body(
    const float* __restrict__ ws, const float* __restrict__ ws_bias, int splits, int n_tiles,
    int n_ci_tiles, int n_co_tiles, int n_ch, int k_ch, int R, float* __restrict__ dw,
    float* __restrict__ dbias, int taps, const int bid) {
  // taps: 3 (1-D / 3x3 row taps) or 1 (1x1 convs of the 16-bit direct kernel)
  __shared__ float4 red[16][16];
  const int tid = threadIdx.x;
  const int rows = taps * 64, tile_f = taps * 4096;
  const int weight_blocks = n_tiles * R * rows;
  if (bid < weight_blocks) {
    // workgroup = (kernel row kr, tile, row = t*64 + co_l); ws[split][kr][tile][t][co_l][ci_l]
    const int kt = bid / rows, row = bid % rows;
    const int kr = kt / n_tiles, tile = kt % n_tiles;
    const int col = tid & 15, sg = tid >> 4;
    const float* src = ws + (size_t)kt * tile_f + row * 64 + col * 4;
    float4 a = emsa_zero4();
    const size_t sstride = (size_t)R * n_tiles * tile_f;
    // eight independent loads in flight per thread: with one (a plain loop) the few hundred
    // workgroups of a 64-channel layer (768 partial tiles of 48 KB) are latency bound, 47 us
    for (int sp = sg; sp < splits; sp += 128) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = sp + 16 * u < splits ? emsa_ld4(src + (size_t)(sp + 16 * u) * sstride) : emsa_zero4();
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w;
      }
    }
    red[sg][col] = a;
    __syncthreads();
    if (sg == 0) {
#pragma unroll
      for (int k = 1; k < 16; ++k) {
        const float4 v = red[k][col];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      const int t = row >> 6, co = (tile / n_ci_tiles) * 64 + (row & 63);
      const int ci = (tile % n_ci_tiles) * 64 + col * 4;
      if (co < n_ch) {
        // OIHW: [co][ci][kr][t] (R = 1: the 3 taps of a 1-D conv; R = 3: a 3x3 kernel)
        float* o = dw + (((size_t)co * k_ch + ci) * R + kr) * taps + t;
        const int cs = taps * R;
        if (ci + 0 < k_ch) o[0] = a.x;
        if (ci + 1 < k_ch) o[cs] = a.y;
        if (ci + 2 < k_ch) o[2 * cs] = a.z;
        if (ci + 3 < k_ch) o[3 * cs] = a.w;
      }
    }
  } else if (dbias != nullptr) {
    // bias: 16 lanes x float4 = the tile's 64 channels, 16 split groups, 8 loads in flight (a
    // plain loop over the splits was the critical path of the whole pass for 64-channel layers:
    // 192 dependent round trips)
    const int co_t = bid - weight_blocks;
    const int col = tid & 15, sg = tid >> 4;
    const float* src = ws_bias + (size_t)co_t * 64 + col * 4;
    const size_t sstride = (size_t)n_co_tiles * 64;
    float4 a = emsa_zero4();
    for (int sp = sg; sp < splits; sp += 128) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = sp + 16 * u < splits ? emsa_ld4(src + (size_t)(sp + 16 * u) * sstride) : emsa_zero4();
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w;
      }
    }
    red[sg][col] = a;
    __syncthreads();
    if (sg == 0) {
#pragma unroll
      for (int k = 1; k < 16; ++k) {
        const float4 v = red[k][col];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      const int co = co_t * 64 + col * 4;
      if (co + 0 < n_ch) dbias[co + 0] = a.x;
      if (co + 1 < n_ch) dbias[co + 1] = a.y;
      if (co + 2 < n_ch) dbias[co + 2] = a.z;
      if (co + 3 < n_ch) dbias[co + 3] = a.w;
    }
  }
}

// The same pass for layers with many tiles and few splits (512 channels: 64 tiles x 12 splits): the
// kernel above runs 12288 workgroups of 12 useful loads each and is bound by their dispatch (16 us
// for the 37.7 MB that take 8 us at 64-256 channels; 3x3 512->512: 31).  Here a thread owns one
// float4 of one (tile, co_l) row and sums all splits of all R x taps weight rows itself, in the
// split order of the kernel above for <= 16 splits (bit-identical): 16 rows per workgroup.
__device__ __forceinline__ void wgrad1d_reduce_rows_body(
    const float* __restrict__ ws, const float* __restrict__ ws_bias, int splits, int n_tiles,
    int n_ci_tiles, int n_co_tiles, int n_ch, int k_ch, int R, float* __restrict__ dw,
    float* __restrict__ dbias, int taps, const int bid) {
  const int tid = threadIdx.x;
  const int tile_f = taps * 4096, nrt = R * taps;
  const int weight_blocks = n_tiles * 4;
  const int col = tid & 15, rw = tid >> 4;
  if (bid < weight_blocks) {
    const int tile = bid >> 2, co_l = (bid & 3) * 16 + rw;
    const size_t sstride = (size_t)R * n_tiles * tile_f;
    const int co = (tile / n_ci_tiles) * 64 + co_l;
    const int ci = (tile % n_ci_tiles) * 64 + col * 4;
    for (int rt = 0; rt < nrt; ++rt) {
      const int kr = rt / taps, t = rt - kr * taps;
      const float* src = ws + (size_t)(kr * n_tiles + tile) * tile_f + (t * 64 + co_l) * 64 + col * 4;
      float4 a = emsa_zero4();
      for (int sp = 0; sp < splits; sp += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = sp + u < splits ? emsa_ld4(src + (size_t)(sp + u) * sstride) : emsa_zero4();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w;
        }
      }
      if (co < n_ch) {
        float* o = dw + ((size_t)co * k_ch + ci) * nrt + rt;     // [co][ci][kr][t], rt = kr * taps + t
        if (ci + 0 < k_ch) o[0] = a.x;
        if (ci + 1 < k_ch) o[nrt] = a.y;
        if (ci + 2 < k_ch) o[2 * nrt] = a.z;
        if (ci + 3 < k_ch) o[3 * nrt] = a.w;
      }
    }
  } else if (dbias != nullptr && rw == 0) {
    const int co_t = bid - weight_blocks;
    const float* src = ws_bias + (size_t)co_t * 64 + col * 4;
    const size_t sstride = (size_t)n_co_tiles * 64;
    float4 a = emsa_zero4();
    for (int sp = 0; sp < splits; ++sp) {
      const float4 v = emsa_ld4(src + (size_t)sp * sstride);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    const int co = co_t * 64 + col * 4;
    if (co + 0 < n_ch) dbias[co + 0] = a.x;
    if (co + 1 < n_ch) dbias[co + 1] = a.y;
    if (co + 2 < n_ch) dbias[co + 2] = a.z;
    if (co + 3 < n_ch) dbias[co + 3] = a.w;
  }
}

__global__ __launch_bounds__(256) void wgrad1d_reduce_kernel(
    const float* __restrict__ ws, const float* __restrict__ ws_bias, int splits, int n_tiles,
    int n_ci_tiles, int n_co_tiles, int n_ch, int k_ch, int R, float* __restrict__ dw,
    float* __restrict__ dbias, int taps) {
  wgrad1d_reduce_body(ws, ws_bias, splits, n_tiles, n_ci_tiles, n_co_tiles, n_ch, k_ch, R, dw, dbias,
                      taps, (int)blockIdx.x);
}
__global__ __launch_bounds__(256) void wgrad1d_reduce_rows_kernel(
    const float* __restrict__ ws, const float* __restrict__ ws_bias, int splits, int n_tiles,
    int n_ci_tiles, int n_co_tiles, int n_ch, int k_ch, int R, float* __restrict__ dw,
    float* __restrict__ dbias, int taps) {
  wgrad1d_reduce_rows_body(ws, ws_bias, splits, n_tiles, n_ci_tiles, n_co_tiles, n_ch, k_ch, R, dw,
                           dbias, taps, (int)blockIdx.x);
}
// the second pass of a multi-job launch: grid.y = job, everything but the four pointers is shared
struct ReduceMulti {
  const float* ws[kWgradMultiMax];
  const float* ws_bias[kWgradMultiMax];
  float* dw[kWgradMultiMax];
  float* dbias[kWgradMultiMax];
  int splits, n_tiles, n_ci_tiles, n_co_tiles, n_ch, k_ch, R, taps;
};
template <bool ROWS>
__global__ __launch_bounds__(256) void wgrad1d_reduce_multi_kernel(const ReduceMulti m) {
  const int j = blockIdx.y;
  if constexpr (ROWS)
    wgrad1d_reduce_rows_body(m.ws[j], m.ws_bias[j], m.splits, m.n_tiles, m.n_ci_tiles, m.n_co_tiles,
                             m.n_ch, m.k_ch, m.R, m.dw[j], m.dbias[j], m.taps, (int)blockIdx.x);
  else
    wgrad1d_reduce_body(m.ws[j], m.ws_bias[j], m.splits, m.n_tiles, m.n_ci_tiles, m.n_co_tiles,
                        m.n_ch, m.k_ch, m.R, m.dw[j], m.dbias[j], m.taps, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
enum ConvTile { TILE_128x128 = 0, TILE_128x64, TILE_64x64, TILE_128x32, TILE_COUNT };

int tile_bm(ConvTile t) { return t == TILE_64x64 ? 64 : 128; }
int tile_bn(ConvTile t) {
  return t == TILE_128x128 ? 128 : t == TILE_128x32 ? 32 : 64;
}

// EMSA_CONV_TILE=0..3 forces a tile configuration (tests / tuning); read on every call
int forced_tile() {
  const char* e = getenv("EMSA_CONV_TILE");
  const int v = (e && *e) ? atoi(e) : -1;
  return (v >= 0 && v < TILE_COUNT) ? v : -1;
}

// Measured on MI355X (tools/conv_bench.py, profiles/r01_b_conv_tiles.md): with the slow fp32 MFMA
// (64 cycles per 32x32x2) operand reuse is not the limiter -- phase overlap is.  The 64x64 tile
// (36 KiB LDS -> 4 workgroups = 16 waves per CU) beats 128x64 / 128x128 on every layer shape of
// the model (c64: 80 vs 60 TF, c128: 94 vs 81, c256/c512: 95 vs 90), so it is the default;
// 128x32 only serves the narrow heads.
ConvTile pick_tile(long M, int n_ch) {
  const int f = forced_tile();
  if (f >= 0) return (ConvTile)f;
  if (n_ch <= 32 && M >= 128 * 256) return TILE_128x32;
  return TILE_64x64;
}

bool geom_mapped(const EmsaConvGeom* g) {
  return g->out_pix_img || g->out_pix_row || g->out_pix_px || g->out_pix_off;
}
bool geom_ok(const EmsaConvGeom* g) {
  if (!g) return false;
  if (g->k_ch <= 0 || g->n_ch <= 0 || (g->k_ch & 3) || (g->in_px_stride & 3)) return false;
  if ((g->in_row_stride & 3) || (g->in_img_stride & 3)) return false;
  if (g->div_h < 1 || g->div_w < 1 || g->kh < 1 || g->kw < 1) return false;
  const long in_elems = (long)g->n_img * g->in_img_stride;
  const long out_elems = (long)g->n_img * g->out_h * g->out_w * (long)g->ld_out;
  // buffer descriptors address < 2 GiB (kOOB = 2^31 must be out of range); outputs int32-indexed
  if (in_elems >= (1L << 29) || out_elems >= (1L << 31)) return false;
  if ((long)g->kh * g->kw * g->n_ch * g->k_ch >= (1L << 29)) return false;
  return true;
}

template <int BM, int BN, int WM, int WN>
int launch_igemm(const ConvArgs& a, hipStream_t st) {
  constexpr size_t lds_main = (size_t)((EMSA_SB & 1) ? 1 : 2) * (BM + BN) * kLD * sizeof(float);
  constexpr size_t lds_epi = (size_t)BM * (BN + 4) * sizeof(float);   // staged epilogue
  constexpr size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, WM, WN>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int grid = a.tiles_m * a.tiles_n;
  constexpr int cls = BN == 128 ? 0 : BN == 32 ? 3 : BM == 128 ? 1 : 2;
  const int ps = prof_begin(cls, algo_flops(a.g), st);
  hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN>), dim3(grid), dim3(WM * WN * 64), lds, st,
                     a);
  prof_end(ps, st);
  return emsa_launch_status();
}

template <int BCO, int BCI, int TT, int WCO, int WCI, int WT, typename T = float>
int launch_wgrad(WgradArgs a, hipStream_t st) {
  constexpr size_t lds = (size_t)((EMSA_SB & 2) ? 1 : 2) * (32 * BCO + TT * 32 * BCI) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad_kernel<BCO, BCI, TT, WCO, WCI, WT, T>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int taps = a.g.kh * a.g.kw;
  a.n_co_tiles = (a.g.n_ch + BCO - 1) / BCO;
  a.n_ci_tiles = (a.g.k_ch + BCI - 1) / BCI;
  a.n_tap_groups = (taps + TT - 1) / TT;
  a.n_tiles = a.n_co_tiles * a.n_ci_tiles * a.n_tap_groups;
  a.steps_total = (a.M + 31) / 32;
  int ksplit = 1024 / a.n_tiles;
  const int max_split = (a.steps_total + 7) / 8;   // at least 8 steps (256 pixels) per block
  if (ksplit > max_split) ksplit = max_split;
  if (ksplit < 1) ksplit = 1;
  a.steps_per_split = (a.steps_total + ksplit - 1) / ksplit;
  ksplit = (a.steps_total + a.steps_per_split - 1) / a.steps_per_split;
  constexpr int cls = sizeof(T) != 4 ? kProfClassWgradH : TT == 7 ? 4 : TT == 3 ? 7 : BCO == 128 ? 5 : 6;
  const int ps = prof_begin(cls, algo_flops(a.g), st);
  hipLaunchKernelGGL((conv_wgrad_kernel<BCO, BCI, TT, WCO, WCI, WT, T>), dim3(a.n_tiles * ksplit),
                     dim3(256), lds, st, a);
  prof_end(ps, st);
  return emsa_launch_status();
}

}  // namespace

extern "C" int emsa_conv_stats_rows(const EmsaConvGeom* g) {
  if (!geom_ok(g)) return EMSA_E_SHAPE;
  const long M = (long)g->n_img * g->out_h * g->out_w;
  const ConvTile t = pick_tile(M, g->n_ch);
  return (int)((M + tile_bm(t) - 1) / tile_bm(t));
}

extern "C" int emsa_conv_igemm(const EmsaConvGeom* g, const float* in, const float* w, float* out,
                               const float* bias, float* stats, const float* scale,
                               const float* shift, const float* residual, int32_t ld_res,
                               const float* mask_src, int32_t ld_mask, int32_t act,
                               void* stream) {
  if (!geom_ok(g)) return EMSA_E_SHAPE;
  if (!in || !w || !out) return EMSA_E_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return EMSA_E_ARG;
  ConvArgs a;
  a.g = *g;
  a.in = in; a.w = w; a.out = out; a.bias = bias; a.stats = stats;
  a.scale = scale; a.shift = shift; a.residual = residual; a.mask_src = mask_src;
  a.ld_res = ld_res; a.ld_mask = ld_mask; a.act = act;
  a.mapped = geom_mapped(g) ? 1 : 0;
  if (a.mapped && stats) return EMSA_E_ARG;          // (statistics count the rows of one launch)
  const long M = (long)g->n_img * g->out_h * g->out_w;
  a.M = (int)M;
  const ConvTile t = pick_tile(M, g->n_ch);
  a.tiles_m = (int)((M + tile_bm(t) - 1) / tile_bm(t));
  a.tiles_n = (g->n_ch + tile_bn(t) - 1) / tile_bn(t);
  a.kchunks = (g->k_ch + kBK - 1) / kBK;
  a.in_bytes = (uint32_t)((size_t)g->n_img * g->in_img_stride * sizeof(float));
  a.w_bytes = (uint32_t)((size_t)g->kh * g->kw * g->n_ch * g->k_ch * sizeof(float));
  a.div_ohw = make_fastdiv((uint32_t)(g->out_h * g->out_w));
  a.div_ow = make_fastdiv((uint32_t)g->out_w);
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  a.vec_epilogue = (g->n_ch % 4 == 0) && (g->ld_out % 4 == 0) && al16(out) && al16(bias) &&
                   al16(scale) && al16(shift) && (!residual || (ld_res % 4 == 0 && al16(residual))) &&
                   (!mask_src || (ld_mask % 4 == 0 && al16(mask_src)));
  if (getenv("EMSA_SCALAR_EPILOGUE")) a.vec_epilogue = 0;
  hipStream_t st = (hipStream_t)stream;
  switch (t) {
    case TILE_128x128: return launch_igemm<128, 128, 2, 2>(a, st);
    case TILE_128x64: return launch_igemm<128, 64, 4, 2>(a, st);   // 8 waves of 32x32
    case TILE_64x64: return launch_igemm<64, 64, 2, 2>(a, st);
    default: return launch_igemm<128, 32, 4, 1>(a, st);
  }
}

namespace {
// stride-1 3-tap 1-D convolution with "same" padding -> halo kernel conv_wgrad1d_kernel.
// Fills the geometry part of `w` and the split-K plan; false if `g` is not such a convolution.
struct Wgrad1dPlan {
  Wgrad1dArgs w;
  int ksplit;
  bool wino;
  int mode;        // 16-bit direct kernel: 0 = stride-1 3 taps, 1 = one tap (1x1), 2 = stride-2 3 taps
};
bool plan_wgrad1d(const EmsaConvGeom* g, bool dout_aligned, Wgrad1dPlan& pl, size_t esize = sizeof(float),
                  bool direct16 = false) {
  // a 3x3 conv = three row taps, each a 3-tap 1-D weight gradient along W on x shifted by a line
  const bool sq = g->kh == 3 && g->kw == 3 && g->off_w == -1 && g->off_h == -1;
  bool along_w = (g->kh == 1 && g->kw == 3 && g->off_w == -1 && g->off_h == 0) || sq;
  const bool along_h = g->kh == 3 && g->kw == 1 && g->off_h == -1 && g->off_w == 0;
  const bool plain = g->step_h == 1 && g->step_w == 1 && g->div_h == 1 && g->div_w == 1 &&
                     dout_aligned && !getenv("EMSA_WGRAD_GENERIC");
  const bool same = g->mul_h == 1 && g->mul_w == 1 && g->in_h == g->out_h && g->in_w == g->out_w;
  // 16-bit direct kernel only: 1x1 convs of stride 1 / 2 (MODE 1) and the stride-2 3-tap convs
  // of a down-sampling block, strided along their own direction only (MODE 2)
  static const bool modes_on = [] {
    const char* e = getenv("EMSA_WGRAD16_MODES");
    return !(e && e[0] == '0');
  }();
  const bool one = direct16 && modes_on && plain && g->kh == 1 && g->kw == 1 && g->off_h == 0 &&
                   g->off_w == 0 && g->mul_h >= 1 && g->mul_h <= 2 && g->mul_w >= 1 && g->mul_w <= 2 &&
                   (g->out_h - 1) * g->mul_h < g->in_h && (g->out_w - 1) * g->mul_w < g->in_w;
  const bool s2 = direct16 && modes_on && plain && !sq &&
                  ((along_w && g->mul_w == 2 && g->mul_h == 1 && g->in_h == g->out_h) ||
                   (along_h && g->mul_h == 2 && g->mul_w == 1 && g->in_w == g->out_w));
  // the 7x7 / 2 stem over the packed NHWC4 input (7 kernel rows of one 32-wide "tap" each) and
  // its rows-as-channels twin for one input channel (2 rows, x lines 4 apart): single-tap rows
  const bool stem = direct16 && modes_on && g->kw == 1 && (g->kh == 7 || g->kh == 2) &&
                    g->k_ch == 32 && g->mul_h == 2 && g->mul_w == 2 && g->off_w == 0 &&
                    g->step_w == 1 && g->div_h == 1 && g->div_w == 1 && dout_aligned &&
                    !getenv("EMSA_WGRAD_GENERIC");
  pl.mode = (one || stem) ? 1 : (s2 ? 2 : 0);
  if (one || stem) along_w = true;
  if (!(one || s2 || stem || ((along_w || along_h) && plain && same))) return false;
  Wgrad1dArgs& w = pl.w;
  w.n_ch = g->n_ch; w.k_ch = g->k_ch;
  const int H = g->out_h, W = g->out_w;
  w.Lr = along_w ? W : H;
  const int A = along_w ? H : W;
  w.A = A;
  w.R = sq ? 3 : (stem ? g->kh : 1);
  w.in_simg = (int)g->in_img_stride;
  w.dy_simg = H * W * g->ld_out;
  if (along_w) {
    w.in_sa = (int)g->in_row_stride * g->mul_h; w.in_sb = g->in_px_stride * g->mul_w;
    w.dy_sa = W * g->ld_out; w.dy_sb = g->ld_out;
    w.in_sb1 = g->in_px_stride; w.in_Lx = g->in_w;
  } else {
    w.in_sa = g->in_px_stride * g->mul_w; w.in_sb = (int)g->in_row_stride * g->mul_h;
    w.dy_sa = g->ld_out; w.dy_sb = W * g->ld_out;
    w.in_sb1 = (int)g->in_row_stride; w.in_Lx = g->in_h;
  }
  w.taps = pl.mode == 1 ? 1 : 3;
  // x line of (output line a, kernel row kr): see Wgrad1dArgs
  if (along_w) {
    w.x_mul_a = g->mul_h; w.in_Ax = g->in_h; w.in_sa1 = (int)g->in_row_stride;
  } else {
    w.x_mul_a = g->mul_w; w.in_Ax = g->in_w; w.in_sa1 = g->in_px_stride;
  }
  w.dl_mul = stem ? g->step_h : (sq ? 1 : 0);
  w.dl_off = stem ? g->off_h : (sq ? -1 : 0);
  // the loader multiplies (image, line, position) by the strides with 24-bit multiplies
  if (w.in_simg >= (1 << 24) || w.dy_simg >= (1 << 24) || w.in_sa >= (1 << 24) ||
      w.dy_sa >= (1 << 24) || g->n_img >= (1 << 24))
    return false;
  // even line length -> Winograd F(3,2) over pixel pairs (EMSA_WGRAD_WINO=0: direct form); an odd
  // line is enumerated with one virtual zero pixel appended (e.g. the 15-pixel lines at /32)
  static const bool wino_on = [] {
    const char* e = getenv("EMSA_WGRAD_WINO");
    return !(e && e[0] == '0');
  }();
  pl.wino = wino_on && !direct16;
  // direct16 (conv_wgrad1d_h_kernel): every line gets a virtual zero pixel behind it
  w.L = (direct16 || (pl.wino && (w.Lr & 1))) ? w.Lr + 1 : w.Lr;
  w.M = g->n_img * A * w.L;
  w.in_bytes = (uint32_t)((size_t)g->n_img * g->in_img_stride * esize);
  w.dout_bytes = (uint32_t)((size_t)g->n_img * H * W * g->ld_out * esize);
  w.div_al = make_fastdiv((uint32_t)(A * w.L));
  w.div_l = make_fastdiv((uint32_t)w.L);
  w.n_co_tiles = (g->n_ch + 63) / 64;
  w.n_ci_tiles = (g->k_ch + 63) / 64;
  w.n_tiles = w.n_co_tiles * w.n_ci_tiles;
  const int pk = direct16 ? kWH_PK : 32;
  w.steps_total = (w.M + pk - 1) / pk;
  // split-K: one round of resident workgroups for the Winograd variant (3 per CU), two rounds
  // of 3 for the direct one (4 per CU) -- measured optima (EMSA_W1D_BLOCKS: tuning only)
  static const int forced_blocks = [] {
    const char* e = getenv("EMSA_W1D_BLOCKS");
    return e ? atoi(e) : 0;
  }();
  // (direct16: 768 / 1024 / 1536 / 2048 workgroups measured 46 / 49 / 55 / 63 us at C=64: more
  //  splits = more partial-tile traffic for the reduction pass)
  const int target_blocks = forced_blocks > 0 ? forced_blocks : (pl.wino || direct16 ? 768 : 1536);
  int ksplit = target_blocks / (w.n_tiles * w.R);
  const int max_split = (w.steps_total + (direct16 ? 3 : 7)) / (direct16 ? 4 : 8);   // >= 256 pixels
  if (ksplit > max_split) ksplit = max_split;
  if (ksplit < 1) ksplit = 1;
  w.steps_per_split = (w.steps_total + ksplit - 1) / ksplit;
  pl.ksplit = (w.steps_total + w.steps_per_split - 1) / w.steps_per_split;
  return true;
}
bool dout_is_aligned(const EmsaConvGeom* g, const float* dout) {
  return ((g->ld_out & 3) == 0) && ((g->n_ch & 3) == 0) && ((((uintptr_t)dout) & 15) == 0);
}

}  // namespace

// bytes of workspace for the deterministic two-pass weight gradient of `g` (0: not available for
// this geometry -> emsa_conv_wgrad accumulates with atomics into the packed layout)
namespace {
// the 16-bit activations' weight gradient: direct bf16 kernel (default) or, EMSA_WGRAD16=wino, the
// Winograd F(3,2) kernel with bf16 loads (A/B)
bool wgrad16_direct() {
  static const bool v = [] {
    const char* e = getenv("EMSA_WGRAD16");
    return !(e && e[0] == 'w');
  }();
  return v;
}
// EMSA_WGRAD16_TR=0: the stride-1 3-tap weight gradients on conv_wgrad1d_h_kernel (register
// transposes) instead of conv_wgrad1d_tr_kernel (ds_read_b64_tr_b16)
bool wgrad16_tr() {
  static const bool v = [] {
    const char* e = getenv("EMSA_WGRAD16_TR");
    return !(e && e[0] == '0');
  }();
  return v;
}
// geometry admits the direct 16-bit kernel (8-byte accesses: 4 channels; dout_is_aligned checks dy)
bool direct16_geom(const EmsaConvGeom* g) { return wgrad16_direct() && !(g->k_ch & 3); }
int64_t wgrad_ws_bytes(const EmsaConvGeom* g, size_t esize) {
  if (!geom_ok(g)) return 0;
  Wgrad1dPlan pl;
  const bool half = esize != sizeof(float);
  const bool d16 = half && direct16_geom(g);
  if (!plan_wgrad1d(g, dout_is_aligned(g, nullptr), pl, esize, d16)) return 0;
  if (half && !d16 && !pl.wino) return 0;
  return (int64_t)pl.ksplit * ((int64_t)pl.w.n_tiles * pl.w.R * pl.w.taps * 4096 +
                               (int64_t)pl.w.n_co_tiles * 64) * (int64_t)sizeof(float);
}
}  // namespace
#if EMSA_WH_DBG
extern "C" int emsa_wgrad1d_h_dbg_read(long long* host, int n_entries) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wh_dbg), (size_t)n_entries * 8) == hipSuccess ? 0 : -3;
}
#endif
extern "C" int64_t emsa_conv_wgrad_ws_bytes(const EmsaConvGeom* g) {
  return wgrad_ws_bytes(g, sizeof(float));
}
// workspace of emsa_conv_wgrad_t for activations of `dtype` (the split-K plan depends on it)
extern "C" int64_t emsa_conv_wgrad_ws_bytes_t(int32_t dtype, const EmsaConvGeom* g) {
  return wgrad_ws_bytes(g, dtype == EMSA_DT_F32 ? sizeof(float) : 2);
}

namespace {
template <typename T>
int conv_wgrad_impl(const EmsaConvGeom* g, const T* in_t, const T* dout_t, float* dw, float* dbias,
                    float* ws, void* stream, const float* in_scale = nullptr,
                    const float* in_shift = nullptr) {
  constexpr bool kHalf = sizeof(T) != 4;
  // (the argument structs carry byte addresses; the kernels read them through typed loaders)
  const float* in = reinterpret_cast<const float*>(in_t);
  const float* dout = reinterpret_cast<const float*>(dout_t);
  if (!geom_ok(g) || geom_mapped(g)) return EMSA_E_SHAPE;
  if (!in || !dout || !dw) return EMSA_E_ARG;
  if (g->div_h != 1 || g->div_w != 1) return EMSA_E_SHAPE;
  WgradArgs a;
  a.g = *g;
  a.in = in; a.dout = dout; a.dw = dw; a.dbias = dbias;
  a.M = g->n_img * g->out_h * g->out_w;
  a.dout_aligned = dout_is_aligned(g, dout);
  if (kHalf && (!a.dout_aligned || (((uintptr_t)in) & 7))) return EMSA_E_SHAPE;
  a.in_bytes = (uint32_t)((size_t)g->n_img * g->in_img_stride * sizeof(T));
  a.dout_bytes = (uint32_t)((size_t)a.M * g->ld_out * sizeof(T));
  a.div_ohw = make_fastdiv((uint32_t)(g->out_h * g->out_w));
  a.div_ow = make_fastdiv((uint32_t)g->out_w);
  hipStream_t st = (hipStream_t)stream;
  const int taps = g->kh * g->kw;
  Wgrad1dPlan pl;
  const bool direct16 = kHalf && direct16_geom(g);
  if (direct16 && ((((uintptr_t)in) | ((uintptr_t)dout)) & 7)) return EMSA_E_SHAPE;
  const bool one_d = plan_wgrad1d(g, a.dout_aligned != 0, pl, sizeof(T), direct16) &&
                     (!kHalf || direct16 || pl.wino);
  if (ws != nullptr && !one_d) return EMSA_E_SHAPE;   // ws_bytes(g) was 0
  if (one_d) {
    Wgrad1dArgs& w = pl.w;
    w.in = in; w.dout = dout; w.dw = dw; w.dbias = dbias;
    w.ws = ws;
    w.in_scale = in_scale; w.in_shift = in_shift;
    if (in_scale && !kHalf && (!pl.wino || w.R != 1)) return EMSA_E_SHAPE;
    // (16-bit: the loader fold exists in the transposed-read kernel, conv_wgrad1d_tr_kernel<T, true>)
    if (in_scale && kHalf && !(direct16 && pl.mode == 0 && w.R == 1)) return EMSA_E_SHAPE;
    w.ws_bias = ws ? ws + (size_t)pl.ksplit * w.n_tiles * w.R * w.taps * 4096 : nullptr;
    constexpr int BCO = 64, BCI = 64;
    constexpr size_t lds = (size_t)(32 * BCO + 34 * BCI) * sizeof(float);
    // algorithmic bytes: x and dy read once, the fp32 gradient written once
    const double wbytes = (double)g->n_img * g->in_h * g->in_w * g->k_ch * sizeof(T) +
                          (double)a.M * g->n_ch * sizeof(T) + (double)taps * g->n_ch * g->k_ch * 4.0;
    const int ps = prof_begin(kHalf ? kProfClassWgradH : (w.R == 3 ? kProfClassWgrad3x3 : 7),
                              algo_flops(a.g), st, wbytes);
    static const bool bf16 = [] {
      const char* e = getenv("EMSA_BF16_MFMA");
      return e && e[0] == '1';
    }();
    if constexpr (kHalf) {
      if (direct16) {
        const dim3 grid(w.n_tiles * w.R * pl.ksplit);
        if (in_scale && pl.mode != 0) {
          prof_end(ps, st);
          return EMSA_E_SHAPE;
        }
        if (pl.mode == 1)
          hipLaunchKernelGGL((conv_wgrad1d_h_kernel<T, 1>), grid, dim3(256),
                             (size_t)2 * 64 * kWH_ROW * sizeof(T), st, w);
        else if (pl.mode == 2)
          hipLaunchKernelGGL((conv_wgrad1d_h_kernel<T, 2>), grid, dim3(256),
                             (size_t)3 * 64 * kWH_ROW * sizeof(T), st, w);
        else if (wgrad16_tr() && !(g->k_ch & 7) && !(g->n_ch & 7) && !(g->ld_out & 7) &&
                 !(g->in_px_stride & 7) && !(g->in_row_stride & 7) && !(g->in_img_stride & 7) &&
                 !((((uintptr_t)in) | ((uintptr_t)dout)) & 15))
        {
          // (16-byte chunks: every pixel row of both tensors starts on a 16-byte boundary)
          if (in_scale)
            hipLaunchKernelGGL((conv_wgrad1d_tr_kernel<T, true>), grid, dim3(256),
                               (size_t)2 * kWT_IMG * sizeof(T) + 2 * 64 * sizeof(float), st, w);
          else
            hipLaunchKernelGGL((conv_wgrad1d_tr_kernel<T>), grid, dim3(256),
                               (size_t)2 * kWT_IMG * sizeof(T), st, w);
        } else if (in_scale) {
          prof_end(ps, st);
          return EMSA_E_SHAPE;                     // conv_wgrad1d_h_kernel has no loader fold
        } else
          hipLaunchKernelGGL((conv_wgrad1d_h_kernel<T, 0>), grid, dim3(256),
                             (size_t)2 * 64 * kWH_ROW * sizeof(T), st, w);
      } else
      // EMSA_WGRAD16=wino: E / D are formed in fp32 and rounded to bf16 as they enter LDS, one
      // v_mfma_f32_32x32x16_bf16 per Winograd component and K step
      hipLaunchKernelGGL((conv_wgrad1d_wino_kernel<BCO, BCI, true, T>),
                         dim3(w.n_tiles * w.R * pl.ksplit), dim3(256),
                         (size_t)(16 * 4 * (BCO + BCI)) * sizeof(float), st, w);
    } else if (pl.wino && in_scale) {
      // (tested BEFORE the opt-in bf16-MFMA form: that kernel has no loader fold and would
      //  multiply against the raw BatchNorm input -- ADVICE r3; the fold stays exact fp32)
      if constexpr (!kHalf)
        hipLaunchKernelGGL((conv_wgrad1d_wino_kernel<BCO, BCI, false, float, true>),
                           dim3(w.n_tiles * w.R * pl.ksplit), dim3(256),
                           (size_t)(16 * 4 * (BCO + BCI) + 2 * BCI) * sizeof(float), st, w);
    } else if (in_scale) {
      prof_end(ps, st);
      return EMSA_E_SHAPE;                         // direct form (EMSA_WGRAD_WINO=0): no loader fold
    } else if (pl.wino && bf16)
      hipLaunchKernelGGL((conv_wgrad1d_wino_kernel<BCO, BCI, true>),
                         dim3(w.n_tiles * w.R * pl.ksplit), dim3(256),
                         (size_t)(16 * 4 * (BCO + BCI)) * sizeof(float), st, w);
    else if (pl.wino)
      hipLaunchKernelGGL((conv_wgrad1d_wino_kernel<BCO, BCI>), dim3(w.n_tiles * w.R * pl.ksplit),
                         dim3(256), (size_t)(16 * 4 * (BCO + BCI)) * sizeof(float), st, w);
    else
      hipLaunchKernelGGL((conv_wgrad1d_kernel<BCO, BCI>), dim3(w.n_tiles * w.R * pl.ksplit),
                         dim3(256), lds, st, w);
    static const bool rows_on = [] {
      const char* e = getenv("EMSA_WGRAD_REDUCE_ROWS");
      return !(e && e[0] == '0');
    }();
    if (ws != nullptr && rows_on && pl.ksplit <= 16 && w.n_tiles >= 32)
      hipLaunchKernelGGL(wgrad1d_reduce_rows_kernel, dim3(w.n_tiles * 4 + w.n_co_tiles),
                         dim3(256), 0, st, ws, w.ws_bias, pl.ksplit, w.n_tiles, w.n_ci_tiles,
                         w.n_co_tiles, w.n_ch, w.k_ch, w.R, dw, dbias, w.taps);
    else if (ws != nullptr)
      hipLaunchKernelGGL(wgrad1d_reduce_kernel, dim3(w.n_tiles * w.R * w.taps * 64 + w.n_co_tiles),
                         dim3(256), 0, st, ws, w.ws_bias, pl.ksplit, w.n_tiles, w.n_ci_tiles,
                         w.n_co_tiles, w.n_ch, w.k_ch, w.R, dw, dbias, w.taps);
    prof_end(ps, st);
    return emsa_launch_status();
  }
  if (in_scale) return EMSA_E_SHAPE;             // only the 1-D Winograd form folds the input's BN
  if (taps == 7 && g->k_ch <= 32) return launch_wgrad<64, 32, 7, 2, 1, 2, T>(a, st);
  if (taps == 2 && g->k_ch <= 32) return launch_wgrad<64, 32, 2, 2, 1, 2, T>(a, st);   // one-channel stem
  if (taps == 1) {
    // (a 128x128 tile was measured slower for the 1x1 convs: 96 / 82 us vs 66 / 63 us at 128 /
    //  256 channels -- four times the split-K partial-tile volume per workgroup)
    return launch_wgrad<64, 64, 1, 2, 2, 1, T>(a, st);
  }
  return launch_wgrad<64, 64, 3, 2, 2, 1, T>(a, st);
}
}  // namespace

extern "C" int emsa_conv_wgrad(const EmsaConvGeom* g, const float* in, const float* dout,
                               float* dw, float* dbias, float* ws, void* stream) {
  return conv_wgrad_impl<float>(g, in, dout, dw, dbias, ws, stream);
}

// Weight gradient of a conv whose input is a = relu(in * in_scale[c] + in_shift[c]) (the forward
// ran as emsa_conv1d_wino_inbn): the x operand is formed in the loader, `in` is the BatchNorm's
// input.  Stride-1 3-tap 1-D convs with an even or odd line (the Winograd F(3,2) form) only.
extern "C" int emsa_conv_wgrad_inbn(const EmsaConvGeom* g, const float* in, const float* dout,
                                    float* dw, float* dbias, float* ws, const float* in_scale,
                                    const float* in_shift, void* stream) {
  if (!in_scale || !in_shift) return EMSA_E_ARG;
  return conv_wgrad_impl<float>(g, in, dout, dw, dbias, ws, stream, in_scale, in_shift);
}

// weight (+bias) gradient from activations in storage type `dtype` (training: EMSA_DT_BF16); the
// gradients are fp32 (master weights), the workspace contract is that of emsa_conv_wgrad
extern "C" int emsa_conv_wgrad_t(int32_t dtype, const EmsaConvGeom* g, const void* in,
                                 const void* dout, float* dw, float* dbias, float* ws,
                                 void* stream) {
  if (dtype == EMSA_DT_F32)
    return conv_wgrad_impl<float>(g, (const float*)in, (const float*)dout, dw, dbias, ws, stream);
  if (dtype == EMSA_DT_BF16)
    return conv_wgrad_impl<emsa_bf16>(g, (const emsa_bf16*)in, (const emsa_bf16*)dout, dw, dbias,
                                      ws, stream);
  return EMSA_E_ARG;
}

// ---- several weight gradients in one launch (16-bit storage, conv_wgrad1d_tr_kernel) ----------
namespace {
// all jobs: stride-1 3-tap 1-D (or 3x3) convs the transposing-read kernel takes, same channel counts
// and kernel rows; common split count from ONE budget of 768 workgroups
bool plan_wgrad_multi(int n_jobs, const EmsaConvGeom* geoms, Wgrad1dPlan* pls, int& ksplit) {
  if (n_jobs < 2 || n_jobs > kWgradMultiMax || !wgrad16_tr()) return false;
  for (int j = 0; j < n_jobs; ++j) {
    const EmsaConvGeom* g = geoms + j;
    if (!geom_ok(g) || geom_mapped(g) || g->div_h != 1 || g->div_w != 1) return false;
    if (!direct16_geom(g) || !plan_wgrad1d(g, dout_is_aligned(g, nullptr), pls[j], 2, true)) return false;
    if (pls[j].mode != 0) return false;
    if ((g->k_ch & 7) || (g->n_ch & 7) || (g->ld_out & 7) || (g->in_px_stride & 7) ||
        (g->in_row_stride & 7) || (g->in_img_stride & 7))
      return false;
    if (pls[j].w.n_tiles != pls[0].w.n_tiles || pls[j].w.R != pls[0].w.R ||
        pls[j].w.n_ch != pls[0].w.n_ch || pls[j].w.k_ch != pls[0].w.k_ch)
      return false;
    // (reduction lengths may differ -- the 3x1 and 1x3 convs of one block already do: pixels are
    //  enumerated along the conv direction and an odd line is padded by one virtual pixel.  A shorter
    //  job leaves its last split(s) empty: wgrad1d_tr_body guards its prologue and its K loop with
    //  s_begin < s_end, so such a workgroup loads nothing and stores an all-zero partial tile for the
    //  reduction pass -- tests/test_ops16_gpu.py::test_conv16_wgrad_multi runs unequal pixel counts
    //  against the single-job launches)
  }
  const Wgrad1dArgs& w0 = pls[0].w;
  // (EMSA_WGRAD_MULTI_WGS: the shared workgroup budget, tuning runs only)
  static const int budget = [] {
    const char* e = getenv("EMSA_WGRAD_MULTI_WGS");
    const int v = e ? atoi(e) : 0;
    return v >= 64 ? v : 768;
  }();
  int ks = budget / (n_jobs * w0.n_tiles * w0.R);
  for (int j = 0; j < n_jobs; ++j) {
    const int max_split = (pls[j].w.steps_total + 3) / 4;
    if (ks > max_split) ks = max_split;
  }
  if (ks < 1) ks = 1;
  // every job gets the same number of splits (grid.x is shared): the LONGEST job fixes it
  int n_split = 0;
  for (int j = 0; j < n_jobs; ++j) {
    Wgrad1dArgs& w = pls[j].w;
    w.steps_per_split = (w.steps_total + ks - 1) / ks;
    const int need = (w.steps_total + w.steps_per_split - 1) / w.steps_per_split;
    if (need > n_split) n_split = need;
  }
  ksplit = n_split;           // (a job with fewer steps leaves its last split(s) empty: zero tiles)
  for (int j = 0; j < n_jobs; ++j) pls[j].ksplit = n_split;
  return true;
}
int64_t wgrad_multi_job_floats(const Wgrad1dPlan& pl) {
  return (int64_t)pl.ksplit * ((int64_t)pl.w.n_tiles * pl.w.R * pl.w.taps * 4096 + (int64_t)pl.w.n_co_tiles * 64);
}
}  // namespace

// workspace bytes of emsa_conv_wgrad_multi_t for these jobs; 0: this set of jobs has no multi-job
// form (run them one by one through emsa_conv_wgrad_t)
extern "C" int64_t emsa_conv_wgrad_multi_ws_bytes(int32_t dtype, int32_t n_jobs, const EmsaConvGeom* geoms) {
  if (dtype != EMSA_DT_BF16 || !geoms) return 0;
  Wgrad1dPlan pls[kWgradMultiMax];
  int ks = 0;
  if (!plan_wgrad_multi(n_jobs, geoms, pls, ks)) return 0;
  int64_t fl = 0;
  for (int j = 0; j < n_jobs; ++j) fl += wgrad_multi_job_floats(pls[j]);
  return fl * (int64_t)sizeof(float);
}

// n_jobs (2 .. 4) weight (+ bias) gradients of convs with the same channel counts in one launch +
// one reduction launch: in / dout / dw / dbias are arrays of n_jobs pointers (dbias[j] may be NULL),
// geoms an array of n_jobs geometries, ws = emsa_conv_wgrad_multi_ws_bytes bytes.  Results are
// written in the OIHW parameter layout like emsa_conv_wgrad_t's two-pass form (deterministic).
namespace {
int conv_wgrad_multi_impl(int32_t dtype, int32_t n_jobs, const EmsaConvGeom* geoms,
                          const void* const* in, const void* const* dout, float* const* dw,
                          float* const* dbias, float* ws, const float* const* in_scale,
                          const float* const* in_shift, void* stream) {
  if (dtype != EMSA_DT_BF16) return EMSA_E_ARG;
  if (!geoms || !in || !dout || !dw || !dbias || !ws) return EMSA_E_ARG;
  Wgrad1dPlan pls[kWgradMultiMax];
  int ks = 0;
  if (!plan_wgrad_multi(n_jobs, geoms, pls, ks)) return EMSA_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  WgradMulti m;
  ReduceMulti r;
  float* wsp = ws;
  double flops = 0.0, bytes = 0.0;
  bool any_xbn = false;
  for (int j = 0; j < kWgradMultiMax; ++j) {
    const int k = j < n_jobs ? j : 0;             // (unused slots repeat job 0; never indexed)
    if (j < n_jobs) {
      const EmsaConvGeom* g = geoms + j;
      if (!in[j] || !dout[j] || !dw[j]) return EMSA_E_ARG;
      if ((((uintptr_t)in[j]) | ((uintptr_t)dout[j])) & 15) return EMSA_E_SHAPE;
      Wgrad1dArgs& w = pls[j].w;
      w.in = reinterpret_cast<const float*>(in[j]);
      w.dout = reinterpret_cast<const float*>(dout[j]);
      w.dw = dw[j];
      w.dbias = dbias[j];
      w.ws = wsp;
      w.ws_bias = wsp + (size_t)ks * w.n_tiles * w.R * w.taps * 4096;
      w.in_scale = in_scale ? in_scale[j] : nullptr;
      w.in_shift = in_shift ? in_shift[j] : nullptr;
      if ((w.in_scale == nullptr) != (w.in_shift == nullptr)) return EMSA_E_ARG;
      any_xbn = any_xbn || w.in_scale != nullptr;
      wsp += wgrad_multi_job_floats(pls[j]);
      flops += algo_flops(*g);
      bytes += (double)g->n_img * g->in_h * g->in_w * g->k_ch * 2.0 +
               (double)g->n_img * g->out_h * g->out_w * g->n_ch * 2.0 + (double)g->kh * g->kw * g->n_ch * g->k_ch * 4.0;
    }
    m.j[j] = pls[k].w;
    r.ws[j] = pls[k].w.ws; r.ws_bias[j] = pls[k].w.ws_bias; r.dw[j] = pls[k].w.dw; r.dbias[j] = pls[k].w.dbias;
  }
  const Wgrad1dArgs& w0 = pls[0].w;
  r.splits = ks; r.n_tiles = w0.n_tiles; r.n_ci_tiles = w0.n_ci_tiles; r.n_co_tiles = w0.n_co_tiles;
  r.n_ch = w0.n_ch; r.k_ch = w0.k_ch; r.R = w0.R; r.taps = w0.taps;
  const int ps = prof_begin(kProfClassWgradH, flops, st, bytes);
  if (any_xbn)
    hipLaunchKernelGGL((conv_wgrad1d_tr_multi_kernel<emsa_bf16, true>), dim3(w0.n_tiles * w0.R * ks, n_jobs),
                       dim3(256), (size_t)2 * kWT_IMG * sizeof(emsa_bf16) + 2 * 64 * sizeof(float), st, m);
  else
    hipLaunchKernelGGL((conv_wgrad1d_tr_multi_kernel<emsa_bf16>), dim3(w0.n_tiles * w0.R * ks, n_jobs),
                       dim3(256), (size_t)2 * kWT_IMG * sizeof(emsa_bf16), st, m);
  if (ks <= 16 && w0.n_tiles >= 32)
    hipLaunchKernelGGL((wgrad1d_reduce_multi_kernel<true>), dim3(w0.n_tiles * 4 + w0.n_co_tiles, n_jobs),
                       dim3(256), 0, st, r);
  else
    hipLaunchKernelGGL((wgrad1d_reduce_multi_kernel<false>),
                       dim3(w0.n_tiles * w0.R * w0.taps * 64 + w0.n_co_tiles, n_jobs), dim3(256), 0, st, r);
  prof_end(ps, st);
  return emsa_launch_status();
}
}  // namespace

extern "C" int emsa_conv_wgrad_multi_t(int32_t dtype, int32_t n_jobs, const EmsaConvGeom* geoms,
                                       const void* const* in, const void* const* dout,
                                       float* const* dw, float* const* dbias, float* ws,
                                       void* stream) {
  return conv_wgrad_multi_impl(dtype, n_jobs, geoms, in, dout, dw, dbias, ws, nullptr, nullptr, stream);
}

// The same launch with the loader fold of emsa_conv_wgrad_inbn_t per job: in_scale[j] / in_shift[j]
// non-NULL -> job j's x operand is relu(in[j] * in_scale[j][c] + in_shift[j][c]) (both arrays have
// n_jobs entries; NULL entries = plain jobs).
extern "C" int emsa_conv_wgrad_multi_inbn_t(int32_t dtype, int32_t n_jobs, const EmsaConvGeom* geoms,
                                            const void* const* in, const void* const* dout,
                                            float* const* dw, float* const* dbias, float* ws,
                                            const float* const* in_scale,
                                            const float* const* in_shift, void* stream) {
  if (!in_scale || !in_shift) return EMSA_E_ARG;
  return conv_wgrad_multi_impl(dtype, n_jobs, geoms, in, dout, dw, dbias, ws, in_scale, in_shift, stream);
}

// 16-bit form of emsa_conv_wgrad_inbn: weight gradient of a stride-1 3-tap 1-D conv whose forward
// ran with its input's BatchNorm + ReLU folded into the loader (emsa_conv1d_rs_inbn_t); `in` is the
// BatchNorm's INPUT in storage type `dtype` (EMSA_DT_BF16).  EMSA_E_SHAPE where the transposed-read
// kernel does not take the geometry (the caller then materialises the BatchNorm output).
extern "C" int emsa_conv_wgrad_inbn_t(int32_t dtype, const EmsaConvGeom* g, const void* in,
                                      const void* dout, float* dw, float* dbias, float* ws,
                                      const float* in_scale, const float* in_shift, void* stream) {
  if (!in_scale || !in_shift) return EMSA_E_ARG;
  if (dtype == EMSA_DT_F32)
    return conv_wgrad_impl<float>(g, (const float*)in, (const float*)dout, dw, dbias, ws, stream,
                                  in_scale, in_shift);
  if (dtype == EMSA_DT_BF16)
    return conv_wgrad_impl<emsa_bf16>(g, (const emsa_bf16*)in, (const emsa_bf16*)dout, dw, dbias,
                                      ws, stream, in_scale, in_shift);
  return EMSA_E_ARG;
}

// ---- profiling C-ABI -------------------------------------------------------------------------
extern "C" int emsa_prof_enable(int32_t every) {
  g_prof_every = every > 0 ? every : 0;
  return EMSA_OK;
}
// The kernels count 2 * pixels * k_ch * n_ch * taps on the channel counts they RUN on; a conv whose
// GEMM is zero-padded (stem: 7x7x3 real taps inside 7 x 32, block-diagonal task heads, channel-
// padded 40 -> 40 / 5 -> 8 heads) declares its real direct-convolution FLOPs for the next launch.
extern "C" int emsa_prof_next_flops(double flops) {
  g_prof_next_flops = g_prof_every > 0 ? flops : 0.0;
  return EMSA_OK;
}
extern "C" int emsa_prof_reset(void) {
  g_prof_used = 0;
  for (int i = 0; i < kProfClasses; ++i) g_prof_seen[i] = 0;
  return EMSA_OK;
}
extern "C" int emsa_prof_seen(int32_t cls) { return (cls >= 0 && cls < kProfClasses) ? g_prof_seen[cls] : 0; }
extern "C" const char* emsa_prof_name(int32_t cls) {
  return (cls >= 0 && cls < kProfClasses) ? kProfNames[cls] : "";
}
// algorithmic HBM bytes summed over the SAMPLED launches of a class (0 when the class does not
// track them): with emsa_prof_read's time this gives the achieved GB/s of an HBM-bound kernel
extern "C" int emsa_prof_read_bytes(int32_t cls, double* total_bytes) {
  if (!total_bytes) return EMSA_E_ARG;
  double b = 0.0;
  for (size_t i = 0; i < g_prof_used; ++i)
    if (g_prof_pool[i].cls == cls) b += g_prof_pool[i].bytes;
  *total_bytes = b;
  return EMSA_OK;
}
// call after the stream has been synchronised; sums over the launches recorded since reset
extern "C" int emsa_prof_read(int32_t cls, double* total_ms, double* total_flops,
                              int32_t* launches) {
  if (!total_ms || !total_flops || !launches) return EMSA_E_ARG;
  double ms = 0.0, fl = 0.0;
  int n = 0;
  for (size_t i = 0; i < g_prof_used; ++i) {
    const ProfSlotImpl& s = g_prof_pool[i];
    if (s.cls != cls) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, s.a, s.b) != hipSuccess) return EMSA_E_LAUNCH;
    ms += t;
    fl += s.flops;
    ++n;
  }
  *total_ms = ms;
  *total_flops = fl;
  *launches = n;
  return EMSA_OK;
}
