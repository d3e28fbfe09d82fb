// Fused NBt1D half-block for small-batch 16-bit inference (round 5).
//
// The NonBottleneck1D block of the encoders / decoders (/root/reference/emsanet/model.py:47-58 composes
// it, args.py:158-164 selects it) is two half-blocks of the same form in eval mode, where BatchNorm is
// a per-channel scale / shift:
//     conv3x1 (+ bias) -> ReLU -> conv1x3 (+ bias) -> scale / shift (-> + residual) -> ReLU
// At batch 1 (BASELINE configs[4]) a launch is its fixed cost: the whole-model hipGraph is a gap-free
// chain of 155 launches of 5-7 us each whatever they compute, so the lever is the
// NUMBER of launches.  This kernel runs one half-block as ONE launch for C = 64 / 128 (the /4 and /8
// stages, where a workgroup can hold all channels of a line): a workgroup owns ONE output row of one
// image,
//   * loads the three input rows r - 1, r, r + 1 (zero rows outside the image) into LDS, with a zero
//     pixel in front of and behind each row: X[3][W + 2][C];
//   * conv3x1: the taps are the three rows, every wave keeps its 32-channel slice of the weights in
//     registers as MFMA B operands (fragment-ordered weight image of emsa_pack_weight_frag_t, the
//     operand conv_rs.hip uses); bias + ReLU, rounded to the storage type exactly where the two-launch
//     path stores its intermediate tensor, written to LDS as Y[W + 2][C] with ZERO pad pixels (the 1x3
//     conv pads the intermediate tensor, not relu(bias));
//   * conv1x3: the taps are pixel shifts of Y; bias, scale / shift, residual, ReLU, store.
// The intermediate never leaves the CU; the input rows are read three times over the launch (from L2:
// the whole map is 1-2.5 MB at batch 1).  Same accumulation order (tap, then 16-channel K step) and the
// same epilogue arithmetic as conv_rs_kernel, so the result is bit-identical to the two launches it
// replaces.  blockIdx.y selects one of up to two tensor sets (the twin modules of DESIGN.md 4.7: rgb |
// depth encoder block, semantic | instance decoder block).
#include <mutex>
#include <set>
#include <utility>

#include "common.h"

namespace {

typedef unsigned int hu32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 hbbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hbf16x8 __attribute__((ext_vector_type(8)));

template <typename T> struct HVec8;
template <> struct HVec8<emsa_bf16> { typedef hbbf16x8 type; };
template <> struct HVec8<emsa_f16> { typedef hbf16x8 type; };

__device__ __forceinline__ f32x16 hmfma(hbbf16x8 a, hbbf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 hmfma(hbf16x8 a, hbf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

struct HalfBlockArgs {
  const void* in[2];
  void* out[2];
  const void* res[2];          // may be NULL
  const void* wfa[2];          // fragment-ordered 16-bit weights of the 3x1 conv
  const void* wfb[2];          // ... of the 1x3 conv
  const float* bias_a[2];
  const float* bias_b[2];
  const float* scale[2];       // may be NULL (no folded BatchNorm)
  const float* shift[2];
  int n_img, H, W;
  int ld_in, ld_out, ld_res;   // pixel strides in elements
  int act;                     // activation behind the 1x3 conv
  int xp;                      // pixels per LDS row of X = 32 * ceil((W + 2) / 32)
};

template <typename T, int C>
__global__ __launch_bounds__(256) void nbt_half_block_kernel(const HalfBlockArgs p) {
  typedef typename HVec8<T>::type V8;
  constexpr int C8 = C / 8;                // 16-byte chunks per pixel
  constexpr int PS = C + 8;                // LDS pixel stride (elements): conflict-free ds_read_b128
  constexpr int KS = C / 16;               // k16 steps per tap
  constexpr int NB = C / 32;               // 32-channel output blocks
  constexpr int WN = NB < 4 ? NB : 4;      // waves along the channels
  constexpr int WM = 4 / WN;               // waves along the pixels
  static_assert(NB == WN, "one 32-channel block per wave column");
  extern __shared__ __attribute__((aligned(16))) unsigned char hb_smem[];
  T* const X = reinterpret_cast<T*>(hb_smem);                 // [3][xp][PS]
  T* const Y = X + (size_t)3 * p.xp * PS;                       // [xp + 2][PS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wn = wave % WN, wm = wave / WN;
  const int set = blockIdx.y;
  const int img = blockIdx.x / p.H, r = blockIdx.x - img * p.H;
  const int W = p.W, XP = p.xp;
  const T* const in = static_cast<const T*>(p.in[set]);
  const V8* const wfa = static_cast<const V8*>(p.wfa[set]);
  const V8* const wfb = static_cast<const V8*>(p.wfb[set]);

  // ---- weights of the 3x1 conv: this wave's 32 output channels, all taps and K steps ------------
  // fragment image [tap][n / 32][k / 16][lane][8]: one coalesced 1 KB load per fragment
  V8 bf[3][KS];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) bf[t][kk] = wfa[((t * NB + wn) * KS + kk) * 64 + lane];

  // ---- input rows r - 1, r, r + 1 -> X (pixel i of a row = image column i - 1) ------------------
  const hu32x4 zero4 = {0u, 0u, 0u, 0u};
  {
    const int per_row = W * C8;
    for (int q = tid; q < 3 * per_row; q += 256) {
      const int t = q / per_row, rem = q - t * per_row;
      const int px = rem / C8, c8 = rem - px * C8;
      const int rr = r - 1 + t;
      hu32x4 v = zero4;
      if (rr >= 0 && rr < p.H)
        v = *reinterpret_cast<const hu32x4*>(in + ((size_t)(img * p.H + rr) * W + px) * p.ld_in + c8 * 8);
      *reinterpret_cast<hu32x4*>(X + ((size_t)t * XP + px + 1) * PS + c8 * 8) = v;
    }
    // zero pad pixels: i = 0 and i in (W, XP) of every row; Y[XP], Y[XP + 1]
    const int npad = XP - W;                      // pixels per row: 1 in front + (XP - W - 1) behind
    for (int q = tid; q < 3 * npad * C8; q += 256) {
      const int t = q / (npad * C8), rem = q - t * (npad * C8);
      const int k = rem / C8, c8 = rem - k * C8;
      const int i = k == 0 ? 0 : W + k;
      *reinterpret_cast<hu32x4*>(X + ((size_t)t * XP + i) * PS + c8 * 8) = zero4;
    }
    for (int q = tid; q < 2 * C8; q += 256)
      *reinterpret_cast<hu32x4*>(Y + (size_t)(XP + q / C8) * PS + (q % C8) * 8) = zero4;
  }
  __syncthreads();

  const int n = wn * 32 + l31;                    // this lane's output channel
  const int MBa = XP / 32;                        // pixel blocks of the intermediate row
  // ---- conv3x1 + bias + ReLU -> Y ----------------------------------------------------------------
  {
    const float ba = p.bias_a[set] ? p.bias_a[set][n] : 0.f;
    for (int mb = wm; mb < MBa; mb += WM) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      const T* const a0 = X + (size_t)(32 * mb + l31) * PS + lh * 8;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
          acc = hmfma(*reinterpret_cast<const V8*>(a0 + (size_t)t * XP * PS + kk * 16), bf[t][kk], acc);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = 32 * mb + (q & 3) + 8 * (q >> 2) + 4 * lh;
        float v = fmaxf(acc[q] + ba, 0.f);
        if (i == 0 || i > W) v = 0.f;             // the 1x3 conv zero-pads the INTERMEDIATE tensor
        Y[(size_t)i * PS + n] = (T)v;
      }
    }
  }
  // ---- weights of the 1x3 conv (the loads fly across the barrier) --------------------------------
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) bf[t][kk] = wfb[((t * NB + wn) * KS + kk) * 64 + lane];
  __syncthreads();

  // ---- conv1x3 + bias, scale / shift, residual, activation -> out --------------------------------
  {
    const float bb = p.bias_b[set] ? p.bias_b[set][n] : 0.f;
    const bool affine = p.scale[set] != nullptr;
    const float sc = affine ? p.scale[set][n] : 1.f, sh = affine ? p.shift[set][n] : 0.f;
    const T* const res = static_cast<const T*>(p.res[set]);
    T* const out = static_cast<T*>(p.out[set]);
    const size_t pix0 = (size_t)(img * p.H + r) * W;
    const int MBb = (W + 31) / 32;
    for (int mb = wm; mb < MBb; mb += WM) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      const T* const a0 = Y + (size_t)(32 * mb + l31) * PS + lh * 8;   // output column j reads Y[j + t]
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
          acc = hmfma(*reinterpret_cast<const V8*>(a0 + (size_t)t * PS + kk * 16), bf[t][kk], acc);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = 32 * mb + (q & 3) + 8 * (q >> 2) + 4 * lh;
        if (j < W) {
          float v = acc[q] + bb;
          if (affine) v = v * sc + sh;
          if (res) v += (float)res[(pix0 + j) * p.ld_res + n];
          if (p.act == EMSA_ACT_RELU) v = fmaxf(v, 0.f);
          out[(pix0 + j) * p.ld_out + n] = (T)v;
        }
      }
    }
  }
}

size_t hb_lds_bytes(int c, int w) {
  const int xp = (w + 2 + 31) / 32 * 32;
  return ((size_t)3 * xp + xp + 2) * (c + 8) * 2;
}

template <typename T>
int hb_launch(const HalfBlockArgs& a, int c, int n_sets, hipStream_t st) {
  const size_t lds = hb_lds_bytes(c, a.W);
  const dim3 grid(a.n_img * a.H, n_sets), block(256);
  auto go = [&](void (*kern)(const HalfBlockArgs)) {
    // more than 64 KB of dynamic LDS has to be asked for, once per kernel and device
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> seen;
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
      std::lock_guard<std::mutex> lk(mu);
      if (seen.insert(std::make_pair(dev, (const void*)kern)).second)
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(kern, grid, block, lds, st, a);
  };
  if (c == 64) go(nbt_half_block_kernel<T, 64>);
  else go(nbt_half_block_kernel<T, 128>);
  return emsa_launch_status();
}

}  // namespace

// 1 when emsa_nbt_half_block_t takes a map of `c` channels and width `w` in storage type `dtype`
extern "C" int emsa_nbt_half_block_supported(int32_t dtype, int32_t c, int32_t w) {
  if (dtype != EMSA_DT_BF16 && dtype != EMSA_DT_F16) return 0;
  if (c != 64 && c != 128) return 0;
  if (w < 1 || hb_lds_bytes(c, w) > 160 * 1024) return 0;
  return 1;
}

// One launch for  out = act((conv1x3(relu(conv3x1(in) + bias_a)) + bias_b) * scale + shift + residual)
// on n_sets (1 or 2) independent tensor sets of the same shape: every pointer argument is an array of
// n_sets pointers (residual[k], scale[k] / shift[k], bias_*[k] may be NULL).  Dense NHWC maps
// (n_img, h, w, c), pixel strides ld_*; wfa / wfb: the fragment-ordered forward weights of the two
// convs (emsa_pack_weight_frag_t / emsa_pack_batch kinds 5, 6).
extern "C" int emsa_nbt_half_block_t(int32_t dtype, int32_t n_sets, int32_t n_img, int32_t h, int32_t w,
                                     int32_t c, const void* const* in, int32_t ld_in,
                                     const void* const* wfa, const float* const* bias_a,
                                     const void* const* wfb, const float* const* bias_b,
                                     const float* const* scale, const float* const* shift,
                                     const void* const* residual, int32_t ld_res, void* const* out,
                                     int32_t ld_out, int32_t act, void* stream) {
  if (!in || !wfa || !wfb || !bias_a || !bias_b || !scale || !shift || !residual || !out)
    return EMSA_E_ARG;
  if (n_sets < 1 || n_sets > 2 || n_img < 1 || h < 1) return EMSA_E_ARG;
  if (!emsa_nbt_half_block_supported(dtype, c, w)) return EMSA_E_SHAPE;
  if ((ld_in & 7) || (ld_out & 7) || ld_in < c || ld_out < c || (long)n_img * h * w * ld_in >= (1L << 31) ||
      (long)n_img * h * w * ld_out >= (1L << 31))
    return EMSA_E_SHAPE;
  HalfBlockArgs a;
  for (int k = 0; k < 2; ++k) {
    const int s = k < n_sets ? k : 0;
    if (!in[s] || !wfa[s] || !wfb[s] || !out[s]) return EMSA_E_ARG;
    if ((scale[s] == nullptr) != (shift[s] == nullptr)) return EMSA_E_ARG;
    if ((((uintptr_t)in[s]) | ((uintptr_t)wfa[s]) | ((uintptr_t)wfb[s])) & 15) return EMSA_E_SHAPE;
    if (residual[s] && ld_res < c) return EMSA_E_SHAPE;
    a.in[k] = in[s]; a.out[k] = out[s]; a.res[k] = residual[s]; a.wfa[k] = wfa[s]; a.wfb[k] = wfb[s];
    a.bias_a[k] = bias_a[s]; a.bias_b[k] = bias_b[s]; a.scale[k] = scale[s]; a.shift[k] = shift[s];
  }
  a.n_img = n_img; a.H = h; a.W = w;
  a.ld_in = ld_in; a.ld_out = ld_out; a.ld_res = ld_res;
  a.act = act;
  a.xp = (w + 2 + 31) / 32 * 32;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EMSA_DT_BF16) return hb_launch<emsa_bf16>(a, c, n_sets, st);
  return hb_launch<emsa_f16>(a, c, n_sets, st);
}
