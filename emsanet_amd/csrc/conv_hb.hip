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

// Everything a workgroup reads from global memory is requested in its first instructions -- both weight
// sets, the residual row, the three input rows -- so that ONE memory round trip is paid, not one per
// phase: a workgroup is alone on its CU (one output row each: 120-240 workgroups per launch), nothing
// else hides latency.  (The first version loaded the 1x3 conv's weights behind the 3x1 conv and the
// residual element by element inside the epilogue: 18.7 us per launch, slower than the two launches
// it replaced.)
template <typename T, int C>
__global__ __launch_bounds__(256) void nbt_half_block_kernel(const HalfBlockArgs p) {
  typedef typename HVec8<T>::type V8;
  typedef float hf32x8 __attribute__((ext_vector_type(8)));
  constexpr int C8 = C / 8;                // 16-byte chunks per pixel
  constexpr int PS = C + 8;                // LDS pixel stride (elements): conflict-free ds_read_b128
  constexpr int KS = C / 16;               // k16 steps per tap
  constexpr int NB = C / 32;               // 32-channel output blocks
  constexpr int WN = NB < 4 ? NB : 4;      // waves along the channels
  constexpr int WM = 4 / WN;               // waves along the pixels
  constexpr int SLD = C + 4;               // fp32 stage row (floats)
  constexpr int XB = 8;                    // input chunks per thread and batch
  constexpr int RCH = 8;                   // residual / output chunks per thread (W * C8 <= 256 * RCH)
  static_assert(NB == WN, "one 32-channel block per wave column");
  extern __shared__ __attribute__((aligned(16))) unsigned char hb_smem[];
  T* const X = reinterpret_cast<T*>(hb_smem);                 // [3][xp][PS]
  T* const Y = X + (size_t)3 * p.xp * PS;                       // [xp + 2][PS]
  float* const stage = reinterpret_cast<float*>(hb_smem);     // [32 * MBb][SLD], overlays X behind conv3x1

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wn = wave % WN, wm = wave / WN;
  const int set = blockIdx.y;
  const int img = blockIdx.x / p.H, r = blockIdx.x - img * p.H;
  const int W = p.W, XP = p.xp;
  const T* const in = static_cast<const T*>(p.in[set]);
  const V8* const wfa = static_cast<const V8*>(p.wfa[set]);
  const V8* const wfb = static_cast<const V8*>(p.wfb[set]);
  const T* const res = static_cast<const T*>(p.res[set]);
  const size_t pix0 = (size_t)(img * p.H + r) * W;
  const hu32x4 zero4 = {0u, 0u, 0u, 0u};

  // ---- all global reads of the workgroup, issued back to back ------------------------------------
  // weights: this wave's 32 output channels of both convs, fragment image [tap][n / 32][k / 16][lane][8]
  V8 bfa[3][KS], bfb[3][KS];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) bfa[t][kk] = wfa[((t * NB + wn) * KS + kk) * 64 + lane];
  // per-channel epilogue operands: this lane's channel of both biases; the 8 channels of this
  // thread's output chunks (q % C8 is the same for all of a thread's chunks: 256 % C8 == 0)
  const int n = wn * 32 + l31;
  const int oc0 = (tid % C8) * 8;
  const bool affine = p.scale[set] != nullptr;
  const float ba = p.bias_a[set] ? p.bias_a[set][n] : 0.f;
  const float bb = p.bias_b[set] ? p.bias_b[set][n] : 0.f;
  float4 sc0 = make_float4(1.f, 1.f, 1.f, 1.f), sc1 = sc0, sh0 = emsa_zero4(), sh1 = sh0;
  if (affine) {
    sc0 = emsa_ld4(p.scale[set] + oc0); sc1 = emsa_ld4(p.scale[set] + oc0 + 4);
    sh0 = emsa_ld4(p.shift[set] + oc0); sh1 = emsa_ld4(p.shift[set] + oc0 + 4);
  }
  // residual row: chunk q = (pixel, 16-byte piece) as the output pass will need it
  hu32x4 rres[RCH];
#pragma unroll
  for (int u = 0; u < RCH; ++u) {
    const int q = tid + 256 * u;
    rres[u] = zero4;
    if (res && q < W * C8)
      rres[u] = *reinterpret_cast<const hu32x4*>(res + (pix0 + q / C8) * p.ld_res + (q % C8) * 8);
  }
  // input rows r - 1, r, r + 1 -> X (pixel i of a row = image column i - 1), XB loads in flight
  {
    const int per_row = W * C8, total = 3 * per_row;
    for (int q0 = tid; q0 < total; q0 += 256 * XB) {
      hu32x4 v[XB];
      int dst[XB];
#pragma unroll
      for (int u = 0; u < XB; ++u) {
        const int q = q0 + 256 * u;
        v[u] = zero4;
        dst[u] = -1;
        if (q < total) {
          const int t = q / per_row, rem = q - t * per_row;
          const int px = rem / C8, c8 = rem - px * C8;
          const int rr = r - 1 + t;
          dst[u] = (t * XP + px + 1) * PS + c8 * 8;
          if (rr >= 0 && rr < p.H)
            v[u] = *reinterpret_cast<const hu32x4*>(in + ((size_t)(img * p.H + rr) * W + px) * p.ld_in + c8 * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < XB; ++u)
        if (dst[u] >= 0) *reinterpret_cast<hu32x4*>(X + dst[u]) = v[u];
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
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) bfb[t][kk] = wfb[((t * NB + wn) * KS + kk) * 64 + lane];
  __syncthreads();

  const int MBa = XP / 32;                        // pixel blocks of the intermediate row
  // ---- conv3x1 + bias + ReLU -> Y ----------------------------------------------------------------
  // (Measured and dropped: the wave's pixel blocks interleaved on separate accumulators -- the
  //  dependent MFMA chain of one block at a time is not what bounds the launch: 13.4-18.6 us instead of
  //  11-15.6 us, the extra accumulators go through AGPR copies at this register budget.)
  {
    for (int mb = wm; mb < MBa; mb += WM) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      const T* const a0 = X + (size_t)(32 * mb + l31) * PS + lh * 8;
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
          acc = hmfma(*reinterpret_cast<const V8*>(a0 + (size_t)t * XP * PS + kk * 16), bfa[t][kk], acc);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = 32 * mb + (q & 3) + 8 * (q >> 2) + 4 * lh;
        float v = fmaxf(acc[q] + ba, 0.f);
        if (i == 0 || i > W) v = 0.f;             // the 1x3 conv zero-pads the INTERMEDIATE tensor
        Y[(size_t)i * PS + n] = (T)v;
      }
    }
  }
  __syncthreads();                                // (Y complete; X is free: the stage overlays it)

  // ---- conv1x3 + bias -> fp32 stage ---------------------------------------------------------------
  const int MBb = (W + 31) / 32;
  {
    for (int mb = wm; mb < MBb; mb += WM) {
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      const T* const a0 = Y + (size_t)(32 * mb + l31) * PS + lh * 8;   // output column j reads Y[j + t]
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
          acc = hmfma(*reinterpret_cast<const V8*>(a0 + (size_t)t * PS + kk * 16), bfb[t][kk], acc);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int j = 32 * mb + (q & 3) + 8 * (q >> 2) + 4 * lh;
        stage[(size_t)j * SLD + n] = acc[q] + bb;
      }
    }
  }
  __syncthreads();

  // ---- output pass: scale / shift, residual, activation; 8 channels = 16 bytes per access --------
  {
    T* const out = static_cast<T*>(p.out[set]);
#pragma unroll
    for (int u = 0; u < RCH; ++u) {
      const int q = tid + 256 * u;
      if (q < W * C8) {
        const int j = q / C8, c0 = oc0;
        const float4 v0 = emsa_ld4(stage + (size_t)j * SLD + c0), v1 = emsa_ld4(stage + (size_t)j * SLD + c0 + 4);
        hf32x8 x = hf32x8{v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        if (affine) {
          x = hf32x8{x[0] * sc0.x + sh0.x, x[1] * sc0.y + sh0.y, x[2] * sc0.z + sh0.z,
                     x[3] * sc0.w + sh0.w, x[4] * sc1.x + sh1.x, x[5] * sc1.y + sh1.y,
                     x[6] * sc1.z + sh1.z, x[7] * sc1.w + sh1.w};
        }
        if (res) x += __builtin_convertvector(__builtin_bit_cast(V8, rres[u]), hf32x8);
        if (p.act == EMSA_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.f);
        }
        const V8 o = __builtin_convertvector(x, V8);
        *reinterpret_cast<hu32x4*>(out + (pix0 + j) * p.ld_out + c0) = __builtin_bit_cast(hu32x4, o);
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
  if ((long)w * (c / 8) > 256 * 8) return 0;                       // output chunks per thread (RCH)
  const long xp = (w + 2 + 31) / 32 * 32;
  if ((long)((w + 31) / 32 * 32) * (c + 4) * 4 > 3 * xp * (c + 8) * 2) return 0;   // stage inside X
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
    if (residual[s] && (ld_res < c || (ld_res & 7) || (((uintptr_t)residual[s]) & 15))) return EMSA_E_SHAPE;
    if ((((uintptr_t)out[s]) & 15) || (scale[s] && ((((uintptr_t)scale[s]) | ((uintptr_t)shift[s])) & 15)))
      return EMSA_E_SHAPE;
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
