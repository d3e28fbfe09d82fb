// Implicit-GEMM convolution with 16-bit operands on the CDNA4 matrix cores
// (v_mfma_f32_32x32x16_bf16 / _f16, fp32 accumulation): the forward convolutions and data gradients
// of BASELINE.json configs[2] (bf16 mixed-precision training) and configs[4] (16-bit inference).
//
//   emsa_conv_igemm_t : out[m][n] = epi( sum_{tap,c} in[gather(m,tap)][c] * w[tap][n][c] )
//
// Same gather geometry, same fused epilogue (bias, BatchNorm statistics partials from the fp32
// accumulators, folded BatchNorm, residual, ReLU, ReLU-backward mask) and the same C-ABI shape as
// emsa_conv_igemm (conv_mfma.hip); what differs is everything the 16x faster matrix pipe changes:
//   * at 2.5 PFLOP/s the 64/128-channel layers are HBM-bound (arithmetic intensity 96 / 192 FLOP/B
//     against a ridge of ~400) and the 256/512-channel ones LDS/issue-bound, never MFMA-bound, so
//     the kernel is built around bytes: activations, weights and LDS tiles are 16-bit, one K step
//     is 64 channels (a 128-byte line per pixel row; 32 for short-K launches, see HK below) and
//     every access is 16 bytes per lane, Winograd is NOT used (it saves matrix instructions, which
//     are free here, and costs transforms + precision);
//   * K steps reach LDS by LDS-DMA (buffer_load ... lds) into two unpadded buffers with the bank
//     swizzle on the source side (PF == 0, default), or through registers into one padded buffer
//     (PF >= 1: rows of 144 bytes = 36 banks); either way one MFMA consumes 8 consecutive k per
//     lane = one conflict-free ds_read_b128, so no K permutation trick is needed;
//   * wave tile 64x32 / 64x64 (2x1 / 2x2 MFMA tiles): per 16-deep k sub-step 3-4 ds_read_b128 feed
//     2-4 MFMAs of 32 cycles, which keeps the LDS pipe below the matrix pipe's issue time;
//   * the epilogue converts to the storage type and stores 16 bytes (8 channels) per lane; the
//     accumulators are staged through LDS one wave-row at a time so that the epilogue's LDS never
//     exceeds the main loop's.
#include <cstdlib>

#include <mutex>
#include <set>
#include <utility>

#include "common.h"
#include "prof.h"

namespace {

// Channels per K step: a template parameter HK of the kernel, 64 (a 128-byte line per pixel row) or
// 32.  With 32 a workgroup's LDS buffers are half the size (more workgroups per CU) and a step is
// half the work between barriers: measured faster where the whole K dimension is <= 6 steps of 64
// (the 64- and 128-channel 1-D convs, the 1x1 convs: c64@/4 57 -> 52 us, 1x1 c128 27 -> 24 us) and
// slower on the long-K shapes (3x3 128->40 147 -> 176 us, c512@/32 31 -> 34 us); +2.3 % on the
// bf16 training step with the per-launch choice below (same box, 623.5 vs 609.2 images/s).
constexpr int kHKMax = 64;

// Build-time probes of where a launch's time goes (tools/jobs/r04b.sh builds one library per value;
// the product build has 0): 1 = no K loop and no loads (output pass only), 2 = no output stores,
// 3 = loads but no MFMAs, 4 = no activation loads, 5 = no weight loads.  Results: DESIGN.md 7.
#ifndef EMSA_CONVH_DBG
#define EMSA_CONVH_DBG 0
#endif

typedef unsigned int hu32x4 __attribute__((ext_vector_type(4)));
typedef float hf32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));
constexpr uint32_t kHOOB = 0x80000000u;

template <typename T> struct Vec8;
template <> struct Vec8<emsa_bf16> { typedef hbf16x8 type; };
template <> struct Vec8<emsa_f16> { typedef hf16x8 type; };

__device__ __forceinline__ f32x16 mfma16(hbf16x8 a, hbf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(hf16x8 a, hf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

struct HFastDiv {
  uint32_t mul, shift, d;
};
inline HFastDiv h_make_fastdiv(uint32_t d) {
  HFastDiv f;
  f.d = d;
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  f.shift = s;
  f.mul = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ uint32_t h_fast_div(uint32_t n, const HFastDiv& f) {
  return (__umulhi(n, f.mul) + n) >> f.shift;
}

struct ConvHArgs {
  EmsaConvGeom g;
  const void* in;
  const void* w;
  void* out;
  const float* bias;
  float* stats;
  const float* scale;
  const float* shift;
  const void* residual;
  const void* mask_src;
  // BatchNorm-backward sums fused into a data gradient (emsa_conv_igemm_bnb_t): mask_src = t (the
  // BatchNorm input), scale / shift = the BatchNorm's affine form (used for the ReLU mask of
  // a = relu(bn(t)) only, NOT applied to the result); the stored result is g = dz * (a > 0) and
  // bnb_out receives per pixel tile  sum g  and  sum g * (t - mean) * invstd
  const float* bnb_mean;
  const float* bnb_invstd;
  float* bnb_out;                   // [2][bnb_rows_alloc][n_ch]
  int bnb_rows_alloc;
  int ld_res, ld_mask, act;
  int M, tiles_m, tiles_n, kchunks;
  int mapped;                       // output pixel map in use (EmsaConvGeom::out_pix_*)
  // tap-split (emsa_conv_igemm_splitk_t): `ksplit` workgroups share one output tile, each over a
  // range of taps, and store raw fp32 accumulators to ws[ksplit][M][n_ch]; conv_h_splitk_finish_kernel
  // sums the slabs in a fixed order and applies the epilogue (batch-1 inference: a 3x3 512 -> 512
  // conv on a 15 x 20 map is 40 tiles of 72 K steps otherwise -- 56 us on an idle GPU)
  float* ws;
  int ksplit;
  uint32_t in_bytes, w_bytes;
  HFastDiv div_ohw, div_ow;
  // twin launch (emsa_conv_igemm_pair_t, PAIR instantiations): the workgroups with blockIdx.y == 1
  // run the SAME geometry on a second set of tensors
  const void* in2;
  const void* w2;
  void* out2;
  const float* bias2;
  const float* scale2;
  const float* shift2;
  const void* residual2;
  float* ws2;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t h_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// per-thread offset (VGPR) + wave-uniform offset in the scalar-offset operand; kHOOB reads zero
__device__ __forceinline__ hu32x4 h_ld16s(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}

struct HGather {
  int mul_h, off_h, step_h, div_h, mul_w, off_w, step_w, div_w, in_h, in_w;
  int row_stride, px_stride;
};
// BYTE offset (2-byte elements) of the gathered pixel, or kHOOB
__device__ __forceinline__ uint32_t h_gather(const HGather& q, int img_off, int bh, int bw, int kh,
                                             int kw) {
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
  return ok ? (uint32_t)(img_off + hn * q.row_stride + wn * q.px_stride) * 2u : kHOOB;
}

// PF = K steps of global loads in flight (register sets).  PF = 2 (a second register set, the loop
// unrolled by two) was built to hide more load latency and measured 10-20 % slower on every layer
// shape (106 / 166 / 253 VGPRs instead of 74 / 98 / 157); kept as an A/B switch (EMSA_CONVH_PF=2).
// BNB: the epilogue with the fused BatchNorm-backward sums (ConvHArgs::bnb_out) as its own
// instantiation (its accumulators and channel vectors spilled in the common kernel).
template <int BM, int BN, int WM, int WN, typename T, int PF, bool BNB = false, int HK = 64,
          bool PAIR = false>
__global__ __launch_bounds__(64 * WM * WN, WM * WN == 8 ? 1 : (BM * BN <= 128 * 64) ? ((PF == 2 || BNB) && BM * BN > 64 * 64 ? 3 : 4) : 2)
void conv_h_kernel(
    const ConvHArgs p_in) {
  // 4 waves (256 threads), or 8 (round 6: the 256 x 128 tile of the dense 3x3 convs -- half the L2
  // operand traffic per FLOP of 128 x 128, one workgroup per CU)
  static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves");
  // PAIR: two independent convs of one geometry in one launch (grid.y = 2)
  ConvHArgs p = p_in;
  if constexpr (PAIR) {
    if (blockIdx.y != 0) {
      p.in = p_in.in2; p.w = p_in.w2; p.out = p_in.out2; p.bias = p_in.bias2;
      p.scale = p_in.scale2; p.shift = p_in.shift2; p.residual = p_in.residual2; p.ws = p_in.ws2;
    }
  }
  static_assert(HK == 64 || HK == 32, "K step");
  constexpr int kHK = HK;               // channels per K step
  constexpr int kHLD = kHK + 8;         // padded LDS row of the register-staged variants (elements)
  constexpr int kHRowLanes = kHK / 8;   // lanes (16 B each) per staged row
  typedef typename Vec8<T>::type V8;
  constexpr int NT = 64 * WM * WN;
  constexpr int kRowsPerPass = NT / kHRowLanes;                   // 32 (4 waves, 64-channel steps)
  static_assert(BM % kRowsPerPass == 0 && BN % kRowsPerPass == 0, "loader mapping");
  constexpr int AR = BM / kRowsPerPass, BR = BN / kRowsPerPass;   // 16-byte loads per thread
  constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* const As = reinterpret_cast<T*>(smem_raw);     // [BM][kHLD]
  T* const Bs = As + BM * kHLD;                     // [BN][kHLD]

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int ksl = p.ksplit > 1 ? (int)(blockIdx.x % p.ksplit) : 0;
  const int wg = p.ksplit > 1 ? (int)(blockIdx.x / p.ksplit) : emsa_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = wg % p.tiles_n, mt = wg / p.tiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  const EmsaConvGeom& g = p.g;
  HGather q;
  q.mul_h = g.mul_h; q.off_h = g.off_h; q.step_h = g.step_h; q.div_h = g.div_h;
  q.mul_w = g.mul_w; q.off_w = g.off_w; q.step_w = g.step_w; q.div_w = g.div_w;
  q.in_h = g.in_h; q.in_w = g.in_w;
  q.row_stride = (int)g.in_row_stride; q.px_stride = g.in_px_stride;

  // ---- loader state ---------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rs_in = h_rsrc(p.in, p.in_bytes);
  const __amdgpu_buffer_rsrc_t rs_w = h_rsrc(p.w, p.w_bytes);
  // PF == 0 (LDS-DMA staging): the LDS tiles are unpadded [row][64 channels] images written
  // linearly by the DMA (wave base + lane x 16 B), so the bank-conflict swizzle sits on the SOURCE
  // side: the lane that writes physical 16-byte chunk c' of row r fetches logical chunk
  // c' ^ (r & 7), and the MFMA operand reads apply the same XOR.  r & 7 == lane >> 3 for every row
  // a lane serves (rows advance by 32 per pass, waves by 8), so the logical chunk is a per-lane
  // constant.
  const int rl = tid / kHRowLanes;
  // (32-channel steps: four chunks per 64-byte row, swizzle term (row >> 2) & 3 -- rows r..r+3 fill
  //  one 256-byte bank row with the same chunk index, the next four rows take the next one)
  const int dma_row = lane / kHRowLanes;           // row within this wave's DMA instruction
  const int dma_swz = kHRowLanes == 8 ? (dma_row & 7) : ((dma_row >> 2) & 3);
  const int c8 = PF == 0 ? (((lane % kHRowLanes) ^ dma_swz) * 8) : (tid % kHRowLanes) * 8;
  int a_bh[AR], a_bw[AR], a_img[AR];
  uint32_t a_off[AR], b_off[BR];
#pragma unroll
  for (int j = 0; j < AR; ++j) {
    const int m = m0 + rl + kRowsPerPass * j;
    if (m < p.M) {
      const int img = (int)h_fast_div((uint32_t)m, p.div_ohw);
      const int rem = m - img * (int)p.div_ohw.d;
      const int oh = (int)h_fast_div((uint32_t)rem, p.div_ow), ow = rem - oh * g.out_w;
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
    b_off[j] = n < g.n_ch ? (uint32_t)(n * g.k_ch + c8) * 2u : kHOOB;
  }
  const int taps = g.kh * g.kw;
  // tap-split: this workgroup's taps [tap0, tap1)
  const int tps = p.ksplit > 1 ? (taps + p.ksplit - 1) / p.ksplit : taps;
  const int tap0 = ksl * tps, tap1 = min(taps, tap0 + tps);
  const int steps = EMSA_CONVH_DBG == 1 ? (p.M < 0 ? 1 : 0) : max(tap1 - tap0, 0) * p.kchunks;
  const uint32_t w_tap_bytes = (uint32_t)g.n_ch * g.k_ch * 2u;
  int tap_n = tap0, kc_n = 0;
  hu32x4 rset[PF > 0 ? PF : 1][AR + BR];
  // kHOOB for the lanes whose channels lie beyond k_ch in the last chunk of a tap
  const uint32_t last_oob = (p.kchunks - 1) * kHK + c8 < g.k_ch ? 0u : kHOOB;

  auto load_regs = [&](hu32x4 (&rr)[AR + BR]) {
    // (tap, channel chunk) are wave-uniform: the gathered pixel offsets change once per tap, the
    // channel advance travels in the scalar offset operand
    if (kc_n == 0) {
      const int kh = tap_n / g.kw, kw = tap_n - kh * g.kw;
#pragma unroll
      for (int j = 0; j < AR; ++j) {
        const uint32_t o = h_gather(q, a_img[j], a_bh[j], a_bw[j], kh, kw);
        a_off[j] = (o & kHOOB) ? kHOOB : o + (uint32_t)c8 * 2u;
      }
    }
    const int k0 = kc_n * kHK;
    const uint32_t sa = (uint32_t)k0 * 2u, sb = (uint32_t)tap_n * w_tap_bytes + (uint32_t)k0 * 2u;
    // partial last channel chunk: lanes beyond k_ch go out of range through a wave-uniform mask
    // (no branch around the loads: a join behind them costs an s_waitcnt on the fresh data)
    const uint32_t pm = k0 + kHK > g.k_ch ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int j = 0; j < AR; ++j) rr[j] = h_ld16s(rs_in, a_off[j] | (last_oob & pm), sa);
#pragma unroll
    for (int j = 0; j < BR; ++j) rr[AR + j] = h_ld16s(rs_w, b_off[j] | (last_oob & pm), sb);
    if (++kc_n == p.kchunks) {
      kc_n = 0;
      ++tap_n;
    }
  };
  auto store_lds = [&](const hu32x4 (&rr)[AR + BR]) {
#pragma unroll
    for (int j = 0; j < AR; ++j)
      *reinterpret_cast<hu32x4*>(As + (rl + kRowsPerPass * j) * kHLD + c8) = rr[j];
#pragma unroll
    for (int j = 0; j < BR; ++j)
      *reinterpret_cast<hu32x4*>(Bs + (rl + kRowsPerPass * j) * kHLD + c8) = rr[AR + j];
  };

  // PF == 0: the step's A and B tiles go straight from global memory into LDS buffer `buf`
  // (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass); out-of-range offsets
  // (zero padding, tails) deliver zeros like the register path
  constexpr int kStageElems = (BM + BN) * kHK;     // elements per LDS buffer, rows of 128 bytes
  auto issue_dma = [&](int buf) {
    if (kc_n == 0) {
      const int kh = tap_n / g.kw, kw = tap_n - kh * g.kw;
#pragma unroll
      for (int j = 0; j < AR; ++j) {
        const uint32_t o = h_gather(q, a_img[j], a_bh[j], a_bw[j], kh, kw);
        a_off[j] = (o & kHOOB) ? kHOOB : o + (uint32_t)c8 * 2u;
      }
    }
    const int k0 = kc_n * kHK;
    const uint32_t sa = (uint32_t)k0 * 2u, sb = (uint32_t)tap_n * w_tap_bytes + (uint32_t)k0 * 2u;
    const uint32_t pm = k0 + kHK > g.k_ch ? 0xFFFFFFFFu : 0u;
    T* const a_dst = As + buf * kStageElems + (wave * (64 / kHRowLanes)) * kHK;     // wave-uniform
    T* const b_dst = a_dst + BM * kHK;
#pragma unroll
    for (int j = 0; j < AR; ++j)
      if (EMSA_CONVH_DBG != 4 || p.M < 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_in, (__attribute__((address_space(3))) void*)(a_dst + kRowsPerPass * j * kHK), 16,
          (int)(a_off[j] | (last_oob & pm)), (int)sa, 0, 0);
#pragma unroll
    for (int j = 0; j < BR; ++j)
      if (EMSA_CONVH_DBG != 5 || p.M < 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          rs_w, (__attribute__((address_space(3))) void*)(b_dst + kRowsPerPass * j * kHK), 16,
          (int)(b_off[j] | (last_oob & pm)), (int)sb, 0, 0);
    if (++kc_n == p.kchunks) {
      kc_n = 0;
      ++tap_n;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // operands of buffer `buf` (PF == 0): row stride 64 elements, logical chunk ks * 2 + lh sits at
  // physical chunk (ks * 2 + lh) ^ (row & 7), row & 7 == l31 & 7
  auto compute_dma = [&](int buf) {
    if (EMSA_CONVH_DBG == 3 && p.M >= 0) return;
    const T* a = As + buf * kStageElems + (wm * TM * 32 + l31) * kHK;
    const T* b = As + buf * kStageElems + BM * kHK + (wn * TN * 32 + l31) * kHK;
    const int sw = kHRowLanes == 8 ? (l31 & 7) : ((l31 >> 2) & 3);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < kHK / 16; ++ks) {
      const int ch = ((ks * 2 + lh) ^ sw) * 8;
      V8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const V8*>(a + i * 32 * kHK + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const V8*>(b + j * 32 * kHK + ch);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(fa[i], fb[j], acc[i][j]);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto compute = [&]() {
    const T* a = As + (wm * TM * 32 + l31) * kHLD + lh * 8;
    const T* b = Bs + (wn * TN * 32 + l31) * kHLD + lh * 8;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < kHK / 16; ++ks) {
      V8 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const V8*>(a + i * 32 * kHLD + ks * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const V8*>(b + j * 32 * kHLD + ks * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma16(fa[i], fb[j], acc[i][j]);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // step s: [issue the loads of step s + PF] -> MFMAs on the LDS tile of step s -> barrier ->
  // LDS <- registers of step s + 1 (loaded PF - 1 iterations ago) -> barrier
  if constexpr (PF == 0) {
    // two LDS buffers: step s + 1 is in flight (DMA) while step s is multiplied; ONE barrier per
    // step: behind it every wave's pieces of step s have landed (each wave waited for its own) and
    // every wave is done reading the other buffer (step s - 1)
    if (EMSA_CONVH_DBG != 1 || p.M < 0) issue_dma(0);
    for (int s = 0; s < steps; ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (s + 1 < steps) issue_dma((s + 1) & 1);
      compute_dma(s & 1);
    }
    __syncthreads();                   // the epilogue reuses the buffers
  } else {
  load_regs(rset[0]);
  if constexpr (PF == 2) {
    if (steps > 1) load_regs(rset[1]);
  }
  store_lds(rset[0]);
  __syncthreads();
  }
  if constexpr (PF == 0) {
  } else if constexpr (PF == 1) {
    for (int s = 0; s < steps; ++s) {
      const bool has_next = s + 1 < steps;
      if (has_next) load_regs(rset[0]);
      compute();
      __syncthreads();                 // every wave is done reading the buffer
      if (has_next) store_lds(rset[0]);
      __syncthreads();
    }
  } else {
    // two register sets, unrolled by two so that each is addressed statically: at the top of
    // iteration s set (s & 1) is free (its step s is in LDS) and set ((s + 1) & 1) is in flight
    for (int s = 0; s < steps; s += 2) {
      if (s + 2 < steps) load_regs(rset[0]);
      compute();
      __syncthreads();
      if (s + 1 < steps) store_lds(rset[1]);
      __syncthreads();
      if (s + 1 >= steps) break;
      if (s + 3 < steps) load_regs(rset[1]);
      compute();
      __syncthreads();
      if (s + 2 < steps) store_lds(rset[0]);
      __syncthreads();
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  float* const smem = reinterpret_cast<float*>(smem_raw);
  const bool want_stats = p.stats != nullptr;
  float bvv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + l31;
    bvv[j] = (n < g.n_ch && p.bias && p.ksplit <= 1) ? p.bias[n] : 0.f;
  }
  if (want_stats) {
    // Per-tile (count, sum, M2 about the tile mean) of the fp32 accumulators (+bias) -> merged with
    // Chan's formula in emsa_bn_finalize.  Deterministic (no atomics).
    float* red = smem;              // [WM][BN]; main-loop LDS reads are all behind a barrier
    float* tmean = smem + WM * BN;  // [BN]
    const int cnt = min(BM, p.M - m0);
    float s1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float a1 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m0 + row < p.M) a1 += acc[i][j][r] + bvv[j];
        }
      s1[j] = a1 + __shfl_xor(a1, 32);
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
    float s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float mu = tmean[(wn * TN + j) * 32 + l31];
      float q2 = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m0 + row < p.M) {
            const float d = acc[i][j][r] + bvv[j] - mu;
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

  // ---- output pass: one wave-row (BM / WM pixel rows) at a time through an fp32 LDS stage -------
  constexpr int SLD = BN + 4;
  constexpr int HR = BM / WM;                      // rows per stage pass
  constexpr int C8 = BN / 8, RPP = NT / C8;        // 8-channel columns, rows per pass of the block
  float* const stage = smem;                       // [HR][SLD] <= LDS of the main loop
  const int col8 = tid % C8, row0 = tid / C8;
  const int n = n0 + col8 * 8;
  const bool nok = n < g.n_ch;
  float4 sc0 = make_float4(1.f, 1.f, 1.f, 1.f), sc1 = sc0, sh0 = emsa_zero4(), sh1 = sh0;
  constexpr bool bnb = BNB;
  if (p.scale && nok && !bnb) {
    sc0 = emsa_ld4(p.scale + n); sc1 = emsa_ld4(p.scale + n + 4);
    sh0 = emsa_ld4(p.shift + n); sh1 = emsa_ld4(p.shift + n + 4);
  }
  hf32x8 bsc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, bsh = bsc, bmu = bsc, b0 = bsc, b1 = bsc;
  if (bnb && nok) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bsc[e] = p.scale[n + e]; bsh[e] = p.shift[n + e]; bmu[e] = p.bnb_mean[n + e];
    }
  }
  const T* const res = reinterpret_cast<const T*>(p.residual);
  const T* const msk = reinterpret_cast<const T*>(p.mask_src);
  T* const outp = reinterpret_cast<T*>(p.out);
#pragma unroll
  for (int h = 0; h < WM; ++h) {
    __syncthreads();                               // statistics / previous pass are done with LDS
    if (wm == h) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;       // within the wave-row
            stage[row * SLD + (wn * TN + j) * 32 + l31] = acc[i][j][r] + bvv[j];
          }
    }
    __syncthreads();
    if (nok) {
#pragma unroll 2
      for (int row = row0; row < HR; row += RPP) {
        const int mrow = m0 + h * HR + row;
        if (mrow >= p.M) break;
        // pixel of the produced tensor: the row index, or through the output map (a phase of a
        // strided data gradient writes every s-th row / column)
        int m = mrow;
        if (p.mapped) {
          const int img = (int)h_fast_div((uint32_t)mrow, p.div_ohw);
          const int rem = mrow - img * (int)p.div_ohw.d;
          const int oh = (int)h_fast_div((uint32_t)rem, p.div_ow), ow = rem - oh * g.out_w;
          m = img * g.out_pix_img + oh * g.out_pix_row + ow * g.out_pix_px + g.out_pix_off;
        }
        const float4 v0 = emsa_ld4(stage + row * SLD + col8 * 8);
        const float4 v1 = emsa_ld4(stage + row * SLD + col8 * 8 + 4);
        if (p.ksplit > 1) {                        // raw partial sums of this tap range
          float* wsp = p.ws + ((size_t)ksl * p.M + m) * g.n_ch + n;
          emsa_st4(wsp, v0);
          emsa_st4(wsp + 4, v1);
          continue;
        }
        hf32x8 v = {v0.x * sc0.x + sh0.x, v0.y * sc0.y + sh0.y, v0.z * sc0.z + sh0.z,
                    v0.w * sc0.w + sh0.w, v1.x * sc1.x + sh1.x, v1.y * sc1.y + sh1.y,
                    v1.z * sc1.z + sh1.z, v1.w * sc1.w + sh1.w};
        if (res) {
          const V8 rr = *reinterpret_cast<const V8*>(res + (size_t)m * p.ld_res + n);
          v += __builtin_convertvector(rr, hf32x8);
        }
        if constexpr (bnb) {
          // msk = t: a = t * scale + shift as bn_act_fwd forms it; g = dz * (a > 0)
          const hf32x8 tt =
              __builtin_convertvector(*reinterpret_cast<const V8*>(msk + (size_t)m * p.ld_mask + n), hf32x8);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = (tt[e] * bsc[e] + bsh[e]) > 0.f ? v[e] : 0.f;
            b0[e] += v[e];
            b1[e] += v[e] * (tt[e] - bmu[e]);
          }
        } else if (msk) {
          const hf32x8 mm =
              __builtin_convertvector(*reinterpret_cast<const V8*>(msk + (size_t)m * p.ld_mask + n), hf32x8);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = mm[e] > 0.f ? v[e] : 0.f;
        }
        if (p.act == EMSA_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (EMSA_CONVH_DBG != 2 || p.M < 0)
          *reinterpret_cast<V8*>(outp + (size_t)m * g.ld_out + n) = __builtin_convertvector(v, V8);
      }
    }
  }
  if constexpr (bnb) {
    // the RPP row lanes' sums -> one value per (pixel tile, channel), fixed order (no atomics)
    __syncthreads();
    float* red0 = smem;                            // [RPP][BN]
    float* red1 = smem + RPP * BN;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red0[row0 * BN + col8 * 8 + e] = nok ? b0[e] : 0.f;
      red1[row0 * BN + col8 * 8 + e] = nok ? b1[e] : 0.f;
    }
    __syncthreads();
    for (int col = tid; col < BN; col += NT) {
      const int nn = n0 + col;
      if (nn < g.n_ch) {
        float a0 = 0.f, a1 = 0.f;
        for (int r = 0; r < RPP; ++r) {
          a0 += red0[r * BN + col];
          a1 += red1[r * BN + col];
        }
        p.bnb_out[((size_t)0 * p.bnb_rows_alloc + mt) * g.n_ch + nn] = a0;
        p.bnb_out[((size_t)1 * p.bnb_rows_alloc + mt) * g.n_ch + nn] = a1 * p.bnb_invstd[nn];
      }
    }
  }
}

// sums the ksplit slabs of a tap-split launch in a fixed order and applies the epilogue of
// emsa_conv_igemm_t (bias, folded BatchNorm, residual, ReLU): one thread per pixel and 8 channels
template <typename T>
__global__ void conv_h_splitk_finish_kernel(const float* __restrict__ ws, int ksplit, long M, int n_ch,
                                            int ld_out, const float* __restrict__ bias,
                                            const float* __restrict__ scale,
                                            const float* __restrict__ shift,
                                            const T* __restrict__ residual, int ld_res, int act,
                                            T* __restrict__ out,
                                            const float* __restrict__ ws2 = nullptr,
                                            const float* __restrict__ bias2 = nullptr,
                                            const float* __restrict__ scale2 = nullptr,
                                            const float* __restrict__ shift2 = nullptr,
                                            const T* __restrict__ residual2 = nullptr,
                                            T* __restrict__ out2 = nullptr) {
  typedef typename Vec8<T>::type V8;
  if (blockIdx.y != 0) {          // second half of a twin launch
    ws = ws2; bias = bias2; scale = scale2; shift = shift2; residual = residual2; out = out2;
  }
  const int c8 = n_ch >> 3;
  const long total = M * c8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long m = i / c8;
    const int n = (int)(i - m * c8) * 8;
    hf32x8 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < ksplit; ++k) {
      const float* q = ws + ((size_t)k * M + m) * n_ch + n;
      const float4 a = emsa_ld4(q), b = emsa_ld4(q + 4);
      v += hf32x8{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float x = v[e] + (bias ? bias[n + e] : 0.f);
      if (scale) x = x * scale[n + e] + shift[n + e];
      v[e] = x;
    }
    if (residual) v += __builtin_convertvector(*reinterpret_cast<const V8*>(residual + m * ld_res + n), hf32x8);
    if (act == EMSA_ACT_RELU) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    *reinterpret_cast<V8*>(out + m * ld_out + n) = __builtin_convertvector(v, V8);
  }
}

enum HTile { HT_128x64 = 0, HT_128x128, HT_64x64, HT_256x128, HT_COUNT };
int ht_bm(HTile t) { return t == HT_64x64 ? 64 : (t == HT_256x128 ? 256 : 128); }
int ht_bn(HTile t) { return (t == HT_128x128 || t == HT_256x128) ? 128 : 64; }

// EMSA_CONVH_TILE=0..3 forces a tile configuration (tests / tuning; 3 = 256 x 128 on eight waves: measured
// slower on every 3x3 shape, profiles/r06_p_conv_h_tile_256x128.txt -- the rule below never picks it)
HTile pick_htile(long M, int n_ch) {
  const char* e = getenv("EMSA_CONVH_TILE");
  const int f = (e && *e) ? atoi(e) : -1;
  if (f >= 0 && f < HT_COUNT) return (HTile)f;
  // measured per layer shape at bs=32 with LDS-DMA staging (tools/conv_bench16.py, forward, us for
  // 128x64 / 128x128 / 64x64): c64@/4 57 / 88 / 60, c128@/8 44 / 40 / 46, c256@/16 33 / 35 / 38,
  // c512@/32 32 / 31 / 38, 3x3 512->512@/32 72 / 71 / 92, 3x3 256->128@/8 136 / 122 / 154,
  // 3x3 128->40@/4 142 / 232 / 170, 1x1 c256@/16 19 / 21 / 21 (data gradients alike): 128x64
  // unless there are >= 128 output channels AND >= ~100 K pixels (128x128: half the operand bytes
  // per FLOP, still >= 600 workgroups); 64x64 only when 128x64 would leave CUs without a tile
  // (batch-1 inference, tiny maps).  In the bf16 training step, same box: 604.6 / 602.2 images/s
  // with this rule vs 582.0 / 585.5 with the one tuned for register staging.
  const long wgs = ((M + 127) / 128) * ((n_ch + 63) / 64);
  if (wgs < 256) return HT_64x64;
  if (n_ch >= 128 && M >= 100000) return HT_128x128;
  return HT_128x64;
}

bool h_geom_ok(const EmsaConvGeom* g) {
  if (!g) return false;
  // 16-byte accesses of 2-byte elements: channel counts and strides in multiples of 8
  if (g->k_ch <= 0 || g->n_ch <= 0 || (g->k_ch & 7) || (g->n_ch & 7) || (g->ld_out & 7)) return false;
  if ((g->in_row_stride & 7) || (g->in_img_stride & 7)) return false;
  if (g->in_px_stride & 7) {
    // a 4-element pixel (the stem's zero-padded NHWC4 image) is fine when every gathered column
    // is even: the 16-byte accesses then start on 8-element boundaries
    const bool even_cols = (g->mul_w % 2 == 0) && (g->off_w % 2 == 0) &&
                           (g->kw == 1 || g->step_w % 2 == 0) && g->div_w == 1;
    if ((g->in_px_stride & 3) || !even_cols) return false;
  }
  if (g->div_h < 1 || g->div_w < 1 || g->kh < 1 || g->kw < 1) return false;
  const long in_elems = (long)g->n_img * g->in_img_stride;
  const long out_elems = (long)g->n_img * g->out_h * g->out_w * (long)g->ld_out;
  if (in_elems >= (1L << 30) || out_elems >= (1L << 31)) return false;   // descriptors < 2 GiB
  if ((long)g->kh * g->kw * g->n_ch * g->k_ch >= (1L << 30)) return false;
  return true;
}

// EMSA_CONVH_PF=0|1|2: how the next K step reaches LDS.  0 (default): LDS-DMA into the second of
// two LDS buffers, one barrier per step, no staging registers -- measured 2-11 % faster than 1 on
// the 3-tap / 3x3 shapes (c256@/16 38.3 -> 35.4 us, 3x3 256->128 136 -> 121 us), equal at C = 64,
// 1-4 % slower on the 1x1 convs (one K step per tap: nothing to overlap), +1.8 % on the bf16
// training step.  1: one register set + one padded LDS buffer (two barriers per step).  2: two
// register sets (10-20 % SLOWER than 1 on every shape: +32 VGPRs cost more occupancy than they hide).
int convh_pf() {              // (read per launch like EMSA_CONVH_TILE: the tests switch it)
  const char* e = getenv("EMSA_CONVH_PF");
  const int x = (e && *e) ? atoi(e) : 0;
  return (x == 1 || x == 2) ? x : 0;
}

// K steps of 64 channels over all taps up to which the 32-channel step is used (LDS-DMA variant)
constexpr int kShortK = 6;

// EMSA_CONVH_SHORTK=0: 64-channel K steps for every launch (A/B)
bool convh_short_k() {
  const char* e = getenv("EMSA_CONVH_SHORTK");
  return !(e && e[0] == '0');
}

// EMSA_CONVH_STEM32=0: the 7-tap stem on 64-channel K steps as before round 5 (A/B)
bool convh_stem32() {
  const char* e = getenv("EMSA_CONVH_STEM32");
  return !(e && e[0] == '0');
}

template <int BM, int BN, int WM, int WN, typename T, int HK = 64>
int launch_h(const ConvHArgs& a_in, hipStream_t st) {
  if constexpr (HK == 64) {
    const int steps64 = a_in.g.kh * a_in.g.kw * ((a_in.g.k_ch + 63) / 64);
    // (k_ch <= 32 -- the 7-tap stem over the packed NHWC4 input: 7 steps of 64 would be half padding)
    if (!a_in.bnb_out && convh_pf() == 0 && (steps64 <= kShortK || (a_in.g.k_ch <= 32 && convh_stem32())) && convh_short_k())
      return launch_h<BM, BN, WM, WN, T, 32>(a_in, st);
  }
  ConvHArgs a = a_in;
  a.kchunks = (a.g.k_ch + HK - 1) / HK;
  constexpr int kHK = HK, kHLD = HK + 8;
  constexpr size_t lds_reg = (size_t)(BM + BN) * kHLD * 2;        // one padded buffer (PF >= 1)
  constexpr size_t lds_dma = (size_t)2 * (BM + BN) * kHK * 2;     // two linear buffers (PF == 0)
  constexpr size_t lds_main = lds_reg > lds_dma ? lds_reg : lds_dma;
  constexpr size_t lds_epi = (size_t)(BM / WM) * (BN + 4) * sizeof(float);
  constexpr size_t lds_stat = (size_t)(WM + 1) * BN * sizeof(float);
  constexpr size_t lds0 = lds_main > lds_epi ? lds_main : lds_epi;
  constexpr size_t lds = lds0 > lds_stat ? lds0 : lds_stat;
  const int grid = a.tiles_m * a.tiles_n * (a.ksplit > 1 ? a.ksplit : 1);
  const double flops = 2.0 * a.g.n_img *
                       ((a.g.div_h > 1 || a.g.div_w > 1) ? (double)a.g.in_h * a.g.in_w
                                                         : (double)a.g.out_h * a.g.out_w) *
                       a.g.k_ch * a.g.n_ch * a.g.kh * a.g.kw;
  // algorithmic bytes: every tensor of the launch once (input, output, weights, fused operands)
  const double px_out = (double)a.M * a.g.n_ch * 2.0;
  const double bytes = (double)a.g.n_img * a.g.in_h * a.g.in_w * a.g.k_ch * 2.0 + px_out +
                       (double)a.g.kh * a.g.kw * a.g.n_ch * a.g.k_ch * 2.0 +
                       (a.residual ? px_out : 0.0) + (a.mask_src ? px_out : 0.0);
  if (a.in2) {
    // twin launch: the LDS-DMA forward kernel only (no statistics / mask / BatchNorm-backward form)
    if (a.bnb_out || a.stats || a.mask_src || convh_pf() != 0) return EMSA_E_SHAPE;
    const int ps2 = emsa_prof_begin(kProfClassConvH, 2.0 * flops, st, 2.0 * bytes);
    hipLaunchKernelGGL((conv_h_kernel<BM, BN, WM, WN, T, 0, false, HK, true>), dim3(grid, 2), dim3(64 * WM * WN),
                       lds, st, a);
    emsa_prof_end(ps2, st);
    return emsa_launch_status();
  }
  if constexpr (lds > 64 * 1024) {
    // (the 256 x 128 tile: 96 KB of LDS-DMA buffers -- more than 64 KB has to be asked for, once per
    //  kernel instantiation and device)
    static std::mutex mu;
    static std::set<std::pair<int, int>> seen;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (seen.insert({dev, HK}).second) {
      (void)hipFuncSetAttribute((const void*)conv_h_kernel<BM, BN, WM, WN, T, 0, false, HK>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)conv_h_kernel<BM, BN, WM, WN, T, 1, false, HK>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
  }
  const int ps = emsa_prof_begin(kProfClassConvH, flops, st, bytes);
  if constexpr (HK == 32) {
    hipLaunchKernelGGL((conv_h_kernel<BM, BN, WM, WN, T, 0, false, 32>), dim3(grid), dim3(64 * WM * WN), lds,
                       st, a);
  } else if (a.bnb_out)
    hipLaunchKernelGGL((conv_h_kernel<BM, BN, WM, WN, T, 1, true>), dim3(grid), dim3(64 * WM * WN), lds, st, a);
  else if (convh_pf() == 0)
    hipLaunchKernelGGL((conv_h_kernel<BM, BN, WM, WN, T, 0>), dim3(grid), dim3(64 * WM * WN), lds, st, a);
  else if (convh_pf() == 2)
    hipLaunchKernelGGL((conv_h_kernel<BM, BN, WM, WN, T, 2>), dim3(grid), dim3(64 * WM * WN), lds, st, a);
  else
    hipLaunchKernelGGL((conv_h_kernel<BM, BN, WM, WN, T, 1>), dim3(grid), dim3(64 * WM * WN), lds, st, a);
  emsa_prof_end(ps, st);
  return emsa_launch_status();
}

template <typename T>
int conv_h_dispatch(const ConvHArgs& a, HTile t, hipStream_t st) {
  switch (t) {
    case HT_128x128: return launch_h<128, 128, 2, 2, T>(a, st);
    case HT_256x128: return launch_h<256, 128, 4, 2, T>(a, st);
    case HT_64x64: return launch_h<64, 64, 2, 2, T>(a, st);
    default: return launch_h<128, 64, 2, 2, T>(a, st);
  }
}

}  // namespace

extern "C" int emsa_conv_stats_rows_t(int32_t dtype, const EmsaConvGeom* g) {
  if (dtype == EMSA_DT_F32) return emsa_conv_stats_rows(g);
  if (!h_geom_ok(g)) return EMSA_E_SHAPE;
  const long M = (long)g->n_img * g->out_h * g->out_w;
  const HTile t = pick_htile(M, g->n_ch);
  return (int)((M + ht_bm(t) - 1) / ht_bm(t));
}

static int conv_igemm_h_impl(int32_t dtype, const EmsaConvGeom* g, const void* in,
                             const void* w, void* out, const float* bias, float* stats,
                             const float* scale, const float* shift, const void* residual,
                             int32_t ld_res, const void* mask_src, int32_t ld_mask,
                             int32_t act, const float* bnb_mean, const float* bnb_invstd,
                             float* bnb_out, int32_t bnb_rows_alloc, void* stream,
                             float* ws = nullptr, int ksplit = 1, const ConvHArgs* twin = nullptr) {
  if (dtype != EMSA_DT_BF16 && dtype != EMSA_DT_F16) return EMSA_E_ARG;
  if (!h_geom_ok(g)) return EMSA_E_SHAPE;
  if (!in || !w || !out) return EMSA_E_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return EMSA_E_ARG;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (!al16(in) || !al16(w) || !al16(out) || !al16(bias) || !al16(scale) || !al16(shift) ||
      (residual && (!al16(residual) || (ld_res & 7))) ||
      (mask_src && (!al16(mask_src) || (ld_mask & 7))))
    return EMSA_E_SHAPE;
  ConvHArgs a;
  a.g = *g;
  a.in = in; a.w = w; a.out = out; a.bias = bias; a.stats = stats;
  a.scale = scale; a.shift = shift; a.residual = residual; a.mask_src = mask_src;
  a.ld_res = ld_res; a.ld_mask = ld_mask; a.act = act;
  a.bnb_mean = bnb_mean; a.bnb_invstd = bnb_invstd; a.bnb_out = bnb_out;
  a.bnb_rows_alloc = bnb_rows_alloc;
  a.ws = ws; a.ksplit = ksplit;
  a.in2 = nullptr; a.w2 = nullptr; a.out2 = nullptr; a.bias2 = nullptr; a.scale2 = nullptr;
  a.shift2 = nullptr; a.residual2 = nullptr; a.ws2 = nullptr;
  if (twin) {
    a.in2 = twin->in2; a.w2 = twin->w2; a.out2 = twin->out2; a.bias2 = twin->bias2;
    a.scale2 = twin->scale2; a.shift2 = twin->shift2; a.residual2 = twin->residual2;
    a.ws2 = twin->ws2;
  }
  a.mapped = (g->out_pix_img || g->out_pix_row || g->out_pix_px || g->out_pix_off) ? 1 : 0;
  if (a.mapped && (stats || bnb_out)) return EMSA_E_ARG;    // (per-tile sums count one launch's rows)
  if (ksplit > 1 && (a.mapped || stats || bnb_out || mask_src || !ws)) return EMSA_E_ARG;
  const long M = (long)g->n_img * g->out_h * g->out_w;
  a.M = (int)M;
  const HTile t = pick_htile(M, g->n_ch);
  a.tiles_m = (int)((M + ht_bm(t) - 1) / ht_bm(t));
  if (bnb_out && (!mask_src || !scale || !bnb_mean || !bnb_invstd || stats ||
                  act != EMSA_ACT_NONE || bnb_rows_alloc < a.tiles_m))
    return EMSA_E_ARG;
  a.tiles_n = (g->n_ch + ht_bn(t) - 1) / ht_bn(t);
  a.kchunks = (g->k_ch + kHKMax - 1) / kHKMax;      // (launch_h sets it for its K step)
  a.in_bytes = (uint32_t)((size_t)g->n_img * g->in_img_stride * 2);
  a.w_bytes = (uint32_t)((size_t)g->kh * g->kw * g->n_ch * g->k_ch * 2);
  a.div_ohw = h_make_fastdiv((uint32_t)(g->out_h * g->out_w));
  a.div_ow = h_make_fastdiv((uint32_t)g->out_w);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EMSA_DT_BF16) return conv_h_dispatch<emsa_bf16>(a, t, st);
  return conv_h_dispatch<emsa_f16>(a, t, st);
}

extern "C" int emsa_conv_igemm_t(int32_t dtype, const EmsaConvGeom* g, const void* in,
                                 const void* w, void* out, const float* bias, float* stats,
                                 const float* scale, const float* shift, const void* residual,
                                 int32_t ld_res, const void* mask_src, int32_t ld_mask,
                                 int32_t act, void* stream) {
  if (dtype == EMSA_DT_F32)
    return emsa_conv_igemm(g, (const float*)in, (const float*)w, (float*)out, bias, stats, scale,
                           shift, (const float*)residual, ld_res, (const float*)mask_src, ld_mask,
                           act, stream);
  return conv_igemm_h_impl(dtype, g, in, w, out, bias, stats, scale, shift, residual, ld_res,
                           mask_src, ld_mask, act, nullptr, nullptr, nullptr, 0, stream);
}

// 16-bit data gradient with the BatchNorm-backward sums of the layer in front fused into the
// epilogue (ConvHArgs::bnb_out; the fp32 twin is emsa_conv1d_wino_bnb): stores
// g = dz * (t * bn_scale + bn_shift > 0) and partial[0/1][tile][c] for the
// emsa_conv_stats_rows_t(dtype, g) pixel tiles; `partial` = float[2][rows_alloc][c],
// rows_alloc >= tiles + 16 (emsa_bn_bwd_apply_rows_t merges the rows).
extern "C" int emsa_conv_igemm_bnb_t(int32_t dtype, const EmsaConvGeom* g, const void* dy,
                                     const void* w, void* out, const void* residual,
                                     int32_t ld_res, const void* t, int32_t ld_t,
                                     const float* bn_scale, const float* bn_shift,
                                     const float* bn_mean, const float* bn_invstd, float* partial,
                                     int32_t rows_alloc, void* stream) {
  if (!partial || !t) return EMSA_E_ARG;
  return conv_igemm_h_impl(dtype, g, dy, w, out, nullptr, nullptr, bn_scale, bn_shift, residual,
                           ld_res, t, ld_t, EMSA_ACT_NONE, bn_mean, bn_invstd, partial, rows_alloc,
                           stream);
}

// ---- tap-split forward conv for maps with few output tiles (batch-1 inference) -------------------
// ksplit workgroups per output tile, each over a range of taps, raw fp32 partial sums in `ws`, then
// conv_h_splitk_finish_kernel (fixed summation order: deterministic).  emsa_conv_igemm_splitk_ws_bytes_t
// returns 0 where the plain launch is the better one (enough tiles, short K, fp32).
namespace {
int splitk_plan(int32_t dtype, const EmsaConvGeom* g) {
  if (dtype != EMSA_DT_BF16 && dtype != EMSA_DT_F16) return 1;
  if (!h_geom_ok(g) || g->out_pix_img || g->out_pix_row || g->out_pix_px || g->out_pix_off) return 1;
  static const bool off = [] {
    const char* e = getenv("EMSA_CONVH_SPLITK");
    return e && e[0] == '0';
  }();
  if (off) return 1;
  const int taps = g->kh * g->kw;
  if (taps < 3 || (long)taps * g->k_ch < 1024) return 1;
  const long M = (long)g->n_img * g->out_h * g->out_w;
  const HTile t = pick_htile(M, g->n_ch);
  const long tiles = ((M + ht_bm(t) - 1) / ht_bm(t)) * ((g->n_ch + ht_bn(t) - 1) / ht_bn(t));
  if (taps >= 9) return tiles <= 100 ? 9 : (tiles <= 256 ? 3 : 1);
  return tiles <= 64 ? taps : 1;
}
}  // namespace

extern "C" int64_t emsa_conv_igemm_splitk_ws_bytes_t(int32_t dtype, const EmsaConvGeom* g) {
  const int ks = splitk_plan(dtype, g);
  if (ks <= 1) return 0;
  return (int64_t)ks * g->n_img * g->out_h * g->out_w * g->n_ch * (int64_t)sizeof(float);
}

extern "C" int emsa_conv_igemm_splitk_t(int32_t dtype, const EmsaConvGeom* g, const void* in,
                                        const void* w, void* out, const float* bias,
                                        const float* scale, const float* shift,
                                        const void* residual, int32_t ld_res, int32_t act,
                                        float* ws, void* stream) {
  const int ks = splitk_plan(dtype, g);
  if (ks <= 1) return EMSA_E_SHAPE;
  if (!ws || (((uintptr_t)ws) & 15)) return EMSA_E_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return EMSA_E_ARG;
  int rc = conv_igemm_h_impl(dtype, g, in, w, out, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                             nullptr, 0, EMSA_ACT_NONE, nullptr, nullptr, nullptr, 0, stream, ws, ks);
  if (rc != EMSA_OK) return rc;
  const long M = (long)g->n_img * g->out_h * g->out_w;
  const long total = M * (g->n_ch >> 3);
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EMSA_DT_BF16)
    hipLaunchKernelGGL(conv_h_splitk_finish_kernel<emsa_bf16>, dim3(grid), dim3(256), 0, st, ws, ks, M,
                       g->n_ch, g->ld_out, bias, scale, shift, (const emsa_bf16*)residual, ld_res, act,
                       (emsa_bf16*)out);
  else
    hipLaunchKernelGGL(conv_h_splitk_finish_kernel<emsa_f16>, dim3(grid), dim3(256), 0, st, ws, ks, M,
                       g->n_ch, g->ld_out, bias, scale, shift, (const emsa_f16*)residual, ld_res, act,
                       (emsa_f16*)out);
  return emsa_launch_status();
}

// Twin launch of emsa_conv_igemm_t / emsa_conv_igemm_splitk_t: the same geometry on two independent
// sets of tensors in ONE launch (grid.y = 2) -- the strided convs and 1x1 skip convs of the rgb | depth
// encoder blocks, the 3x3 and 1x1 skip-fusion convs of the semantic | instance decoder modules
// (in0 may equal in1).  Forward epilogue only (bias, folded BatchNorm, residual, ReLU); each half ==
// its own launch bit for bit.  ws0 / ws1: emsa_conv_igemm_splitk_ws_bytes_t(dtype, g) bytes each where
// that is > 0 (the tap-split form is then used for both halves), else NULL.
extern "C" int emsa_conv_igemm_pair_t(int32_t dtype, const EmsaConvGeom* g, const void* in0,
                                      const void* in1, const void* w0, const void* w1, void* out0,
                                      void* out1, const float* bias0, const float* bias1,
                                      const float* scale0, const float* scale1, const float* shift0,
                                      const float* shift1, const void* residual0,
                                      const void* residual1, int32_t ld_res, int32_t act,
                                      float* ws0, float* ws1, void* stream) {
  if (dtype != EMSA_DT_BF16 && dtype != EMSA_DT_F16) return EMSA_E_SHAPE;
  if (!in1 || !w1 || !out1 || out0 == out1) return EMSA_E_ARG;
  if ((bias0 == nullptr) != (bias1 == nullptr) || (scale0 == nullptr) != (scale1 == nullptr) ||
      (shift0 == nullptr) != (shift1 == nullptr) || (residual0 == nullptr) != (residual1 == nullptr) ||
      (scale1 == nullptr) != (shift1 == nullptr))
    return EMSA_E_ARG;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (!al16(in1) || !al16(w1) || !al16(out1) || !al16(bias1) || !al16(scale1) || !al16(shift1) ||
      !al16(residual1) || !al16(ws0) || !al16(ws1))
    return EMSA_E_SHAPE;
  const int ks = splitk_plan(dtype, g);
  ConvHArgs tw;
  tw.in2 = in1; tw.w2 = w1; tw.out2 = out1; tw.bias2 = bias1; tw.scale2 = scale1; tw.shift2 = shift1;
  tw.residual2 = residual1; tw.ws2 = ws1;
  if (ks <= 1)
    return conv_igemm_h_impl(dtype, g, in0, w0, out0, bias0, nullptr, scale0, shift0, residual0, ld_res,
                             nullptr, 0, act, nullptr, nullptr, nullptr, 0, stream, nullptr, 1, &tw);
  if (!ws0 || !ws1 || ws0 == ws1) return EMSA_E_ARG;
  // tap-split: raw partial sums of both halves, then ONE finish launch for both
  tw.bias2 = nullptr; tw.scale2 = nullptr; tw.shift2 = nullptr; tw.residual2 = nullptr;
  int rc = conv_igemm_h_impl(dtype, g, in0, w0, out0, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                             nullptr, 0, EMSA_ACT_NONE, nullptr, nullptr, nullptr, 0, stream, ws0, ks, &tw);
  if (rc != EMSA_OK) return rc;
  const long M = (long)g->n_img * g->out_h * g->out_w;
  const long total = M * (g->n_ch >> 3);
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == EMSA_DT_BF16)
    hipLaunchKernelGGL(conv_h_splitk_finish_kernel<emsa_bf16>, dim3(grid, 2), dim3(256), 0, st, ws0, ks, M,
                       g->n_ch, g->ld_out, bias0, scale0, shift0, (const emsa_bf16*)residual0, ld_res, act,
                       (emsa_bf16*)out0, ws1, bias1, scale1, shift1, (const emsa_bf16*)residual1,
                       (emsa_bf16*)out1);
  else
    hipLaunchKernelGGL(conv_h_splitk_finish_kernel<emsa_f16>, dim3(grid, 2), dim3(256), 0, st, ws0, ks, M,
                       g->n_ch, g->ld_out, bias0, scale0, shift0, (const emsa_f16*)residual0, ld_res, act,
                       (emsa_f16*)out0, ws1, bias1, scale1, shift1, (const emsa_f16*)residual1,
                       (emsa_f16*)out1);
  return emsa_launch_status();
}
