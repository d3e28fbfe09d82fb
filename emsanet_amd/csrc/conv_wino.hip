// 1-D Winograd F(2,3) convolution on fp32 MFMA for the stride-1, 3-tap, "same"-padded 3x1 / 1x3
// convolutions of the NonBottleneck1D blocks (forward AND data gradient): 79 % of the model's
// MACs.
//
//   y(2p)   = m0 + m1 + m2          m_j = sum_c U_j[n][c] * V_j[p][c]          (4 GEMMs, j=0..3)
//   y(2p+1) = m1 - m2 - m3          V = (d0-d2, d1+d2, d2-d1, d1-d3),  d_r = x(2p-1+r)
//                                   U = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2)   (emsa_pack_wino)
// i.e. 4 MFMA GEMMs over HALF the pixels instead of 3 over all of them: 1.5x fewer matrix
// instructions for the same direct-convolution result (coefficients 1, 1/2: exact in fp32, the
// result differs from the direct kernel by ordinary fp32 roundoff).
//
// Workgroup = 32 output pairs (64 pixels along the convolution direction, pairs may sit on
// different lines) x 64 output channels, 4 waves: wave j owns Winograd component j (2 MFMA
// tiles of 32x32).  Per K step (16 channels) the RAW input rows d0..d3 of every pair are staged
// once ([4][32][20] floats) and wave j forms its V_j operand on the fly as the sum/difference of
// two conflict-free ds_read_b128; U_j is staged as [4][64][20].  Single LDS buffer + register
// prefetch (30 KiB -> 5 workgroups per CU).  The epilogue transposes m_j through LDS, applies the
// output transform and the same fused epilogue as conv_igemm (bias, BatchNorm statistics
// partials, folded BatchNorm, residual, ReLU, ReLU-backward mask) with 16-B stores.
#include <cstdlib>

#include "common.h"
#include "prof.h"

namespace {

constexpr int kWK = 16;          // channels per K step
constexpr int kWLD = kWK;        // LDS row = 16 floats (64 B), unpadded: 16-B chunks XOR-swizzled by the
                                 // row (wswz) -> conflict-free ds_write_b128 (8 lanes = 2 whole rows = 32
                                 // banks) AND ds_read_b128 (the 16 rows of a lane group hit 16 slots)
constexpr int kPairs = 32;       // output pairs per workgroup

typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
constexpr uint32_t kOOBw = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wrsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 wbuf_ld4(__amdgpu_buffer_rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
}
// per-thread offset in a VGPR + wave-uniform offset in an SGPR (the K-loop advance costs no VALU
// instruction: on gfx950 every VALU instruction takes its four cycles away from the fp32 MFMA
// pipe -- same ALUs, tools/mfma_peak.hip).  kOOBw in `voff` still reads as zero.
__device__ __forceinline__ float4 wbuf_ld4s(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float4,
                            __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
typedef float wf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int wu32x2 __attribute__((ext_vector_type(2)));

struct WinoArgs {
  const float* in;
  const float* u;
  float* out;
  const float* bias;
  float* stats;
  const float* scale;
  const float* shift;
  const float* residual;
  const float* mask_src;
  const uint64_t* mask_bits;        // alternative to mask_src: 1 bit per element, see relu_bits
  uint64_t* relu_bits;              // out: (result > 0), one word per (pixel, 64-channel tile):
                                    // bit = component*16 + float4 column
  // BatchNorm-backward sums fused into a data gradient (emsa_conv1d_wino_bnb): the result dz is the
  // gradient w.r.t. a = relu(bn(t)); the epilogue applies the ReLU mask recomputed from t
  // (mask_src = t, scale/shift = the BatchNorm's affine form, NOT applied to the result), stores
  // g = dz * (a > 0) and emits per tile  sum g  and  sum g * (t - mean) * invstd  -- the rows
  // bn_bwd_reduce would produce, without its pass over dz and t.
  const float* bnb_mean;
  const float* bnb_invstd;
  float* bnb_out;                   // [2][bnb_rows_alloc][n_ch], rows [0, tiles_m) written
  int bnb_rows_alloc;
  // BatchNorm + ReLU of the INPUT folded into the loader (emsa_conv1d_wino_inbn): the kernel
  // convolves a = relu(in * in_scale[c] + in_shift[c]) without a ever existing in memory -- the
  // NBt1D block's bn1 normalise+ReLU pass (one read + one write of the tensor) disappears into
  // conv3x1_2 (reference block: /root/reference/emsanet/model.py:47-58).  Padding stays zero.
  const float* in_scale;
  const float* in_shift;
  int ld_out, ld_res, ld_mask, act;
  int L, PL, A, MP;                 // line length, pairs per line, lines per image, total pairs
  int k_ch, n_ch;                   // k_ch = R * c_in (GEMM K), n_ch output channels
  int R, c_in, ksteps_c;            // perpendicular taps (1: 1-D conv, 3: 3x3), channels and K steps per tap
  int in_simg, in_sa, in_sb;        // input element strides (image, line, position on the line)
  int px_simg, px_sa, px_sb;        // output PIXEL strides (x ld_out / ld_res / ld_mask)
  int tiles_m, tiles_n, ksteps;
  uint32_t in_bytes, u_bytes;
  uint32_t mul_pl, sh_pl, mul_a, sh_a;     // magic division by PL and by A
};

// float offset of logical 16-B chunk `c4/4` of LDS row `row`: physical chunk = chunk ^ ((row>>2)&3)
__device__ __forceinline__ int wswz(int row, int c4) {
  return row * kWLD + (c4 ^ (((row >> 2) & 3) << 2));
}

__device__ __forceinline__ uint32_t wdiv(uint32_t n, uint32_t mul, uint32_t sh) {
  return (__umulhi(n, mul) + n) >> sh;
}

// BF16 (EMSA_BF16_MFMA=1, NOT the default and not the headline metric: BASELINE config 3's mixed
// precision): the operands are rounded to bf16 when they leave LDS and one
// v_mfma_f32_32x32x16_bf16 replaces eight fp32 MFMAs; accumulation, the Winograd transforms and
// everything in HBM stay fp32.
#ifndef EMSA_WINO_ST_AUX
#define EMSA_WINO_ST_AUX 0   // cache policy of the output stores (tuning builds: 2 = non-temporal)
#endif
#ifndef EMSA_WINO_PRIO
#define EMSA_WINO_PRIO 2   // 0 = no s_setprio, 1 = raised around the MFMA cluster, 2 = also: the next step's loads
                          // issue at the highest priority (measured +1..2 %)
#endif
// BNB: the epilogue with the fused BatchNorm-backward sums (WinoArgs::bnb_out) is its own
// instantiation with one wave per SIMD less: compiled into the common kernel it cost that one 13-15
// spilled registers at 5 waves per SIMD.
// INBN: the input's BatchNorm + ReLU applied where the K step enters LDS (WinoArgs::in_scale); 1-D
// convs only (R = 1).  Costs 12 vector instructions per K step and thread (2 packed FMAs + 4
// v_med3 per float4: max(v, 0) and the zero-padding mask in one instruction, cap = +inf / 0).
template <int kWN, bool BF16 = false, bool BNB = false, bool INBN = false>   // kWN: output channels per workgroup, 64 (2 MFMA tiles per wave) or 32
__global__ __launch_bounds__(256, kWN == 64 ? (BNB ? 4 : 5) : 6) void conv1d_wino_kernel(const WinoArgs p) {
  constexpr int NT = kWN / 32;                    // accumulator tiles per wave
  constexpr int NC4 = kWN / 4;                    // float4 columns of a tile row
  constexpr int RG = 256 / NC4;                   // pixel rows covered by one pass of the block
  constexpr int PXI = 64 / RG;                    // output pixels per thread in the epilogue
  constexpr int BLD = 4 * kWN * 4 / 256;          // B float4 loads per thread and K step
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const As = smem;                          // [4 rows][32 pairs][kWLD]
  float* const Bs = smem + 4 * kPairs * kWLD;      // [4 comps][kWN n][kWLD]
  float* const Ts = Bs + 4 * kWN * kWLD;           // INBN: [2: scale, shift][ksteps_c * kWK]

  // wave = component j; uniform -> kept in an SGPR (so are the row choice and sign below)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  // Tile order: every XCD (own 4 MiB L2; workgroup b runs on XCD b % 8) gets a contiguous range of
  // pixel tiles with ALL their channel tiles -- input rows are fetched into one L2 once.  (Round 3
  // measured the opposite assignment for the convs whose transformed weights exceed an L2 -- 3x3
  // 512->512, U = 12.6 MB: each XCD owning one channel tile for all pixel tiles -- at 293 vs 286 us
  // forward and 270 vs 270 us data gradient: the 545 MB of U re-fetches come from the Infinity
  // Cache, not from HBM, and are not what bounds the launch.  Not kept.)
  const int wg = emsa_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = wg % p.tiles_n, mt = wg / p.tiles_n;
  const int q0 = mt * kPairs, n0 = nt * kWN;
  const __amdgpu_buffer_rsrc_t rs_in = wrsrc(p.in, p.in_bytes);
  const __amdgpu_buffer_rsrc_t rs_u = wrsrc(p.u, p.u_bytes);

  // ---- loader state: a tile's pixels do not change over its K loop ----------------------
  // A: 4 rows x 32 pairs x 4 float4 = 512 float4 -> 2 per thread; B: 4 x 64 x 4 = 1024 -> 4.
  // Every load address = per-thread byte offset (fixed for the tile, or per kernel row of a 3x3)
  // + a wave-uniform K offset that travels in the instruction's scalar-offset operand.
  const int c4 = (tid & 3) * 4;
  uint32_t a_base[2], a_voff[2], b_voff[BLD];
  int a_line[2];                                    // line index within the image (3x3: row taps)
  auto setup_a = [&](int mtile) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rowid = (tid >> 2) + 64 * j;          // 0..127 = rr*32 + pair
      const int rr = rowid >> 5, pr = rowid & 31;
      const int q = mtile * kPairs + pr;
      uint32_t off = kOOBw;
      int al = 0;
      if (q < p.MP) {
        const int line = (int)wdiv((uint32_t)q, p.mul_pl, p.sh_pl);
        const int pp = q - line * p.PL;
        const int img = (int)wdiv((uint32_t)line, p.mul_a, p.sh_a);
        const int a = line - img * p.A;
        const int b = 2 * pp - 1 + rr;
        if (b >= 0 && b < p.L)
          off = (uint32_t)(img * p.in_simg + a * p.in_sa + b * p.in_sb + c4) * 4u;
        al = a;
      }
      a_base[j] = off;
      a_line[j] = al;
    }
  };
  setup_a(mt);
#pragma unroll
  for (int j = 0; j < BLD; ++j) {
    const int rowid = (tid >> 2) + 64 * j;          // 0..4*kWN-1 = comp*kWN + n
    const int comp = rowid / kWN, n = n0 + (rowid % kWN);
    b_voff[j] = n < p.n_ch ? (uint32_t)((comp * p.n_ch + n) * p.k_ch + c4) * 4u : kOOBw;
  }
  // offsets of the tile's pixels on kernel row r (r = 0 only for the 1-D convs): the line shifts
  // by r - 1 and may leave the image (zero padding)
  auto set_row = [&](int r) {
    const int dr = p.R == 3 ? r - 1 : 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int al = a_line[j] + dr;
      const bool ok = !(a_base[j] & kOOBw) && al >= 0 && al < p.A;
      a_voff[j] = ok ? a_base[j] + (uint32_t)(dr * p.in_sa * 4) : kOOBw;
    }
  };
  // kOOBw for the lanes whose channels lie beyond c_in in the LAST chunk of a row
  const uint32_t last_oob = (p.ksteps_c - 1) * kWK + c4 < p.c_in ? 0u : kOOBw;
  float4 ra[2], rb[BLD];
  auto load_regs = [&](int s) {
    // K step s = (perpendicular tap r, channel chunk), all wave-uniform
    const int r = s >= 2 * p.ksteps_c ? 2 : (s >= p.ksteps_c ? 1 : 0);
    const int ch = (s - r * p.ksteps_c) * kWK;                     // first input channel
    if (ch == 0) set_row(r);
    const uint32_t sa = (uint32_t)ch * 4u, sb = (uint32_t)(r * p.c_in + ch) * 4u;
    // last, partial channel chunk (c_in not a multiple of 16): the lanes beyond c_in go out of
    // range.  One v_and_or per load with a wave-uniform mask -- NOT a branch around the loads: a
    // control-flow join behind them makes the compiler wait for the data right there, before the
    // step's MFMAs (measured: +0.2 us per K step at batch 1)
    const uint32_t pm = ch + kWK > p.c_in ? 0xFFFFFFFFu : 0u;
#pragma unroll
    for (int j = 0; j < 2; ++j) ra[j] = wbuf_ld4s(rs_in, a_voff[j] | (last_oob & pm), sa);
#pragma unroll
    for (int j = 0; j < BLD; ++j) rb[j] = wbuf_ld4s(rs_u, b_voff[j] | (last_oob & pm), sb);
  };
  auto store_lds = [&](int s) {                     // s: the K step the registers hold
    if constexpr (INBN) {
      // a = relu(x * scale + shift); v_med3(v, 0, cap): max(v, 0) for cap = +inf, 0 for cap = 0
      const int kt = p.ksteps_c * kWK;
      const float4 sc = emsa_ld4(Ts + s * kWK + c4), sh = emsa_ld4(Ts + kt + s * kWK + c4);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // cap: +inf for a real pixel row, 0 for a padding row (top bit of the row's load offset;
        // R = 1: a_voff is fixed for the tile) -- rebuilt here rather than held in a register
        const float cap = __uint_as_float(0x7F800000u & ~(uint32_t)((int32_t)a_voff[j] >> 31));
        const wf32x2 lo = __builtin_elementwise_fma(wf32x2{ra[j].x, ra[j].y}, wf32x2{sc.x, sc.y},
                                                    wf32x2{sh.x, sh.y});
        const wf32x2 hi = __builtin_elementwise_fma(wf32x2{ra[j].z, ra[j].w}, wf32x2{sc.z, sc.w},
                                                    wf32x2{sh.z, sh.w});
        ra[j] = make_float4(__builtin_amdgcn_fmed3f(lo.x, 0.f, cap),
                            __builtin_amdgcn_fmed3f(lo.y, 0.f, cap),
                            __builtin_amdgcn_fmed3f(hi.x, 0.f, cap),
                            __builtin_amdgcn_fmed3f(hi.y, 0.f, cap));
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) emsa_st4(As + wswz((tid >> 2) + 64 * j, c4), ra[j]);
#pragma unroll
    for (int j = 0; j < BLD; ++j) emsa_st4(Bs + wswz((tid >> 2) + 64 * j, c4), rb[j]);
  };

  // wave j: V_j = row[ra_] + sg * row[rb_]
  const int ra_ = wave == 0 ? 0 : wave == 2 ? 2 : 1;
  const int rb_ = wave == 2 ? 1 : wave == 3 ? 3 : 2;
  const float sg = wave == 1 ? 1.f : -1.f;

  load_regs(0);
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if constexpr (INBN) {
    // the affine table of all input channels (zeros beyond c_in: those K entries meet zero
    // weights and must stay finite); R = 1, so the rows' validity is fixed for the tile
    const int kt = p.ksteps_c * kWK;
    for (int i = tid; i < kt; i += 256) {
      Ts[i] = i < p.c_in ? p.in_scale[i] : 0.f;
      Ts[kt + i] = i < p.c_in ? p.in_shift[i] : 0.f;
    }
    __syncthreads();
  }
  store_lds(0);
  __syncthreads();
  for (int s = 0; s < p.ksteps; ++s) {
    const bool has_next = s + 1 < p.ksteps;
#if EMSA_WINO_PRIO == 2
    __builtin_amdgcn_s_setprio(3);           // the next step's loads go out first
#endif
    if (has_next) load_regs(s + 1);
    // rows ra_*32 + l31 etc.: the swizzle term depends on (l31 >> 2) & 3 only (row bases are
    // multiples of 32); logical chunk of step t = lh + 2*t
    const int sw = ((l31 >> 2) & 3) << 2;
    const float* a0 = As + (ra_ * kPairs + l31) * kWLD;
    const float* a1 = As + (rb_ * kPairs + l31) * kWLD;
    const float* b = Bs + (wave * kWN + l31) * kWLD;
#if EMSA_WINO_PRIO != 0
    __builtin_amdgcn_s_setprio(1);
#endif
    if constexpr (BF16) {
      // this lane's 8 k values of the step: chunks lh and lh + 2 (the same permutation for A and B)
      const int c0 = (lh << 2) ^ sw, c1 = ((lh + 2) << 2) ^ sw;
      const float4 p0 = emsa_ld4(a0 + c0), q0 = emsa_ld4(a1 + c0);
      const float4 p1 = emsa_ld4(a0 + c1), q1 = emsa_ld4(a1 + c1);
      wbf16x8 va;
      va[0] = (__bf16)(p0.x + sg * q0.x); va[1] = (__bf16)(p0.y + sg * q0.y);
      va[2] = (__bf16)(p0.z + sg * q0.z); va[3] = (__bf16)(p0.w + sg * q0.w);
      va[4] = (__bf16)(p1.x + sg * q1.x); va[5] = (__bf16)(p1.y + sg * q1.y);
      va[6] = (__bf16)(p1.z + sg * q1.z); va[7] = (__bf16)(p1.w + sg * q1.w);
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const float4 w0 = emsa_ld4(b + u * 32 * kWLD + c0), w1 = emsa_ld4(b + u * 32 * kWLD + c1);
        wbf16x8 vb;
        vb[0] = (__bf16)w0.x; vb[1] = (__bf16)w0.y; vb[2] = (__bf16)w0.z; vb[3] = (__bf16)w0.w;
        vb[4] = (__bf16)w1.x; vb[5] = (__bf16)w1.y; vb[6] = (__bf16)w1.z; vb[7] = (__bf16)w1.w;
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[u], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int t = 0; t < kWK / 8; ++t) {
      const int co = ((lh + 2 * t) << 2) ^ sw;     // swizzled chunk offset (floats)
      const float4 x0 = emsa_ld4(a0 + co), x1 = emsa_ld4(a1 + co);
      // input transform x0 +- x1 as two packed FMAs (v_pk_fma_f32) instead of four scalar ones
      const wf32x2 sg2 = {sg, sg};
      const wf32x2 vlo = __builtin_elementwise_fma(wf32x2{x1.x, x1.y}, sg2, wf32x2{x0.x, x0.y});
      const wf32x2 vhi = __builtin_elementwise_fma(wf32x2{x1.z, x1.w}, sg2, wf32x2{x0.z, x0.w});
      const float4 v = make_float4(vlo.x, vlo.y, vhi.x, vhi.y);
      float4 fb[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) fb[u] = emsa_ld4(b + u * 32 * kWLD + co);
#pragma unroll
      for (int u = 0; u < NT; ++u)
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, fb[u].x, acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NT; ++u)
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, fb[u].y, acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NT; ++u)
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, fb[u].z, acc[u], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < NT; ++u)
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, fb[u].w, acc[u], 0, 0, 0);
    }
    }
#if EMSA_WINO_PRIO != 0
    __builtin_amdgcn_s_setprio(0);
#endif
    __syncthreads();
    if (has_next) store_lds(s + 1);
    __syncthreads();
  }

  // ---- epilogue: m_j -> LDS, output transform, fused epilogue ---------------------------------
  // in two halves of 16 pairs (stage [4 comps][16 pairs][kWN + 4] = 17 KiB: with the 24 KiB of the
  // main loop the workgroup stays under 32 KiB -> 5 workgroups per CU); thread
  // (px = tid/NC4 + RG*k, col4 = tid%NC4) owns PXI of the 64 output pixels x one float4 of channels
  constexpr int SLD = kWN + 4;
  constexpr int HP = kPairs / 2;
  // The epilogue is pure VALU work and on gfx950 every VALU instruction is four cycles the fp32
  // MFMA pipe does not get (for a 64-channel layer the K loop is only 64 MFMAs per wave), so it is
  // written for instruction count: one pixel decomposition per LANE (fetched with wave shuffles),
  // packed fp32 math, 32-bit buffer addressing with out-of-range offsets instead of branches.
  constexpr uint32_t kNoPix = 0xFFFFFFFFu;
  float* const stage = smem;
  const int col4 = tid % NC4, n = n0 + col4 * 4;
  const int rgi = tid / NC4;                       // row group of this thread
  const bool nok = n < p.n_ch;
  // lane l of every wave: element index (pixel * 1) of the tile's output pixel l = pair*2 + e
  uint32_t lane_pix = kNoPix;
  {
    const int q = q0 + (lane >> 1);
    if (q < p.MP) {
      const int line = (int)wdiv((uint32_t)q, p.mul_pl, p.sh_pl);
      const int pp = q - line * p.PL;
      const int img = (int)wdiv((uint32_t)line, p.mul_a, p.sh_a);
      const int a = line - img * p.A;
      const int b = 2 * pp + (lane & 1);
      if (b < p.L) lane_pix = (uint32_t)(img * p.px_simg + a * p.px_sa + b * p.px_sb);
    }
  }
  wf32x2 ylo[PXI], yhi[PXI];
  uint32_t opix[PXI];          // pixel index (< 2^29, checked by emsa_conv1d_wino_supported)
  bool ok[PXI];
  const float4 bv = (nok && p.bias) ? emsa_ld4(p.bias + n) : emsa_zero4();
  const wf32x2 blo = {bv.x, bv.y}, bhi = {bv.z, bv.w};
  // out(e=0) = m0 + m1 + m2, out(e=1) = m1 - m2 - m3  ==  m1 + sg * (m2 + m03); e = px & 1 is a
  // per-thread constant (RG is even)
  const float sgo = (rgi & 1) ? -1.f : 1.f;
  const wf32x2 sg2o = {sgo, sgo};
  const int m03row = (rgi & 1) ? 3 : 0;
#pragma unroll
  for (int k = 0; k < PXI; ++k) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute((rgi + RG * k) * 4, (int)lane_pix);
    ok[k] = nok && o != kNoPix;
    opix[k] = o;
  }
  // every tensor touched by the epilogue is < 2 GiB (checked by the launcher): byte offsets are
  // 32-bit and an offset with the top bit set is out of range for these descriptors -- loads
  // return zero, stores are dropped -- so dead pixels / channel columns need no branch
  const __amdgpu_buffer_rsrc_t rs_out = wrsrc(p.out, kOOBw);
  const __amdgpu_buffer_rsrc_t rs_res = wrsrc(p.residual, kOOBw);
  const __amdgpu_buffer_rsrc_t rs_msk = wrsrc(p.mask_src, kOOBw);
  const __amdgpu_buffer_rsrc_t rs_mb = wrsrc(p.mask_bits, kOOBw);
  // ALL the epilogue's loads are issued HERE, before the accumulators are staged through LDS:
  // their latency hides behind the staging and the statistics, and none of them sits behind a
  // store (a load behind a store is followed by `s_waitcnt vmcnt(0)`, which on gfx9 also waits
  // for the store's write acknowledgement -- four serialised round trips per workgroup)
  float4 rres[PXI], rmsk[PXI];
  wu32x2 rbits[PXI];
#pragma unroll
  for (int k = 0; k < PXI; ++k) {
    const bool live = ok[k];
    if (p.residual)
      rres[k] = wbuf_ld4(rs_res, live ? (opix[k] * (uint32_t)p.ld_res + (uint32_t)n) * 4u : kOOBw);
    if (p.mask_src)
      rmsk[k] = wbuf_ld4(rs_msk, live ? (opix[k] * (uint32_t)p.ld_mask + (uint32_t)n) * 4u : kOOBw);
    if constexpr (kWN == 64) {
      if (p.mask_bits)
        rbits[k] = __builtin_bit_cast(
            wu32x2, __builtin_amdgcn_raw_buffer_load_b64(
                        rs_mb, (int)(live ? (opix[k] * (uint32_t)p.tiles_n + (uint32_t)nt) * 8u
                                          : kOOBw), 0, 0));
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
  if (h) __syncthreads();                          // half 0 of the stage has been consumed
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 8 * h; r < 8 * h + 8; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;          // in [16h, 16h + 16)
      stage[(wave * HP + row - HP * h) * SLD + t * 32 + l31] = acc[t][r];
    }
  __syncthreads();
#pragma unroll
  for (int k = h * (PXI / 2); k < (h + 1) * (PXI / 2); ++k) {
    const int px = rgi + RG * k;                   // 0..63 = pair*2 + e, in [32h, 32h + 32)
    const int prl = (px >> 1) - HP * h;
    const float4 m1 = emsa_ld4(stage + (1 * HP + prl) * SLD + col4 * 4);
    const float4 m2 = emsa_ld4(stage + (2 * HP + prl) * SLD + col4 * 4);
    const float4 m03 = emsa_ld4(stage + (m03row * HP + prl) * SLD + col4 * 4);
    const wf32x2 tlo = wf32x2{m2.x, m2.y} + wf32x2{m03.x, m03.y};
    const wf32x2 thi = wf32x2{m2.z, m2.w} + wf32x2{m03.z, m03.w};
    ylo[k] = __builtin_elementwise_fma(tlo, sg2o, wf32x2{m1.x, m1.y}) + blo;
    yhi[k] = __builtin_elementwise_fma(thi, sg2o, wf32x2{m1.z, m1.w}) + bhi;
  }
  }   // halves

  if (p.stats != nullptr) {
    // per-tile (sum, M2 about the tile mean, count) over the tile's VALID output pixels
    __syncthreads();                               // everyone has read `stage`
    float* red = smem;                             // [RG row groups][kWN]
    float* tmean = smem + RG * kWN;                // [kWN]
    int* scnt = reinterpret_cast<int*>(tmean + kWN);   // [RG] valid pixels per row group
    wf32x2 s1lo = {0.f, 0.f}, s1hi = {0.f, 0.f};
    float wgt[PXI];
    int cnt = 0;
#pragma unroll
    for (int k = 0; k < PXI; ++k) {
      wgt[k] = ok[k] ? 1.f : 0.f;
      const wf32x2 w2 = {wgt[k], wgt[k]};
      s1lo = __builtin_elementwise_fma(ylo[k], w2, s1lo);
      s1hi = __builtin_elementwise_fma(yhi[k], w2, s1hi);
      cnt += ok[k] ? 1 : 0;
    }
    emsa_st4(red + rgi * kWN + col4 * 4, make_float4(s1lo.x, s1lo.y, s1hi.x, s1hi.y));
    // the valid-pixel count is the same for every channel column: column 0 threads publish it
    // (n0 < n_ch for every launched tile, so their `ok` flags are the pixel validity)
    if (col4 == 0) scnt[rgi] = cnt;
    __syncthreads();
    if (tid < kWN) {
      float a1 = 0.f;
#pragma unroll
      for (int g = 0; g < RG; ++g) a1 += red[g * kWN + tid];
      int c = 0;
#pragma unroll
      for (int g = 0; g < RG; ++g) c += scnt[g];
      tmean[tid] = c > 0 ? a1 / (float)c : 0.f;
      if (n0 + tid < p.n_ch) {
        p.stats[((size_t)0 * p.tiles_m + mt) * p.n_ch + n0 + tid] = a1;
        p.stats[((size_t)2 * p.tiles_m + mt) * p.n_ch + n0 + tid] = (float)c;
      }
    }
    __syncthreads();
    const float4 mu = emsa_ld4(tmean + col4 * 4);
    const wf32x2 mulo = {mu.x, mu.y}, muhi = {mu.z, mu.w};
    wf32x2 s2lo = {0.f, 0.f}, s2hi = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < PXI; ++k) {
      const wf32x2 w2 = {wgt[k], wgt[k]};
      const wf32x2 dlo = ylo[k] - mulo, dhi = yhi[k] - muhi;
      s2lo = __builtin_elementwise_fma(dlo * w2, dlo, s2lo);
      s2hi = __builtin_elementwise_fma(dhi * w2, dhi, s2hi);
    }
    __syncthreads();
    emsa_st4(red + rgi * kWN + col4 * 4, make_float4(s2lo.x, s2lo.y, s2hi.x, s2hi.y));
    __syncthreads();
    if (tid < kWN && n0 + tid < p.n_ch) {
      float a2 = 0.f;
#pragma unroll
      for (int g = 0; g < RG; ++g) a2 += red[g * kWN + tid];
      p.stats[((size_t)1 * p.tiles_m + mt) * p.n_ch + n0 + tid] = a2;
    }
  }

  {
    wf32x2 sclo = {1.f, 1.f}, schi = {1.f, 1.f}, shlo = {0.f, 0.f}, shhi = {0.f, 0.f};
    constexpr bool bnb = BNB;
    const bool affine = p.scale != nullptr && !bnb;
    if (p.scale != nullptr && nok) {
      const float4 sc = emsa_ld4(p.scale + n), sh = emsa_ld4(p.shift + n);
      sclo = wf32x2{sc.x, sc.y}; schi = wf32x2{sc.z, sc.w};
      shlo = wf32x2{sh.x, sh.y}; shhi = wf32x2{sh.z, sh.w};
    }
    wf32x2 mulo = {0.f, 0.f}, muhi = {0.f, 0.f};
    wf32x2 b0lo = {0.f, 0.f}, b0hi = {0.f, 0.f}, b1lo = {0.f, 0.f}, b1hi = {0.f, 0.f};
    if (bnb && nok) {
      const float4 mu = emsa_ld4(p.bnb_mean + n);
      mulo = wf32x2{mu.x, mu.y}; muhi = wf32x2{mu.z, mu.w};
    }
#pragma unroll
    for (int k = 0; k < PXI; ++k) {
      const bool live = ok[k];
      wf32x2 vlo = ylo[k], vhi = yhi[k];
      if (affine) {
        vlo = __builtin_elementwise_fma(vlo, sclo, shlo);
        vhi = __builtin_elementwise_fma(vhi, schi, shhi);
      }
      if (p.residual) {
        vlo += wf32x2{rres[k].x, rres[k].y};
        vhi += wf32x2{rres[k].z, rres[k].w};
      }
      float4 v = make_float4(vlo.x, vlo.y, vhi.x, vhi.y);
      if constexpr (bnb) {
        // rmsk holds t (the BatchNorm input): a = t * scale + shift exactly as bn_act_fwd forms it
        const float4 tt = rmsk[k];
        const wf32x2 alo = __builtin_elementwise_fma(wf32x2{tt.x, tt.y}, sclo, shlo);
        const wf32x2 ahi = __builtin_elementwise_fma(wf32x2{tt.z, tt.w}, schi, shhi);
        const float lv = live ? 1.f : 0.f;
        v.x = alo.x > 0.f ? v.x * lv : 0.f; v.y = alo.y > 0.f ? v.y * lv : 0.f;
        v.z = ahi.x > 0.f ? v.z * lv : 0.f; v.w = ahi.y > 0.f ? v.w * lv : 0.f;
        const wf32x2 glo = {v.x, v.y}, ghi = {v.z, v.w};
        b0lo += glo; b0hi += ghi;
        b1lo = __builtin_elementwise_fma(glo, wf32x2{tt.x, tt.y} - mulo, b1lo);
        b1hi = __builtin_elementwise_fma(ghi, wf32x2{tt.z, tt.w} - muhi, b1hi);
      } else if (p.mask_src) {
        const float4 mm = rmsk[k];
        v.x = mm.x > 0.f ? v.x : 0.f; v.y = mm.y > 0.f ? v.y : 0.f;
        v.z = mm.z > 0.f ? v.z : 0.f; v.w = mm.w > 0.f ? v.w : 0.f;
      }
      if constexpr (kWN == 64) {
        // ReLU masks as bits: the 16 lanes that share a pixel share one 64-bit word; bit
        // comp * 16 + col4 -> x, y in the low word, z, w in the high word
        if (p.mask_bits) {
          const uint32_t t0 = rbits[k].x >> col4, t1 = rbits[k].y >> col4;
          v.x = __uint_as_float(__float_as_uint(v.x) & (uint32_t)__builtin_amdgcn_sbfe((int)t0, 0, 1));
          v.y = __uint_as_float(__float_as_uint(v.y) & (uint32_t)__builtin_amdgcn_sbfe((int)t0, 16, 1));
          v.z = __uint_as_float(__float_as_uint(v.z) & (uint32_t)__builtin_amdgcn_sbfe((int)t1, 0, 1));
          v.w = __uint_as_float(__float_as_uint(v.w) & (uint32_t)__builtin_amdgcn_sbfe((int)t1, 16, 1));
        }
      }
      if (p.act == EMSA_ACT_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      {
        const uint32_t off = live ? (opix[k] * (uint32_t)p.ld_out + (uint32_t)n) * 4u : kOOBw;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4w, v), rs_out, (int)off, 0,
                                               EMSA_WINO_ST_AUX);
      }
      if constexpr (kWN == 64) {
        if (p.relu_bits) {                           // uniform branch: ballots see every lane
          const uint64_t b0 = __ballot(live && v.x > 0.f), b1 = __ballot(live && v.y > 0.f);
          const uint64_t b2 = __ballot(live && v.z > 0.f), b3 = __ballot(live && v.w > 0.f);
          const int sh16 = (lane >> 4) * 16;         // this pixel's 16 lanes within the wave
          if (col4 == 0 && opix[k] != kNoPix)
            p.relu_bits[(size_t)opix[k] * p.tiles_n + nt] =
                ((b0 >> sh16) & 0xFFFFull) | (((b1 >> sh16) & 0xFFFFull) << 16) |
                (((b2 >> sh16) & 0xFFFFull) << 32) | (((b3 >> sh16) & 0xFFFFull) << 48);
        }
      }
    }
    if constexpr (bnb) {
      // the RG row groups' sums -> one value per (tile, channel), fixed order (no atomics)
      __syncthreads();                             // every wave is done with `stage`
      float* red0 = smem;                          // [RG][kWN]
      float* red1 = smem + RG * kWN;
      emsa_st4(red0 + rgi * kWN + col4 * 4, make_float4(b0lo.x, b0lo.y, b0hi.x, b0hi.y));
      emsa_st4(red1 + rgi * kWN + col4 * 4, make_float4(b1lo.x, b1lo.y, b1hi.x, b1hi.y));
      __syncthreads();
      if (tid < kWN && n0 + tid < p.n_ch) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int g2 = 0; g2 < RG; ++g2) {
          a0 += red0[g2 * kWN + tid];
          a1 += red1[g2 * kWN + tid];
        }
        p.bnb_out[((size_t)0 * p.bnb_rows_alloc + mt) * p.n_ch + n0 + tid] = a0;
        p.bnb_out[((size_t)1 * p.bnb_rows_alloc + mt) * p.n_ch + n0 + tid] =
            a1 * p.bnb_invstd[n0 + tid];
      }
    }
  }
}

// U[4][n][R*K] from OIHW weights [co][ci][R][3] (R = 1: 3x1 / 1x3 taps contiguous; R = 3: 3x3,
// Winograd along kw): forward (n = cout, K = cin) and/or data gradient (n = cin, K = cout, taps
// flipped in both directions)
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ u,
                                 float* __restrict__ ud, int cout, int cin, int R) {
  const int total = cout * cin * R;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i % R, ci = (i / R) % cin, co = i / (R * cin);
    const float* g = w + (size_t)i * 3;          // [co][ci][r][3]
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    const float s = 0.5f * (g0 + g2), h = 0.5f * g1;
    if (u) {
      const size_t NK = (size_t)cout * R * cin, o = ((size_t)co * R + r) * cin + ci;
      u[0 * NK + o] = g0;
      u[1 * NK + o] = s + h;
      u[2 * NK + o] = s - h;
      u[3 * NK + o] = g2;
    }
    if (ud) {
      const size_t NK = (size_t)cin * R * cout, o = ((size_t)ci * R + (R - 1 - r)) * cout + co;
      ud[0 * NK + o] = g2;
      ud[1 * NK + o] = s + h;
      ud[2 * NK + o] = s - h;
      ud[3 * NK + o] = g0;
    }
  }
}

// same transform from the PACKED implicit-GEMM layout wp[tap = r*3 + t][n][k] (what the merged /
// channel-padded head convs are assembled in): U_j[n][r*k_ch + k]; flip = 1 for the data-gradient
// pack ([tap][cin][cout], taps not flipped there) -> taps flipped here
__global__ void pack_wino_packed_kernel(const float* __restrict__ wp, float* __restrict__ u, int n,
                                        int k, int R, int flip) {
  const int total = n * k * R;
  const size_t NK = (size_t)total, plane = (size_t)n * k;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int kk = i % k, r = (i / k) % R, nn = i / (k * R);
    const int rs = flip ? R - 1 - r : r;
    const float* g = wp + (size_t)rs * 3 * plane + (size_t)nn * k + kk;
    float g0 = g[0], g1 = g[plane], g2 = g[2 * plane];
    if (flip) { const float t = g0; g0 = g2; g2 = t; }
    const float s = 0.5f * (g0 + g2), h = 0.5f * g1;
    const size_t o = ((size_t)nn * R + r) * k + kk;
    u[0 * NK + o] = g0;
    u[1 * NK + o] = s + h;
    u[2 * NK + o] = s - h;
    u[3 * NK + o] = g2;
  }
}


// ---- all weight transforms of a model in ONE launch ------------------------------------------
// (the per-layer transforms above are ~5 us kernels: 380 of them per training step cost 2.3 ms)
// job j owns workgroups [first_block[j], first_block[j+1]); kind 0: OIHW -> packed implicit-GEMM
// layouts (dst0 = [tap][cout][cin], dst1 = [tap][cin][cout]); kind 1: OIHW -> Winograd U
// (dst0 forward, dst1 data gradient; rows = 3 for a 3x3 kernel); NULL outputs are skipped
// kind 2 / 3: as kind 0 with bf16 / fp16 outputs (operands of conv_h.hip)
// kind 5 / 6: as 2 / 3 in the fragment order of conv_rs.hip (channel counts multiples of 32)
// kind 5 / 6: bf16 / fp16 in the MFMA-fragment order of conv_rs.hip (emsa_pack_weight_frag_t):
// element (tap r, row n, column k) of an [r][nT][kT] operand lives at
//   (((r * nT/32 + n/32) * kT/16 + k/16) * 64 + n%32 + 32 * ((k%16)/8)) * 8 + k%8
__device__ __forceinline__ void pack_store(float* base, int r, int n, int k, int nT, int kT,
                                           float v, int kind) {
  size_t idx = ((size_t)r * nT + n) * kT + k;
  if (kind == 5 || kind == 6)
    idx = ((((size_t)r * (nT >> 5) + (n >> 5)) * (kT >> 4) + (k >> 4)) * 64 + (n & 31) +
           32 * ((k & 15) >> 3)) * 8 + (k & 7);
  if (kind == 2 || kind == 5)
    reinterpret_cast<emsa_bf16*>(base)[idx] = (emsa_bf16)v;
  else if (kind == 3 || kind == 6)
    reinterpret_cast<emsa_f16*>(base)[idx] = (emsa_f16)v;
  else
    base[idx] = v;
}
__global__ __launch_bounds__(256) void pack_batch_kernel(const EmsaPackJob* __restrict__ jobs,
                                                         int n_jobs) {
  // 32 (co) x 32 (ci) tiles: reads and forward-layout writes run along ci, the transposed
  // (data-gradient) layouts are written along co through an LDS transpose -- the element-wise
  // version scattered 4-byte stores with stride cout and took 1.2 ms for the 63 M weights
  __shared__ float tr[4][32][33];
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {                       // last job with first_block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const EmsaPackJob jb = jobs[lo];
  const int nblk = (lo + 1 < n_jobs ? jobs[lo + 1].first_block : (int)gridDim.x) - jb.first_block;
  const int cout = jb.cout, cin = jb.cin;
  // placement inside a wider operand (padded / merged heads); totals default to the own size
  const int coT = jb.cout_total > 0 ? jb.cout_total : cout, ciT = jb.cin_total > 0 ? jb.cin_total : cin;
  const int coO = jb.cout_off, ciO = jb.cin_off;
  if (jb.kind == 4) {                    // bias vector copy
    for (int i = (blockIdx.x - jb.first_block) * 256 + threadIdx.x; i < cout; i += nblk * 256)
      jb.dst0[coO + i] = jb.src[i];
    return;
  }
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
  const int tco = (cout + 31) / 32, tci = (cin + 31) / 32;
  const bool wino = jb.kind == 1;
  const int taps = jb.kh * jb.kw;
  const int R = wino ? ((jb.kh == 3 && jb.kw == 3) ? 3 : 1) : taps;   // slices per (co, ci)
  const int n_tiles = tco * tci * R;
  for (int tile = blockIdx.x - jb.first_block; tile < n_tiles; tile += nblk) {
    const int r = tile % R, t2 = tile / R, ci0 = (t2 % tci) * 32, co0 = (t2 / tci) * 32;
    float vals[4][4];                      // [row group k][component]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int co = co0 + ty + 8 * k, ci = ci0 + tx;
      const bool ok = co < cout && ci < cin;
      if (wino) {
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
        if (ok) {
          const float* g = jb.src + (((size_t)co * cin + ci) * R + r) * 3;
          g0 = g[0]; g1 = g[1]; g2 = g[2];
        }
        const float sm = 0.5f * (g0 + g2), h = 0.5f * g1;
        vals[k][0] = g0; vals[k][1] = sm + h; vals[k][2] = sm - h; vals[k][3] = g2;
        if (ok && jb.dst0) {
          const size_t NK = (size_t)coT * R * ciT, o = ((size_t)(co + coO) * R + r) * ciT + ci + ciO;
#pragma unroll
          for (int j = 0; j < 4; ++j) jb.dst0[j * NK + o] = vals[k][j];
        }
      } else {
        const float v = ok ? jb.src[((size_t)co * cin + ci) * taps + r] : 0.f;
        vals[k][0] = v;
        if (ok && jb.dst0)
          pack_store(jb.dst0, r, co + coO, ci + ciO, coT, ciT, v, jb.kind);
      }
    }
    if (jb.dst1) {
      const int ncomp = wino ? 4 : 1;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k)
        for (int j = 0; j < ncomp; ++j) tr[j][ty + 8 * k][tx] = vals[k][j];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int ci = ci0 + ty + 8 * k, co = co0 + tx;          // transposed roles
        if (ci < cin && co < cout) {
          if (wino) {
            // data gradient: taps flipped (components 0 <-> 3) and kernel rows reversed
            const size_t NK = (size_t)ciT * R * coT,
                         o = ((size_t)(ci + ciO) * R + (R - 1 - r)) * coT + co + coO;
            jb.dst1[0 * NK + o] = tr[3][tx][ty + 8 * k];
            jb.dst1[1 * NK + o] = tr[1][tx][ty + 8 * k];
            jb.dst1[2 * NK + o] = tr[2][tx][ty + 8 * k];
            jb.dst1[3 * NK + o] = tr[0][tx][ty + 8 * k];
          } else {
            pack_store(jb.dst1, r, ci + ciO, co + coO, ciT, coT, tr[0][tx][ty + 8 * k], jb.kind);
          }
        }
      }
    }
  }
}

inline void magic(uint32_t d, uint32_t& mul, uint32_t& sh) {
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  sh = s;
  mul = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
}

}  // namespace

extern "C" int emsa_pack_wino(const float* w_oihw, float* u, float* u_dgrad, int32_t cout,
                              int32_t cin, int32_t rows, void* stream) {
  if (!w_oihw || (!u && !u_dgrad)) return EMSA_E_ARG;
  if (rows != 1 && rows != 3) return EMSA_E_SHAPE;
  const int total = cout * cin * rows;
  int grid = (total + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pack_wino_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, u,
                     u_dgrad, cout, cin, rows);
  return emsa_launch_status();
}

extern "C" int emsa_pack_batch(const EmsaPackJob* jobs_device, int32_t n_jobs,
                               int32_t total_blocks, void* stream) {
  if (!jobs_device) return EMSA_E_ARG;
  if (n_jobs < 1 || total_blocks < n_jobs) return EMSA_E_SHAPE;
  hipLaunchKernelGGL(pack_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream,
                     jobs_device, n_jobs);
  return emsa_launch_status();
}

extern "C" int emsa_pack_wino_packed(const float* w_packed, float* u, int32_t n_ch, int32_t k_ch,
                                     int32_t rows, int32_t flip, void* stream) {
  if (!w_packed || !u) return EMSA_E_ARG;
  if (rows != 1 && rows != 3) return EMSA_E_SHAPE;
  const int total = n_ch * k_ch * rows;
  int grid = (total + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pack_wino_packed_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     w_packed, u, n_ch, k_ch, rows, flip ? 1 : 0);
  return emsa_launch_status();
}

// 1 if `g` (forward or data-gradient geometry of a conv) is a stride-1 3-tap 1-D "same" conv
extern "C" int emsa_conv1d_wino_supported(const EmsaConvGeom* g) {
  if (g && (g->out_pix_img || g->out_pix_row || g->out_pix_px || g->out_pix_off)) return 0;
  if (!g) return 0;
  // forward geometry: off = -1, step = +1; data gradient: off = +1, step = -1
  auto tap3 = [](int off, int step) { return (off == -1 || off == 1) && step == -off; };
  const bool aw = g->kh == 1 && g->kw == 3 && g->off_h == 0 && tap3(g->off_w, g->step_w);
  const bool ah = g->kh == 3 && g->kw == 1 && g->off_w == 0 && tap3(g->off_h, g->step_h);
  const bool sq = g->kh == 3 && g->kw == 3 && tap3(g->off_w, g->step_w) &&
                  tap3(g->off_h, g->step_h) && g->off_h == g->off_w;
  if (!(aw || ah || sq)) return 0;
  if (g->mul_h != 1 || g->mul_w != 1 || g->div_h != 1 || g->div_w != 1) return 0;
  if (g->in_h != g->out_h || g->in_w != g->out_w) return 0;
  if ((g->k_ch & 3) || (g->n_ch & 3) || (g->ld_out & 3) || (g->in_px_stride & 3)) return 0;
  if ((long)g->n_img * g->in_img_stride >= (1L << 29)) return 0;
  return 1;
}

extern "C" int64_t emsa_conv_relu_bits_words(int64_t pixels, int32_t n_ch) {
  return pixels * ((n_ch + 63) / 64);
}

extern "C" int emsa_conv1d_wino_stats_rows(const EmsaConvGeom* g) {
  if (!emsa_conv1d_wino_supported(g)) return EMSA_E_SHAPE;
  const bool aw = g->kw == 3;
  const int L = aw ? g->out_w : g->out_h, A = aw ? g->out_h : g->out_w;
  const long mp = (long)g->n_img * A * ((L + 1) / 2);
  return (int)((mp + kPairs - 1) / kPairs);
}

static int conv1d_wino_impl(const EmsaConvGeom* g, const float* in, const float* u, float* out,
                            const float* bias, float* stats, const float* scale,
                            const float* shift, const float* residual, int32_t ld_res,
                            const float* mask_src, int32_t ld_mask, int32_t act,
                            const uint64_t* mask_bits, uint64_t* relu_bits,
                            const float* bnb_mean, const float* bnb_invstd, float* bnb_out,
                            int32_t bnb_rows_alloc, const float* in_scale, const float* in_shift,
                            void* stream) {
  if (!emsa_conv1d_wino_supported(g)) return EMSA_E_SHAPE;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return EMSA_E_ARG;
  if (!in || !u || !out) return EMSA_E_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return EMSA_E_ARG;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (!al16(out) || !al16(bias) || !al16(scale) || !al16(shift) ||
      (residual && (!al16(residual) || (ld_res & 3))) ||
      (mask_src && (!al16(mask_src) || (ld_mask & 3))))
    return EMSA_E_SHAPE;
  WinoArgs a;
  a.in = in; a.u = u; a.out = out; a.bias = bias; a.stats = stats; a.scale = scale;
  a.shift = shift; a.residual = residual; a.mask_src = mask_src;
  a.mask_bits = mask_bits; a.relu_bits = relu_bits;
  a.bnb_mean = bnb_mean; a.bnb_invstd = bnb_invstd; a.bnb_out = bnb_out;
  a.bnb_rows_alloc = bnb_rows_alloc;
  a.in_scale = in_scale; a.in_shift = in_shift;
  a.ld_out = g->ld_out; a.ld_res = ld_res; a.ld_mask = ld_mask; a.act = act;
  const bool aw = g->kw == 3;
  const int H = g->out_h, W = g->out_w;
  a.L = aw ? W : H;
  a.A = aw ? H : W;
  a.PL = (a.L + 1) / 2;
  a.MP = g->n_img * a.A * a.PL;
  a.R = g->kh == 3 && g->kw == 3 ? 3 : 1;
  a.c_in = g->k_ch;
  a.k_ch = a.R * g->k_ch; a.n_ch = g->n_ch;
  a.in_simg = (int)g->in_img_stride;
  a.in_sa = aw ? (int)g->in_row_stride : g->in_px_stride;
  a.in_sb = aw ? g->in_px_stride : (int)g->in_row_stride;
  a.px_simg = H * W;
  a.px_sa = aw ? W : 1;
  a.px_sb = aw ? 1 : W;
  // channel-tile width: EMSA_WINO_WN=32|64 forces it (tuning)
  static const int forced_wn = [] {
    const char* e = getenv("EMSA_WINO_WN");
    return e ? atoi(e) : 0;
  }();
  const int wn = forced_wn == 32 || forced_wn == 64 ? forced_wn : 64;
  if ((mask_bits || relu_bits) && wn != 64) return EMSA_E_SHAPE;
  if (mask_bits && mask_src) return EMSA_E_ARG;
  a.tiles_m = (a.MP + kPairs - 1) / kPairs;
  a.tiles_n = (g->n_ch + wn - 1) / wn;
  if (bnb_out) {
    // fused BatchNorm-backward sums: t through the mask_src operand, the affine form through
    // scale / shift; room for the tile rows plus the slice sums of emsa_bn_bwd_apply_rows
    if (!mask_src || !scale || !bnb_mean || !bnb_invstd || mask_bits || relu_bits || stats ||
        act != EMSA_ACT_NONE || !al16(bnb_mean) || bnb_rows_alloc < a.tiles_m)
      return EMSA_E_ARG;
  }
  {
    // the epilogue addresses out / residual / mask with 32-bit byte offsets below 2 GiB
    const long px = (long)g->n_img * H * W, lim = 1L << 29;
    if (px * g->ld_out >= lim || (residual && px * ld_res >= lim) ||
        (mask_src && px * ld_mask >= lim) || px * a.tiles_n * 2 >= lim)
      return EMSA_E_SHAPE;
  }
  a.ksteps_c = (g->k_ch + kWK - 1) / kWK;
  a.ksteps = a.R * a.ksteps_c;
  a.in_bytes = (uint32_t)((size_t)g->n_img * g->in_img_stride * sizeof(float));
  a.u_bytes = (uint32_t)((size_t)4 * g->n_ch * a.k_ch * sizeof(float));
  magic((uint32_t)a.PL, a.mul_pl, a.sh_pl);
  magic((uint32_t)a.A, a.mul_a, a.sh_a);
  if (in_scale && (a.R != 1 || bnb_out || wn != 64)) return EMSA_E_SHAPE;   // 1-D forward convs only
  const size_t lds_main = (size_t)(4 * kPairs + 4 * wn) * kWLD * sizeof(float) +
                          (in_scale ? (size_t)2 * a.ksteps_c * kWK * sizeof(float) : 0);
  const size_t lds_epi = (size_t)4 * (kPairs / 2) * (wn + 4) * sizeof(float);
  const size_t lds = lds_main > lds_epi ? lds_main : lds_epi;
  // algorithmic = direct-convolution FLOPs (3 taps); the kernel executes 4/6 of them on the MFMA
  const double flops = 2.0 * g->n_img * H * W * (double)g->k_ch * g->n_ch * 3.0 * a.R;
  const int ps = emsa_prof_begin(a.R == 3 ? kProfClassWino3x3 : kProfClassWino, flops,
                                 (hipStream_t)stream);
  // one workgroup per tile: a persistent grid with cross-tile prefetch measured SLOWER on MI355X
  // (static tile partition quantises to whole rounds: c256 /16 124 us vs 103 us)
  static const bool bf16 = [] {
    const char* e = getenv("EMSA_BF16_MFMA");
    return e && e[0] == '1';
  }();
  if (in_scale) {
    hipLaunchKernelGGL((conv1d_wino_kernel<64, false, false, true>),
                       dim3(a.tiles_m * a.tiles_n), dim3(256), lds, (hipStream_t)stream, a);
  } else if (bnb_out) {
    if (wn != 64) return EMSA_E_SHAPE;
    hipLaunchKernelGGL((conv1d_wino_kernel<64, false, true>), dim3(a.tiles_m * a.tiles_n),
                       dim3(256), lds, (hipStream_t)stream, a);
  } else if (wn == 64 && bf16)
    hipLaunchKernelGGL((conv1d_wino_kernel<64, true>), dim3(a.tiles_m * a.tiles_n), dim3(256), lds,
                       (hipStream_t)stream, a);
  else if (wn == 64)
    hipLaunchKernelGGL((conv1d_wino_kernel<64, false>), dim3(a.tiles_m * a.tiles_n), dim3(256),
                       lds, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((conv1d_wino_kernel<32, false>), dim3(a.tiles_m * a.tiles_n), dim3(256),
                       lds, (hipStream_t)stream, a);
  emsa_prof_end(ps, (hipStream_t)stream);
  return emsa_launch_status();
}

extern "C" int emsa_conv1d_wino(const EmsaConvGeom* g, const float* in, const float* u, float* out,
                                const float* bias, float* stats, const float* scale,
                                const float* shift, const float* residual, int32_t ld_res,
                                const float* mask_src, int32_t ld_mask, int32_t act,
                                const uint64_t* mask_bits, uint64_t* relu_bits, void* stream) {
  return conv1d_wino_impl(g, in, u, out, bias, stats, scale, shift, residual, ld_res, mask_src,
                          ld_mask, act, mask_bits, relu_bits, nullptr, nullptr, nullptr, 0, nullptr,
                          nullptr, stream);
}

// Forward 1-D conv of a = relu(in * in_scale[c] + in_shift[c]) (per INPUT channel, c_in floats
// each) with the affine map + ReLU applied in the loader: `in` is the BatchNorm's INPUT, the
// normalised tensor is never written (see WinoArgs::in_scale).  Zero padding pads `a`, not `in`.
extern "C" int emsa_conv1d_wino_inbn(const EmsaConvGeom* g, const float* in, const float* u,
                                     float* out, const float* bias, float* stats,
                                     const float* in_scale, const float* in_shift, int32_t act,
                                     uint64_t* relu_bits, void* stream) {
  if (!in_scale || !in_shift) return EMSA_E_ARG;
  return conv1d_wino_impl(g, in, u, out, bias, stats, nullptr, nullptr, nullptr, 0, nullptr, 0, act,
                          nullptr, relu_bits, nullptr, nullptr, nullptr, 0, in_scale, in_shift,
                          stream);
}

// Data gradient with the BatchNorm-backward sums of the layer in front fused into the epilogue
// (see WinoArgs::bnb_out): dz -> g = dz * (t * bn_scale + bn_shift > 0) stored to `out`, and
// partial[0][tile][c] = sum g, partial[1][tile][c] = sum g * (t - mean) * invstd for the
// emsa_conv1d_wino_stats_rows(g) tiles; `partial` = float[2][rows_alloc][c] with rows_alloc >=
// tiles + 16 (emsa_bn_bwd_apply_rows merges the rows).
extern "C" int emsa_conv1d_wino_bnb(const EmsaConvGeom* g, const float* dy, const float* u,
                                    float* out, const float* residual, int32_t ld_res,
                                    const float* t, int32_t ld_t, const float* bn_scale,
                                    const float* bn_shift, const float* bn_mean,
                                    const float* bn_invstd, float* partial, int32_t rows_alloc,
                                    void* stream) {
  if (!partial || !t) return EMSA_E_ARG;
  return conv1d_wino_impl(g, dy, u, out, nullptr, nullptr, bn_scale, bn_shift, residual, ld_res, t,
                          ld_t, EMSA_ACT_NONE, nullptr, nullptr, bn_mean, bn_invstd, partial,
                          rows_alloc, nullptr, nullptr, stream);
}
