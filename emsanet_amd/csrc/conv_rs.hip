// Register-stationary streaming convolution for the stride-1 3-tap 1-D convs of the NBt1D blocks
// (/root/reference/emsanet/model.py:47-58, args.py:158-164: conv3x1 / conv1x3 with C_in = C_out in
// {64, 128, 256, 512}) in 16-bit storage: forward and data gradient of BASELINE.json configs[2] / [4].
//
// What conv_h.hip's generic implicit GEMM does per 128x64 tile -- fetch the weight panel again,
// fetch the activation rows once per tap, 6-24 barrier-separated K steps, drain -- is a serial
// latency chain per workgroup (0.24 of the HBM roofline for three rounds).  This kernel
// turns the loop nest inside out:
//   * WEIGHTS STAY IN REGISTERS.  A workgroup is persistent (one or two per CU for the whole launch)
//     and each wave keeps its slice of the weights as ready-made MFMA B operands: 24 fragments of
//     v_mfma_f32_32x32x16 (3 taps x 128 input channels x 32 output channels, or 3 x 64 x 64 at
//     C = 64) = 96 VGPRs, loaded ONCE per launch with fully coalesced 1 KB loads from the
//     fragment-ordered weight image of emsa_pack_weight_frag_t.  Waves of a workgroup split the
//     output channels (WN) and, for C >= 256, the input channels (WK; partial sums meet in LDS).
//   * ACTIVATIONS STREAM THROUGH LDS ONCE.  A tile of output pixels is loaded with its halo by
//     LDS-DMA (buffer_load ... lds), double buffered across tiles, and all three taps read it at
//     shifted rows: no per-tap re-fetch, no K loop over global memory, one barrier pair per tile.
//     1x3: 32*TM*WM consecutive pixels + 2 halo pixels, the left / right image borders redirected to
//     a row of zeros; 3x1: a TH x TW patch + 2 halo rows, zero padding by out-of-range DMA offsets.
//   * per MFMA one ds_read_b128 of the A operand (two MFMAs at C = 64) and nothing else: the LDS
//     pipe runs at <= 50 % of the matrix pipe's issue time, A reads are software-pipelined by hand.
//   * same fused epilogue as emsa_conv_igemm_t (bias, BatchNorm batch statistics, folded BatchNorm,
//     residual, ReLU, ReLU-backward mask, fused BatchNorm-backward sums), taken from fp32 values in
//     an LDS stage so that every global access is 16 bytes per lane / full 128-byte lines.
// HBM traffic = input + output once; L2 -> LDS traffic = input x (C_out / channels per workgroup).
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>

#include "common.h"
#include "prof.h"

namespace {

typedef unsigned int ru32x4 __attribute__((ext_vector_type(4)));
typedef float rf32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rbf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rf16x8 __attribute__((ext_vector_type(8)));
constexpr uint32_t kROOB = 0x80000000u;

// EMSA_RS_DBG=1 (probe builds only, tools/jobs/r04_rs2.sh): wave 0 of every workgroup accumulates the
// shader-clock time of each phase of the tile loop; emsa_conv1d_rs_dbg_read returns the table
#ifndef EMSA_RS_DBG
#define EMSA_RS_DBG 0
#endif
#if EMSA_RS_DBG
__device__ long long g_rs_dbg[8 * 2048];
#define RS_MARK(ph)                                         \
  do {                                                      \
    const long long t_ = (long long)__builtin_readcyclecounter(); \
    dbg_t[ph] += t_ - dbg_prev;                             \
    dbg_prev = t_;                                          \
  } while (0)
#else
#define RS_MARK(ph) do {} while (0)
#endif

template <typename T> struct RVec8;
template <> struct RVec8<emsa_bf16> { typedef rbf16x8 type; };
template <> struct RVec8<emsa_f16> { typedef rf16x8 type; };

__device__ __forceinline__ f32x16 rmfma(rbf16x8 a, rbf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 rmfma(rf16x8 a, rf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

struct RFastDiv {
  uint32_t mul, shift, d;
};
inline RFastDiv r_make_fastdiv(uint32_t d) {
  RFastDiv f;
  f.d = d;
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  f.shift = s;
  f.mul = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  return f;
}
__device__ __forceinline__ uint32_t r_fast_div(uint32_t n, const RFastDiv& f) {
  return (__umulhi(n, f.mul) + n) >> f.shift;
}

struct ConvRSArgs {
  const void* in;
  const void* wf;
  void* out;
  const float* bias;
  float* stats;
  const float* scale;
  const float* shift;
  const void* residual;
  const void* mask_src;
  const float* bnb_mean;
  const float* bnb_invstd;
  float* bnb_out;
  int bnb_rows_alloc;
  int ld_res, ld_mask, act, ld_out;
  uint32_t in_bytes, wf_bytes, out_bytes, res_bytes, mask_bytes;
  int M, H, W, n_img;
  int px_bytes, row_bytes, img_bytes;  // input strides in bytes
  int sign;                 // tap t reads the pixel at sign * (t - 1) along the conv direction
  int n_ch;
  int tiles, tiles_w, twl, th;   // 3x1: TW = 1 << twl columns x th rows per tile
  int gx;                   // workgroups per XCD and channel slice
  int nslice;
  int ni;                   // DMA instructions (1 KB each) per tile
  int abuf;                 // bytes of one A buffer
  int stat_rows;            // rows of the statistics / BatchNorm-backward partial buffers
  int bacc_off;             // LDS offset of the BNB accumulators [16][threads]
  RFastDiv div_w, div_tpi, div_tw;
  // twin launch (emsa_conv1d_rs_pair_t, PAIR instantiations): the workgroups with blockIdx.y == 1 run
  // the SAME geometry on a second set of tensors (the other encoder / decoder of the model)
  const void* in2;
  const void* wf2;
  void* out2;
  const float* bias2;
  const float* scale2;
  const float* shift2;
  const void* residual2;
  // INBN (emsa_conv1d_rs_inbn_t): the conv runs on relu(in * in_scale[c] + in_shift[c]) -- the
  // BatchNorm + ReLU in front of it, formed where the staged tile goes from registers to LDS;
  // inaff_off = LDS offset of the [2][C_in] table
  const float* in_scale;
  const float* in_shift;
  int inaff_off;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t r_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

// One LDS-DMA instruction (64 lanes x 16 bytes -> 1 KB of LDS at `lds_addr`, lane-linear) as INLINE
// ASM: the compiler's wait-count pass treats a visible LDS-DMA as a pending write to all of LDS and
// puts s_waitcnt vmcnt(0) in front of later ds_reads (here: the first read of the output stage,
// right behind the barrier -- i.e. it drained the NEXT tile's DMA in every iteration).  Hidden from
// it, the DMA is ordered by this kernel's own counted waits; the compiler's counted waits for its
// own loads / stores stay correct (in-order vmcnt: extra older or younger operations only make
// them more conservative).
__device__ __forceinline__ void r_dma16(ru32x4 rsrc, uint32_t lds_addr, uint32_t voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
               :
               : "s"(lds_addr), "v"(voff), "s"(rsrc)
               : "memory");   // (M0 is used by nothing else in this kernel: gfx9 DS ops do not read it)
}

// bank swizzle of the A tile (rows of RB bytes written lane-linearly by the DMA): the 16-byte chunk
// c of row R lives at chunk c ^ swz(R).  ds_read_b128 serves 16 lanes (= 16 rows, one chunk index)
// per LDS cycle; they must hit 16 different 16-byte slots of the 256-byte bank row.
template <int CPR>
__device__ __forceinline__ int r_swz(int R) {
  return CPR == 8 ? ((R >> 1) & 7) : (R & 15);
}

// T storage type; KC = C_in; wave tile 32*TM pixels x 32*TN channels x (KC / WK) input channels;
// workgroup = WM x WN x WK waves; DIRH: taps along H (3x1) instead of W (1x3); BNB: the data
// gradient with the BatchNorm-backward sums of the layer in front (emsa_conv_igemm_bnb_t).
#ifndef EMSA_RS_DMA
#define EMSA_RS_DMA 0
#endif
// pixel tiles of the 256- / 512-channel kernels: 32 * TM pixels (A/B builds)
#ifndef EMSA_RS_TM256
#define EMSA_RS_TM256 1
#endif
#ifndef EMSA_RS_TM512
#define EMSA_RS_TM512 2
#endif
// EMSA_RS_W8: bit mask of channel counts (1 = 64, 2 = 128, 4 = 256) whose workgroups have EIGHT waves
// (two pixel sub-tiles per workgroup, one workgroup per CU) instead of four (two workgroups per CU):
// half as many workgroups fetch a weight slice at launch time
#ifndef EMSA_RS_W8
#define EMSA_RS_W8 0
#endif
#ifndef EMSA_RS_OCC
#define EMSA_RS_OCC 2
#endif
// Wave-tile variants (round 6 A/B builds): 1 = every wave owns TWO 32-channel output tiles (TN = 2) of
// a smaller K range -- one ds_read_b128 of the A operand then feeds two MFMAs (half the LDS read
// volume of the MFMA loop) at the price of a K split twice as deep (2x the fp32 stage).
//   V128: <TN 1, WN 4, WK 1, TM 2> (64-pixel tiles)  ->  <TN 2, WN 2, WK 2, TM 1> (32-pixel tiles)
//   V256: <TN 1, WN 2, WK 2>                         ->  <TN 2, WN 1, WK 4>
//   V512: <TN 1, WN 2, WK 4, TM 1|2>                 ->  <TN 2, WN 1, WK 8, TM 1>
#ifndef EMSA_RS_V128
#define EMSA_RS_V128 0
#endif
#ifndef EMSA_RS_V256
#define EMSA_RS_V256 0
#endif
#ifndef EMSA_RS_V512
#define EMSA_RS_V512 0
#endif
// EPI: the epilogue reads a residual and / or a mask tensor (as its own instantiation: the waits for
// those loads would otherwise sit in every launch's output pass and drain the next tile's DMA).
// INBN: the input tile is normalised + rectified on its way into LDS (the NBt1D block's bn1 folded
// into conv3x1_2, ref emsanet/model.py:47-58): per lane 8 fixed input channels, scale / shift from an
// LDS table, 8 fma + 8 max per 16-byte piece; pieces loaded out of range (zero padding above / below
// the image, pixels beyond the tile) stay zero -- the padding pads the NORMALISED tensor.
template <typename T, int KC, int TN, int WM, int WN, int WK, int TM, bool DIRH, bool BNB, bool EPI,
          bool PAIR = false, bool INBN = false>
__global__ __launch_bounds__(64 * WM * WN * WK, EMSA_RS_OCC) void conv_rs_kernel(const ConvRSArgs p_in) {
  static_assert(!INBN || (DIRH && !BNB && !EPI && !PAIR && EMSA_RS_DMA == 0),
                "INBN: the 3x1 forward conv on the register-staged loader");
  typedef typename RVec8<T>::type V8;
  // PAIR: two independent convs of one geometry in one launch (grid.y = 2); everything below sees
  // the tensors of its half through `p`
  ConvRSArgs p = p_in;
  if constexpr (PAIR) {
    if (blockIdx.y != 0) {
      p.in = p_in.in2; p.wf = p_in.wf2; p.out = p_in.out2; p.bias = p_in.bias2;
      p.scale = p_in.scale2; p.shift = p_in.shift2; p.residual = p_in.residual2;
    }
  }
  constexpr int NWV = WM * WN * WK, NT = 64 * NWV;
  constexpr int KS = KC / 16, KSW = KS / WK;          // k16 steps: all / per wave
  constexpr int RB = KC * 2, CPR = KC / 8, RPI = 64 / CPR;
  // register staging (default): the tile travels global -> VGPRs -> LDS, rows padded by 16 bytes
  // (conflict-free ds_read_b128 without a swizzle, k16 steps as immediate offsets), ONE A buffer;
  // EMSA_RS_DMA=1 builds the LDS-DMA loader (two unpadded, XOR-swizzled buffers) for A/B runs
  constexpr bool kDMA = EMSA_RS_DMA != 0;
  constexpr int RBP = kDMA ? RB : RB + 16;
  constexpr int NIWM = (KC >= 256 && TM == 2) ? 9 : 5;   // 16-byte loads per lane and tile, at most
  constexpr int BM = 32 * TM * WM, NWG = 32 * TN * WN;
  constexpr int SLD = NWG + 4;
  constexpr int NF = 3 * KSW;                         // A fragments per 32-pixel row tile
  static_assert(KS % WK == 0 && 64 % CPR == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

#if EMSA_RS_DBG
  long long dbg_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long dbg_prev = (long long)__builtin_readcyclecounter();
  const long long dbg_start = dbg_prev;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);

  // ---- persistent schedule: XCD x gets the tiles [x T/8, (x+1) T/8) so that neighbouring tiles
  // (halo rows) and the channel slices of one tile meet in one L2 ------------------------------
  const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
  const int slice = qb % p.nslice, idx = qb / p.nslice;
  const int tile_hi = (int)(((long)(xcd + 1) * p.tiles) >> 3);
  int tile = (int)(((long)xcd * p.tiles) >> 3) + idx;
  const int n_base = slice * NWG;                     // first output channel of this workgroup

  const int zoff = (kDMA ? 2 : 1) * p.abuf;           // row of zeros (1x3 borders)
  const int eoff = zoff + RB + 16;                    // per-channel epilogue vectors [3][NWG]
  const int soff = eoff + 3 * NWG * 4;                // fp32 stage [WK][BM][SLD]
  float* const stage = reinterpret_cast<float*>(smem + soff);
  float* const evec = reinterpret_cast<float*>(smem + eoff);
  // BNB: the per-thread BatchNorm-backward sums (2 x 8 channels) live in LDS, not in registers
  float* const bacc = reinterpret_cast<float*>(smem + p.bacc_off);

  if (!DIRH) {
    for (int i = tid; i < RB / 4; i += NT) reinterpret_cast<uint32_t*>(smem + zoff)[i] = 0u;
  }
  // folded-BatchNorm scale / shift (or the BatchNorm-backward affine + mean) of this workgroup's
  // channels: read from LDS in the output pass -- a global load there would wait for (in-order
  // vmcnt) the DMA of the next tile
  if (p.scale) {
    for (int i = tid; i < NWG; i += NT) {
      evec[i] = p.scale[n_base + i];
      evec[NWG + i] = p.shift[n_base + i];
      evec[2 * NWG + i] = BNB ? p.bnb_mean[n_base + i] : 0.f;
    }
  }

  if constexpr (INBN) {
    float* tb = reinterpret_cast<float*>(smem + p.inaff_off);
    for (int i = tid; i < KC; i += NT) {
      tb[i] = p.in_scale[i];
      tb[KC + i] = p.in_shift[i];
    }
  }

  const uint64_t in_base = (uint64_t)p.in;
  const ru32x4 rs_in = {(uint32_t)in_base, (uint32_t)(in_base >> 32) & 0xFFFFu, p.in_bytes, 0x00020000u};
  const __amdgpu_buffer_rsrc_t rs_inb = r_rsrc(p.in, p.in_bytes);
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>(smem);

  // ---- loader of one tile (+ halo): lane -> (row of the 1 KB piece, 16-byte chunk) -------------
  const int lrow = lane / CPR, pc = lane % CPR;
  auto tile_voff = [&](int tl, int qi) -> uint32_t {
    // byte offset of this lane's 16 bytes of piece qi of tile tl; out of range -> zeros
    const int R = qi * RPI + lrow;
    const int lc = kDMA ? (pc ^ r_swz<CPR>(R)) : pc;
    if constexpr (!DIRH) {
      return (uint32_t)((tl * BM - 1 + R) * p.px_bytes + lc * 16);   // < 0 or beyond the tensor: zeros
    } else {
      const int img = (int)r_fast_div((uint32_t)tl, p.div_tpi);
      const int rem = tl - img * (int)p.div_tpi.d;
      const int ty = (int)r_fast_div((uint32_t)rem, p.div_tw);
      const int h = ty * p.th - 1 + (R >> p.twl);
      const int w = ((rem - ty * p.tiles_w) << p.twl) + (R & ((1 << p.twl) - 1));
      const bool ok = (unsigned)h < (unsigned)p.H && w < p.W;
      return ok ? (uint32_t)(img * p.img_bytes + h * p.row_bytes + w * p.px_bytes + lc * 16) : kROOB;
    }
  };
  [[maybe_unused]] auto issue_dma = [&](int tl, int buf) {
    const int niw = (p.ni + NWV - 1) / NWV;
    for (int jj = 0; jj < niw; ++jj) {
      const int qi = wave + jj * NWV;                 // wave-uniform
      if (qi < p.ni)
        r_dma16(rs_in, (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + (uint32_t)(buf * p.abuf + qi * 1024))),
                tile_voff(tl, qi));
    }
  };
  ru32x4 areg[NIWM];
  [[maybe_unused]] uint32_t areal = 0u;               // INBN: bit jj = piece jj was loaded from a real pixel
  [[maybe_unused]] auto load_tile = [&](int tl) {
    if constexpr (INBN) areal = 0u;
#pragma unroll
    for (int jj = 0; jj < NIWM; ++jj) {
      const int qi = wave + jj * NWV;
      // (pieces beyond the tile: an out-of-range offset instead of a branch around the load)
      const uint32_t vo = qi < p.ni ? tile_voff(tl, qi) : kROOB;
      if constexpr (INBN) areal |= (vo != kROOB ? 1u : 0u) << jj;
      areg[jj] = __builtin_amdgcn_raw_buffer_load_b128(rs_inb, (int)vo, 0, 0);
    }
  };
  [[maybe_unused]] auto store_tile = [&]() {
    [[maybe_unused]] rf32x8 isc, ish;
    if constexpr (INBN) {
      const float* tb = reinterpret_cast<const float*>(smem + p.inaff_off) + pc * 8;
      const float4 s0 = emsa_ld4(tb), s1 = emsa_ld4(tb + 4), t0 = emsa_ld4(tb + KC), t1 = emsa_ld4(tb + KC + 4);
      isc = rf32x8{s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      ish = rf32x8{t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
    }
#pragma unroll
    for (int jj = 0; jj < NIWM; ++jj) {
      const int qi = wave + jj * NWV;
      if constexpr (INBN) {
        const rf32x8 v = __builtin_convertvector(__builtin_bit_cast(V8, areg[jj]), rf32x8);
        const bool real = (areal >> jj) & 1u;
        rf32x8 a;
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = real ? fmaxf(__builtin_fmaf(v[e], isc[e], ish[e]), 0.f) : 0.f;
        areg[jj] = __builtin_bit_cast(ru32x4, __builtin_convertvector(a, V8));
      }
      if (qi < p.ni)
        *reinterpret_cast<ru32x4*>(smem + (qi * RPI + lrow) * RBP + pc * 16) = areg[jj];
    }
  };

  // ---- output pass geometry (fixed per thread) -----------------------------------------------
  constexpr int C8 = NWG / 8, RPP = NT / C8;
  constexpr int PASSES = BM / RPP > 0 ? BM / RPP : 1;
  const int col8 = tid % C8, orow = tid / C8;
  const int n = n_base + col8 * 8;
  const bool has_affine = p.scale != nullptr && !BNB;
  // (registers where they fit: the 64- / 128-channel kernels keep two accumulator tiles per wave)
  constexpr bool kBaccLds = BNB && KC <= 128;
  rf32x8 b0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b1 = b0;
  if constexpr (kBaccLds) {
#pragma unroll
    for (int e = 0; e < 16; ++e) bacc[e * NT + tid] = 0.f;
  }
  const __amdgpu_buffer_rsrc_t rs_out = r_rsrc(p.out, p.out_bytes);
  const __amdgpu_buffer_rsrc_t rs_res = r_rsrc(p.residual ? p.residual : p.out, p.res_bytes);
  const __amdgpu_buffer_rsrc_t rs_msk = r_rsrc(p.mask_src ? p.mask_src : p.out, p.mask_bytes);
  const bool has_res = EPI && p.residual != nullptr, has_msk = EPI && p.mask_src != nullptr;
  // BatchNorm batch statistics in the row domain: per thread 8 channels, shifted sums about the
  // first value the thread sees (d = v - v0: no E[x^2] - mean^2 cancellation), merged per
  // workgroup with Chan's formula at the end
  const bool want_stats = !BNB && !EPI && p.stats != nullptr;   // (statistics: forward convs, no residual / mask)
  rf32x8 st_s = b0, st_d = b0, st_q = b0;
  float st_n = 0.f;
  float bias_v[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
    bias_v[j] = (p.bias && wk == 0) ? p.bias[n_base + (wn * TN + j) * 32 + l31] : 0.f;

  if constexpr (kDMA) issue_dma(tile, 0);
  else load_tile(tile);
  // ---- weights: 3 * KSW * TN fragments, resident for the whole launch ------------------------
  // (loaded behind the first tile's DMA; the empty asm pins every fragment as "arrived" here --
  //  otherwise the compiler's own waits for these loads sit in front of the first MFMAs INSIDE the
  //  tile loop, end in s_waitcnt vmcnt(0) and drain the next tile's DMA in every iteration)
  V8 bf[3][KSW][TN];
  {
    const __amdgpu_buffer_rsrc_t rs_w = r_rsrc(p.wf, p.wf_bytes);
    const int NB = p.n_ch >> 5;
    const int nb0 = (n_base >> 5) + wn * TN;
    ru32x4 raw[3][KSW][TN];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int kk = 0; kk < KSW; ++kk) {
          const int frag = ((t * NB + nb0 + j) * KS + wk * KSW + kk);
          raw[t][kk][j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, frag * 1024, 0);
        }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int kk = 0; kk < KSW; ++kk) {
          asm volatile("" : "+v"(raw[t][kk][j]));
          bf[t][kk][j] = __builtin_bit_cast(V8, raw[t][kk][j]);
        }
  }

  if constexpr (INBN) __syncthreads();                // the input-affine table is in LDS
  if constexpr (!kDMA) store_tile();
  RS_MARK(0);
  int it = 0;
  for (; tile < tile_hi; tile += p.gx, ++it) {
    const int buf = kDMA ? (it & 1) : 0;
    const bool more = tile + p.gx < tile_hi;
    if constexpr (kDMA) {
      // tile `it` has landed (every VMEM op younger than its DMA is one of the PASSES stores of the
      // previous output pass: in-order completion on gfx9), is visible to every wave behind the
      // barrier, and the stage and the other buffer are free
      if (it == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PASSES) : "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's pieces of the tile are in LDS
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    RS_MARK(1);
    // the next tile starts its trip now: by LDS-DMA into the other buffer, or into registers that
    // are written to the (single) A buffer behind this tile's output pass
    if (more) {
      if constexpr (kDMA) issue_dma(tile + p.gx, buf ^ 1);
      else load_tile(tile + p.gx);
    }

    // ---- A operand addresses of this tile ------------------------------------------------------
    int m0 = 0, h0 = 0, w0 = 0, img_pix = 0;
    if constexpr (!DIRH) {
      m0 = tile * BM;
    } else {
      const int img = (int)r_fast_div((uint32_t)tile, p.div_tpi);
      const int rem = tile - img * (int)p.div_tpi.d;
      const int ty = (int)r_fast_div((uint32_t)rem, p.div_tw);
      h0 = ty * p.th;
      w0 = (rem - ty * p.tiles_w) << p.twl;
      img_pix = img * p.H * p.W;
    }
    int abase[TM][3];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = (wm * TM + i) * 32 + l31;
      int wq = 0;
      if constexpr (!DIRH) {
        const uint32_t m = (uint32_t)(m0 + r);
        wq = (int)(m - r_fast_div(m, p.div_w) * p.div_w.d);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int dt = p.sign * (t - 1);
        int R, ok = 1;
        if constexpr (!DIRH) {
          R = r + 1 + dt;
          ok = (unsigned)(wq + dt) < (unsigned)p.W;
        } else {
          R = r + ((1 + dt) << p.twl);
        }
        int a;
        if constexpr (kDMA) {
          const int s = r_swz<CPR>(R);
          a = buf * p.abuf + R * RB + (((lh ^ (s & 1)) << 4) | ((s >> 1) << 5)) + (wk * KSW * 32);
        } else {
          a = R * RBP + (lh << 4) + (wk * KSW * 32);   // (wk * KSW) << 5: this wave's K range
        }
        abase[i][t] = ok ? a : zoff + (lh << 4);
      }
    }

    RS_MARK(2);
    // ---- MFMAs: per (row tile, tap, k16 step) one ds_read_b128, TN MFMAs; the reads run PD
    // fragments ahead of their use ----------------------------------------------------------------
    constexpr int NACC = (TM * TN == 1) ? 2 : 1;       // split the dependent chain of a lone tile
    f32x16 acc[TM][TN][NACC];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][a][r] = 0.f;
    {
      // fragment order (tap, k16 step, row tile): consecutive MFMAs hit different accumulators
      constexpr int TOT = TM * NF, PD = BNB ? 2 : 4;
      V8 af[PD];
      auto a_addr = [&](int f) {
        const int i = f % TM, t = (f / TM) / KSW, kk = (f / TM) % KSW;
        return kDMA ? (abase[i][t] ^ (kk << 5)) : (abase[i][t] + (kk << 5));
      };
#pragma unroll
      for (int f = 0; f < PD && f < TOT; ++f)
        af[f] = *reinterpret_cast<const V8*>(smem + a_addr(f));
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int f = 0; f < TOT; ++f) {
        const int i = f % TM, t = (f / TM) / KSW, kk = (f / TM) % KSW;
        const V8 cur = af[f % PD];
        if (f + PD < TOT) af[f % PD] = *reinterpret_cast<const V8*>(smem + a_addr(f + PD));
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j][kk % NACC] = rmfma(cur, bf[t][kk][j], acc[i][j][kk % NACC]);
      }
      __builtin_amdgcn_s_setprio(0);
    }

    RS_MARK(3);
    // ---- accumulators (+ bias) -> fp32 stage [wk][row][channel] ----------------------------------
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          float v = acc[i][j][0][r] + bias_v[j];
          if constexpr (NACC == 2) v += acc[i][j][1][r];
          stage[(wk * BM + row) * SLD + (wn * TN + j) * 32 + l31] = v;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    RS_MARK(4);
    if constexpr (!kDMA && BNB) {
      // (the fused BatchNorm-backward epilogue needs the staging registers: the next tile goes to
      //  LDS before the output pass instead of behind it -- the A buffer is free from here on)
      if (more) store_tile();
    }

    // ---- output pass: 8 channels x PASSES rows per thread, every global access 16 bytes ----------
    constexpr int PG = BNB ? 1 : (PASSES > 2 ? 2 : PASSES);   // passes per group (register budget)
#pragma unroll
    for (int pg = 0; pg < PASSES; pg += PG) {
      uint32_t ooff[PG];
      rf32x8 v[PG];
      ru32x4 rres[PG], rmsk[PG];
#pragma unroll
      for (int q = 0; q < PG; ++q) {
        const int row = (pg + q) * RPP + orow;
        bool ok = row < BM;
        int pix;
        if constexpr (!DIRH) {
          pix = m0 + row;
          ok = ok && pix < p.M;
        } else {
          const int hh = h0 + (row >> p.twl), ww = w0 + (row & ((1 << p.twl) - 1));
          ok = ok && hh < p.H && ww < p.W;
          pix = img_pix + hh * p.W + ww;
        }
        ooff[q] = ok ? (uint32_t)pix : kROOB;
        if (has_res)
          rres[q] = __builtin_amdgcn_raw_buffer_load_b128(
              rs_res, ok ? (pix * p.ld_res + n) * 2 : (int)kROOB, 0, 0);
        if (has_msk)
          rmsk[q] = __builtin_amdgcn_raw_buffer_load_b128(
              rs_msk, ok ? (pix * p.ld_mask + n) * 2 : (int)kROOB, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < PG; ++q) {
        const int row = (pg + q) * RPP + orow;
        const float* sp = stage + (row < BM ? row : 0) * SLD + col8 * 8;
        float4 v0 = emsa_ld4(sp), v1 = emsa_ld4(sp + 4);
#pragma unroll
        for (int k = 1; k < WK; ++k) {
          const float4 u0 = emsa_ld4(sp + k * BM * SLD), u1 = emsa_ld4(sp + k * BM * SLD + 4);
          v0.x += u0.x; v0.y += u0.y; v0.z += u0.z; v0.w += u0.w;
          v1.x += u1.x; v1.y += u1.y; v1.z += u1.z; v1.w += u1.w;
        }
        v[q] = rf32x8{v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      }
#pragma unroll
      for (int q = 0; q < PG; ++q) {
        const bool ok = ooff[q] != kROOB;
        rf32x8 x = v[q];
        if (want_stats && ok) {
          if (st_n == 0.f) st_s = x;
          st_n += 1.f;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float d = x[e] - st_s[e];
            st_d[e] += d;
            st_q[e] += d * d;
          }
        }
        if (has_affine) {
          const float4 sc0 = emsa_ld4(evec + col8 * 8), sc1 = emsa_ld4(evec + col8 * 8 + 4);
          const float4 sh0 = emsa_ld4(evec + NWG + col8 * 8), sh1 = emsa_ld4(evec + NWG + col8 * 8 + 4);
          x = rf32x8{x[0] * sc0.x + sh0.x, x[1] * sc0.y + sh0.y, x[2] * sc0.z + sh0.z,
                     x[3] * sc0.w + sh0.w, x[4] * sc1.x + sh1.x, x[5] * sc1.y + sh1.y,
                     x[6] * sc1.z + sh1.z, x[7] * sc1.w + sh1.w};
        }
        if (has_res) x += __builtin_convertvector(__builtin_bit_cast(V8, rres[q]), rf32x8);
        if constexpr (BNB) {
          const rf32x8 tt = __builtin_convertvector(__builtin_bit_cast(V8, rmsk[q]), rf32x8);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            x[e] = (ok && (tt[e] * evec[col8 * 8 + e] + evec[NWG + col8 * 8 + e]) > 0.f) ? x[e] : 0.f;
            const float xm = x[e] * (tt[e] - evec[2 * NWG + col8 * 8 + e]);
            if constexpr (kBaccLds) {
              bacc[e * NT + tid] += x[e];
              bacc[(8 + e) * NT + tid] += xm;
            } else {
              b0[e] += x[e];
              b1[e] += xm;
            }
          }
        } else if (has_msk) {
          const rf32x8 mm = __builtin_convertvector(__builtin_bit_cast(V8, rmsk[q]), rf32x8);
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = mm[e] > 0.f ? x[e] : 0.f;
        }
        if (p.act == EMSA_ACT_RELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) x[e] = fmaxf(x[e], 0.f);
        }
        const V8 o = __builtin_convertvector(x, V8);
        __builtin_amdgcn_raw_buffer_store_b128(
            __builtin_bit_cast(ru32x4, o), rs_out,
            ok ? (int)((ooff[q] * (uint32_t)p.ld_out + (uint32_t)n) * 2u) : (int)kROOB, 0, 0);
      }
    }
  

    if constexpr (!kDMA && !BNB) {
      if (more) store_tile();      // (every wave is behind the barrier that ended the reads of this tile)
    }
#if EMSA_RS_DBG
    RS_MARK(5);
    dbg_t[7] += 1;
#endif
  }
  {
#if EMSA_RS_DBG
    if (tid == 0 && blockIdx.x < 2048) {
      dbg_t[6] = (long long)__builtin_readcyclecounter() - dbg_start;
      for (int k = 0; k < 8; ++k) g_rs_dbg[blockIdx.x * 8 + k] = dbg_t[k];
    }
#endif
  }

  // ---- per-workgroup partial rows: statistics (sum, M2, count) / BatchNorm-backward sums ---------
  if (want_stats || BNB) {
    const int srow = xcd * p.gx + idx;                 // one row per workgroup of a channel slice
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* red = stage;                                // [RPP][3 or 2][NWG]
    if constexpr (BNB) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(orow * 2 + 0) * NWG + col8 * 8 + e] = kBaccLds ? bacc[e * NT + tid] : b0[e];
        red[(orow * 2 + 1) * NWG + col8 * 8 + e] = kBaccLds ? bacc[(8 + e) * NT + tid] : b1[e];
      }
    } else {
      // (n, mean, M2) of this thread's rows
      const float inv = st_n > 0.f ? 1.f / st_n : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(orow * 3 + 0) * NWG + col8 * 8 + e] = st_s[e] + st_d[e] * inv;
        red[(orow * 3 + 1) * NWG + col8 * 8 + e] = fmaxf(st_q[e] - st_d[e] * st_d[e] * inv, 0.f);
      }
      if (col8 == 0) red[RPP * 3 * NWG + orow] = st_n;
    }
    __syncthreads();
    for (int c = tid; c < NWG; c += NT) {
      const int nn = n_base + c;
      if constexpr (BNB) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll 1
        for (int r = 0; r < RPP; ++r) {
          a0 += red[(r * 2 + 0) * NWG + c];
          a1 += red[(r * 2 + 1) * NWG + c];
        }
        p.bnb_out[((size_t)0 * p.bnb_rows_alloc + srow) * p.n_ch + nn] = a0;
        p.bnb_out[((size_t)1 * p.bnb_rows_alloc + srow) * p.n_ch + nn] = a1 * p.bnb_invstd[nn];
      } else {
        float na = 0.f, ma = 0.f, qa = 0.f;            // Chan et al.: merge (n, mean, M2) pairs
#pragma unroll 1
        for (int r = 0; r < RPP; ++r) {
          const float nb = red[RPP * 3 * NWG + r];
          if (nb > 0.f) {
            const float mb = red[(r * 3 + 0) * NWG + c], q2 = red[(r * 3 + 1) * NWG + c];
            const float nn2 = na + nb, dl = mb - ma;
            qa += q2 + dl * dl * (na * nb / nn2);
            ma += dl * (nb / nn2);
            na = nn2;
          }
        }
        p.stats[((size_t)0 * p.stat_rows + srow) * p.n_ch + nn] = ma * na;
        p.stats[((size_t)1 * p.stat_rows + srow) * p.n_ch + nn] = qa;
        p.stats[((size_t)2 * p.stat_rows + srow) * p.n_ch + nn] = na;
      }
    }
  }
}

// ---- host side ----------------------------------------------------------------------------------
struct RSPlan {
  int kc = 0;         // 64 / 128 / 256 / 512
  int niwm = 5;       // staging registers (16-byte pieces per lane and tile)
  int tm = 1;         // 32-pixel row tiles per wave (256 / 512 channels)
  bool dirh = false;
  int bm = 0, nwg = 0, nwv = 0, wk = 1, rpi = 0;
  int tiles = 0, tiles_w = 0, twl = 0, th = 0;
  int ni = 0, abuf = 0, lds = 0, bacc_off = 0, gx = 0, nslice = 0, sign = 1;
};

// CUs the persistent grid is planned for: the current device's count (cached per device; several
// devices / threads in one process are fine -- ADVICE r4), capped by the CU budget (EMSA_RS_CUS or
// emsa_conv_rs_set_cu_budget: room for co-resident collective kernels, see functional.py)
constexpr int kMaxDev = 64;
std::atomic<int> g_dev_cus[kMaxDev];
std::atomic<int> g_cu_budget{-1};       // -1: not initialised (EMSA_RS_CUS read on first use), 0: none

int rs_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
  return dev;
}

int rs_cu_count() {
  const int dev = rs_device();
  int cus = g_dev_cus[dev].load(std::memory_order_relaxed);
  if (!cus) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
    if (cus <= 0) cus = 256;
    g_dev_cus[dev].store(cus, std::memory_order_relaxed);
  }
  int budget = g_cu_budget.load(std::memory_order_relaxed);
  if (budget < 0) {
    const char* e = getenv("EMSA_RS_CUS");
    budget = e ? atoi(e) : 0;
    if (budget < 0) budget = 0;
    g_cu_budget.store(budget, std::memory_order_relaxed);
  }
  if (budget > 0 && budget < cus) cus = budget < 8 ? 8 : budget;
  return cus;
}

// EMSA_CONV_RS=0 switches the kernel off (the generic implicit GEMM of conv_h.hip runs instead)
bool rs_enabled() {
  const char* e = getenv("EMSA_CONV_RS");
  return !(e && e[0] == '0');
}

bool rs_plan(const EmsaConvGeom* g, RSPlan& pl) {
  if (!g) return false;
  if (g->k_ch != g->n_ch) return false;
  if (g->k_ch != 64 && g->k_ch != 128 && g->k_ch != 256 && g->k_ch != 512) return false;
  const bool w3 = g->kh == 1 && g->kw == 3, h3 = g->kh == 3 && g->kw == 1;
  if (!w3 && !h3) return false;
  if (g->in_h != g->out_h || g->in_w != g->out_w) return false;
  if (g->mul_h != 1 || g->mul_w != 1 || g->div_h != 1 || g->div_w != 1) return false;
  if (g->out_pix_img || g->out_pix_row || g->out_pix_px || g->out_pix_off) return false;
  // tap t reads pixel + off + t * step along the conv direction: forward off = -1, step = 1;
  // data gradient off = +1, step = -1
  const int off = w3 ? g->off_w : g->off_h, step = w3 ? g->step_w : g->step_h;
  if (!((off == -1 && step == 1) || (off == 1 && step == -1))) return false;
  if ((w3 ? g->off_h : g->off_w) != 0) return false;
  if ((g->in_px_stride & 7) || (g->in_row_stride & 7) || (g->in_img_stride & 7) || (g->ld_out & 7))
    return false;
  if (g->in_px_stride < g->k_ch || g->ld_out < g->n_ch) return false;
  const long M = (long)g->n_img * g->out_h * g->out_w;
  const long in_bytes = (long)g->n_img * g->in_img_stride * 2;
  if (in_bytes >= (1L << 31) - (1L << 22) || M * g->ld_out * 2 >= (1L << 31) - (1L << 22)) return false;
  pl.kc = g->k_ch;
  pl.dirh = h3;
  pl.sign = step;
  pl.rpi = 1024 / (pl.kc * 2);
  switch (pl.kc) {
    case 64: pl.bm = (EMSA_RS_W8 & 1) ? 256 : 128; pl.nwg = 64; pl.nwv = (EMSA_RS_W8 & 1) ? 8 : 4; pl.wk = 1; break;
    case 128:
      if (EMSA_RS_V128) { pl.bm = 32; pl.nwg = 128; pl.nwv = 4; pl.wk = 2; break; }
      pl.bm = (EMSA_RS_W8 & 2) ? 128 : 64; pl.nwg = 128; pl.nwv = (EMSA_RS_W8 & 2) ? 8 : 4; pl.wk = 1; break;
    case 256: pl.tm = EMSA_RS_TM256; pl.nwg = 64; pl.nwv = (EMSA_RS_W8 & 4) ? 8 : 4; pl.wk = EMSA_RS_V256 ? 4 : 2; break;
    default:
      // 64-pixel tiles at 512 channels once there are enough pixels to give every workgroup a
      // few of them; small maps (batch-1 inference: 300 pixels at /32) keep 32-pixel tiles
      pl.tm = (EMSA_RS_TM512 == 2 && M >= 4096 && !EMSA_RS_V512) ? 2 : 1; pl.nwg = 64; pl.nwv = 8;
      pl.wk = EMSA_RS_V512 ? 8 : 4; break;
  }
  if (pl.kc >= 256) {
    pl.bm = 32 * pl.tm * ((pl.kc == 256 && (EMSA_RS_W8 & 4)) ? 2 : 1);
    pl.niwm = pl.tm == 2 ? 9 : 5;
  }
  pl.nslice = g->n_ch / pl.nwg;
  int rs;
  if (!h3) {
    // flattened pixels: needs the dense NHWC pixel grid
    if (g->in_row_stride != (long)g->in_w * g->in_px_stride ||
        g->in_img_stride != (long)g->in_h * g->in_row_stride)
      return false;
    pl.tiles = (int)((M + pl.bm - 1) / pl.bm);
    rs = pl.bm + 2;
  } else {
    // TH x TW patches, TW a power of two >= the DMA granule: least padded pixels, then widest
    long best = -1;
    for (int twl = 0; (1 << twl) <= 32 && (1 << twl) <= pl.bm; ++twl) {
      const int tw = 1 << twl, th = pl.bm / tw;
      if (tw < pl.rpi) continue;
      if ((pl.bm + 2 * tw + pl.rpi - 1) / pl.rpi > pl.niwm * pl.nwv) continue;   // staging registers
      const long tws = (g->out_w + tw - 1) / tw, ths = (g->out_h + th - 1) / th;
      const long cost = tws * ths * (long)(th + 2) * tw;       // pixels loaded per image
      if (best < 0 || cost <= best) {
        best = cost;
        pl.twl = twl;
        pl.th = th;
        pl.tiles_w = (int)tws;
        pl.tiles = (int)(tws * ths * g->n_img);
      }
    }
    if (best < 0) return false;
    rs = pl.bm + 2 * (1 << pl.twl);
  }
  pl.ni = (rs + pl.rpi - 1) / pl.rpi;
  if (pl.ni > pl.niwm * pl.nwv) return false;         // staging registers per wave
  pl.abuf = EMSA_RS_DMA ? pl.ni * 1024 : (pl.ni * pl.rpi * (pl.kc * 2 + 16) + 1023) / 1024 * 1024;
  const int rpp = pl.nwv * 64 / (pl.nwg / 8);
  const int stage = pl.wk * pl.bm * (pl.nwg + 4) * 4;
  const int red = rpp * 3 * pl.nwg * 4 + rpp * 4;
  pl.bacc_off = (EMSA_RS_DMA ? 2 : 1) * pl.abuf + pl.kc * 2 + 16 + 3 * pl.nwg * 4 + (stage > red ? stage : red);
  pl.lds = pl.bacc_off + (pl.kc <= 128 ? 16 * pl.nwv * 64 * 4 : 0);   // (+ LDS accumulators of the BNB form)
  if (pl.lds > 160 * 1024) return false;
  // persistent grid: 8 XCDs x gx workgroups x channel slices; every workgroup gets >= 1 tile
  // EMSA_RS_PER_CU=1: one workgroup per CU even where two fit (A/B: two launches on two streams
  // then share the CUs instead of queueing behind each other)
  static const int per_cu_max = [] {
    const char* e = getenv("EMSA_RS_PER_CU");
    return (e && e[0] == '1') ? 1 : 2;
  }();
  const int per_cu = (pl.nwv == 8 || 2 * pl.lds > 160 * 1024) ? 1 : per_cu_max;
  // (fewer tiles than workgroups: the surplus workgroups find no tile, write empty partial rows
  //  and leave -- batch-1 inference at /16 and /32)
  int gx = rs_cu_count() * per_cu / (8 * pl.nslice);
  if (gx > (pl.tiles + 7) / 8) gx = (pl.tiles + 7) / 8;
  if (gx < 1 || pl.tiles < 1) return false;
  pl.gx = gx;
  return true;
}

template <typename T, int KC, int TN, int WM, int WN, int WK, int TM>
int rs_launch(const ConvRSArgs& a, const RSPlan& pl, bool bnb, hipStream_t st) {
  const bool pair = a.in2 != nullptr;
  const bool inbn = a.in_scale != nullptr;
  const dim3 grid(8 * pl.gx * pl.nslice, pair ? 2 : 1), block(64 * WM * WN * WK);
  const bool epi = a.residual != nullptr || a.mask_src != nullptr;
  auto go = [&](void (*kern)(const ConvRSArgs)) {
    // more than 64 KB of dynamic LDS has to be asked for, once per kernel AND device (the
    // cache was process-wide and unguarded -- a second GPU or thread could launch without it)
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> seen;
    const std::pair<int, const void*> key(rs_device(), (const void*)kern);
    {
      std::lock_guard<std::mutex> lk(mu);
      if (seen.insert(key).second)
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(kern, grid, block, pl.lds, st, a);
  };
  // (the fused BatchNorm-backward form exists for bf16 only: fp16 is an inference storage type)
  constexpr bool kTrain = sizeof(T) == 2 && !__is_same(T, emsa_f16);
  if (bnb && !kTrain) return EMSA_E_ARG;
  if (inbn) {
    // the folded input BatchNorm: 3x1 forward conv, plain epilogue (bias, statistics, ReLU), bf16
    if (pair || bnb || epi || !pl.dirh || EMSA_RS_DMA != 0) return EMSA_E_SHAPE;
    if constexpr (kTrain && EMSA_RS_DMA == 0) {
      go(conv_rs_kernel<T, KC, TN, WM, WN, WK, TM, true, false, false, false, true>);
      return emsa_launch_status();
    } else {
      return EMSA_E_SHAPE;
    }
  }
  if (pair) {
    // twin launch: forward epilogues only (bias, folded BatchNorm, residual, ReLU)
    if (bnb || a.mask_src || a.stats) return EMSA_E_ARG;
    if (pl.dirh) {
      if (epi) go(conv_rs_kernel<T, KC, TN, WM, WN, WK, TM, true, false, true, true>);
      else go(conv_rs_kernel<T, KC, TN, WM, WN, WK, TM, true, false, false, true>);
    } else {
      if (epi) go(conv_rs_kernel<T, KC, TN, WM, WN, WK, TM, false, false, true, true>);
      else go(conv_rs_kernel<T, KC, TN, WM, WN, WK, TM, false, false, false, true>);
    }
    return emsa_launch_status();
  }
  if (pl.dirh) {
    if (bnb) {
      if constexpr (kTrain) go(conv_rs_kernel<T, KC, TN, WM, WN, WK, TM, true, true, true>);
    } else if (epi) go(conv_rs_kernel<T, KC, TN, WM, WN, WK, TM, true, false, true>);
    else go(conv_rs_kernel<T, KC, TN, WM, WN, WK, TM, true, false, false>);
  } else {
    if (bnb) {
      if constexpr (kTrain) go(conv_rs_kernel<T, KC, TN, WM, WN, WK, TM, false, true, true>);
    } else if (epi) go(conv_rs_kernel<T, KC, TN, WM, WN, WK, TM, false, false, true>);
    else go(conv_rs_kernel<T, KC, TN, WM, WN, WK, TM, false, false, false>);
  }
  return emsa_launch_status();
}

template <typename T>
int rs_dispatch(const ConvRSArgs& a, const RSPlan& pl, bool bnb, hipStream_t st) {
  switch (pl.kc) {
    case 64: return rs_launch<T, 64, 2, (EMSA_RS_W8 & 1) ? 8 : 4, 1, 1, 1>(a, pl, bnb, st);
#if EMSA_RS_V128
    case 128: return rs_launch<T, 128, 2, 1, 2, 2, 1>(a, pl, bnb, st);
#else
    case 128: return rs_launch<T, 128, 1, (EMSA_RS_W8 & 2) ? 2 : 1, 4, 1, 2>(a, pl, bnb, st);
#endif
#if EMSA_RS_V256
    case 256: return rs_launch<T, 256, 2, (EMSA_RS_W8 & 4) ? 2 : 1, 1, 4, EMSA_RS_TM256>(a, pl, bnb, st);
#else
    case 256: return rs_launch<T, 256, 1, (EMSA_RS_W8 & 4) ? 2 : 1, 2, 2, EMSA_RS_TM256>(a, pl, bnb, st);
#endif
    default:
#if EMSA_RS_V512
      return rs_launch<T, 512, 2, 1, 1, 8, 1>(a, pl, bnb, st);
#else
      if (pl.tm == 2) return rs_launch<T, 512, 1, 1, 2, 4, 2>(a, pl, bnb, st);
      return rs_launch<T, 512, 1, 1, 2, 4, 1>(a, pl, bnb, st);
#endif
  }
}

int conv_rs_impl(int32_t dtype, const EmsaConvGeom* g, const void* in, const void* wf, void* out,
                 const float* bias, float* stats, const float* scale, const float* shift,
                 const void* residual, int32_t ld_res, const void* mask_src, int32_t ld_mask,
                 int32_t act, const float* bnb_mean, const float* bnb_invstd, float* bnb_out,
                 int32_t bnb_rows_alloc, void* stream, const ConvRSArgs* twin = nullptr,
                 const float* in_scale = nullptr, const float* in_shift = nullptr) {
  if (dtype != EMSA_DT_BF16 && dtype != EMSA_DT_F16) return EMSA_E_ARG;
  RSPlan pl;
  if (!rs_plan(g, pl)) return EMSA_E_SHAPE;
  int inaff_off = 0;
  if (in_scale) {
    if (!in_shift || twin || !pl.dirh || dtype != EMSA_DT_BF16) return EMSA_E_SHAPE;
    if ((((uintptr_t)in_scale) | ((uintptr_t)in_shift)) & 3) return EMSA_E_SHAPE;
    inaff_off = (pl.lds + 15) / 16 * 16;
    pl.lds = inaff_off + 2 * pl.kc * 4;
    if (pl.lds > 160 * 1024) return EMSA_E_SHAPE;
  }
  if (twin) {
    // both halves resident at once: half the persistent workgroups per half where the full grid
    // would not fit twice
    const int per_cu = (pl.nwv == 8 || 2 * pl.lds > 160 * 1024) ? 1 : 2;
    int gx2 = rs_cu_count() * per_cu / (16 * pl.nslice);
    if (gx2 < 1) gx2 = 1;
    if (pl.gx > gx2) pl.gx = gx2;
  }
  if (!in || !wf || !out) return EMSA_E_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return EMSA_E_ARG;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (!al16(in) || !al16(wf) || !al16(out) || !al16(bias) || !al16(scale) || !al16(shift) ||
      (residual && (!al16(residual) || (ld_res & 7))) || (mask_src && (!al16(mask_src) || (ld_mask & 7))))
    return EMSA_E_SHAPE;
  const long M = (long)g->n_img * g->out_h * g->out_w;
  if ((residual && M * ld_res * 2 >= (1L << 31)) || (mask_src && M * ld_mask * 2 >= (1L << 31)))
    return EMSA_E_SHAPE;
  if (stats && (residual || mask_src)) return EMSA_E_SHAPE;   // no such launch in the model: conv_h takes it
  const bool bnb = bnb_out != nullptr;
  if (bnb && (!mask_src || !scale || !bnb_mean || !bnb_invstd || stats || act != EMSA_ACT_NONE ||
              bnb_rows_alloc < 8 * pl.gx))
    return EMSA_E_ARG;
  ConvRSArgs a;
  a.in = in; a.wf = wf; a.out = out; a.bias = bias; a.stats = stats; a.scale = scale; a.shift = shift;
  a.residual = residual; a.mask_src = mask_src;
  a.bnb_mean = bnb_mean; a.bnb_invstd = bnb_invstd; a.bnb_out = bnb_out; a.bnb_rows_alloc = bnb_rows_alloc;
  a.ld_res = ld_res; a.ld_mask = ld_mask; a.act = act; a.ld_out = g->ld_out;
  a.in_bytes = (uint32_t)((long)g->n_img * g->in_img_stride * 2);
  a.wf_bytes = (uint32_t)(3L * g->n_ch * g->k_ch * 2);
  a.out_bytes = (uint32_t)(M * g->ld_out * 2);
  a.res_bytes = residual ? (uint32_t)(M * ld_res * 2) : a.out_bytes;
  a.mask_bytes = mask_src ? (uint32_t)(M * ld_mask * 2) : a.out_bytes;
  a.M = (int)M; a.H = g->out_h; a.W = g->out_w; a.n_img = g->n_img;
  a.px_bytes = g->in_px_stride * 2; a.row_bytes = (int)(g->in_row_stride * 2);
  a.img_bytes = (int)(g->in_img_stride * 2);
  a.sign = pl.sign; a.n_ch = g->n_ch;
  a.tiles = pl.tiles; a.tiles_w = pl.tiles_w > 0 ? pl.tiles_w : 1; a.twl = pl.twl; a.th = pl.th;
  a.gx = pl.gx; a.nslice = pl.nslice; a.ni = pl.ni; a.abuf = pl.abuf; a.stat_rows = 8 * pl.gx;
  a.bacc_off = pl.bacc_off;
  a.in2 = nullptr; a.wf2 = nullptr; a.out2 = nullptr; a.bias2 = nullptr; a.scale2 = nullptr;
  a.shift2 = nullptr; a.residual2 = nullptr;
  a.in_scale = in_scale; a.in_shift = in_shift; a.inaff_off = inaff_off;
  if (twin) {
    a.in2 = twin->in2; a.wf2 = twin->wf2; a.out2 = twin->out2; a.bias2 = twin->bias2;
    a.scale2 = twin->scale2; a.shift2 = twin->shift2; a.residual2 = twin->residual2;
  }
  a.div_w = r_make_fastdiv((uint32_t)g->out_w);
  const int tpi = pl.dirh ? pl.tiles / g->n_img : 1;
  a.div_tpi = r_make_fastdiv((uint32_t)(tpi > 0 ? tpi : 1));
  a.div_tw = r_make_fastdiv((uint32_t)a.tiles_w);
  hipStream_t st = (hipStream_t)stream;
  const double flops = 2.0 * M * g->k_ch * g->n_ch * 3.0;
  const double px = (double)M * g->n_ch * 2.0;
  const double bytes = 2.0 * px + 3.0 * g->n_ch * g->k_ch * 2.0 + (residual ? px : 0.0) + (mask_src ? px : 0.0);
  const int ps = emsa_prof_begin(kProfClassConvRS, twin ? 2.0 * flops : flops, st, twin ? 2.0 * bytes : bytes);
  const int rc = dtype == EMSA_DT_BF16 ? rs_dispatch<emsa_bf16>(a, pl, bnb, st)
                                       : rs_dispatch<emsa_f16>(a, pl, bnb, st);
  emsa_prof_end(ps, st);
  return rc;
}

// OIHW fp32 parameter [cout][cin][3] -> the fragment-ordered 16-bit operand of conv_rs_kernel:
//   wf[tap][n / 32][k / 16][lane = n % 32 + 32 * ((k % 16) / 8)][k % 8]
// (one B operand of v_mfma_f32_32x32x16 = 1 KB contiguous); forward: (n, k) = (cout, cin), data
// gradient: (n, k) = (cin, cout).
template <typename T>
__global__ void pack_frag_kernel(const float* __restrict__ w, T* __restrict__ fwd,
                                 T* __restrict__ dgr, int cout, int cin) {
  const long total = 3L * cout * cin;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), ln = (int)((i >> 3) & 63);
    const long fr = i >> 9;
    if (fwd) {
      const int ks = (int)(fr % (cin / 16)), nb = (int)((fr / (cin / 16)) % (cout / 32)),
                t = (int)(fr / ((long)(cin / 16) * (cout / 32)));
      const int nn = nb * 32 + (ln & 31), k = ks * 16 + (ln >> 5) * 8 + e;
      fwd[i] = (T)w[((long)nn * cin + k) * 3 + t];
    }
    if (dgr) {
      const int ks = (int)(fr % (cout / 16)), nb = (int)((fr / (cout / 16)) % (cin / 32)),
                t = (int)(fr / ((long)(cout / 16) * (cin / 32)));
      const int nn = nb * 32 + (ln & 31), k = ks * 16 + (ln >> 5) * 8 + e;
      dgr[i] = (T)w[((long)k * cin + nn) * 3 + t];
    }
  }
}

}  // namespace

#if EMSA_RS_DBG
extern "C" int emsa_conv1d_rs_dbg_read(long long* host, int n_entries) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rs_dbg), (size_t)n_entries * 8) == hipSuccess ? 0 : -3;
}
#endif

// cus <= 0: no budget (all CUs of the device).  Returns the CU count conv_rs now plans with.  The
// statistics-row count of a launch depends on it: re-query emsa_conv1d_rs_stats_rows afterwards.
extern "C" int emsa_conv_rs_set_cu_budget(int32_t cus) {
  g_cu_budget.store(cus > 0 ? cus : 0, std::memory_order_relaxed);
  return rs_cu_count();
}

extern "C" int emsa_conv1d_rs_supported(int32_t dtype, const EmsaConvGeom* g) {
  if (dtype != EMSA_DT_BF16 && dtype != EMSA_DT_F16) return 0;
  RSPlan pl;
  return rs_enabled() && rs_plan(g, pl) ? 1 : 0;
}

extern "C" int emsa_conv1d_rs_stats_rows(int32_t dtype, const EmsaConvGeom* g) {
  if (dtype != EMSA_DT_BF16 && dtype != EMSA_DT_F16) return EMSA_E_ARG;
  RSPlan pl;
  if (!rs_plan(g, pl)) return EMSA_E_SHAPE;
  return 8 * pl.gx;
}

extern "C" int emsa_conv1d_rs_t(int32_t dtype, const EmsaConvGeom* g, const void* in,
                                const void* wfrag, void* out, const float* bias, float* stats,
                                const float* scale, const float* shift, const void* residual,
                                int32_t ld_res, const void* mask_src, int32_t ld_mask, int32_t act,
                                void* stream) {
  return conv_rs_impl(dtype, g, in, wfrag, out, bias, stats, scale, shift, residual, ld_res,
                      mask_src, ld_mask, act, nullptr, nullptr, nullptr, 0, stream);
}

// Forward conv on relu(in * in_scale[c] + in_shift[c]): the BatchNorm + ReLU in front of a 3x1 conv
// (the NBt1D block's bn1 in front of conv3x1_2, ref emsanet/model.py:47-58) folded into the loader --
// `in` is the BatchNorm's INPUT, the normalised tensor is never written.  16-bit twin of
// emsa_conv1d_wino_inbn; bf16, taps along H, epilogue: bias, statistics rows, ReLU.  EMSA_E_SHAPE
// where this form does not exist (the caller runs the normalise pass + emsa_conv1d_rs_t).
extern "C" int emsa_conv1d_rs_inbn_t(int32_t dtype, const EmsaConvGeom* g, const void* in,
                                     const void* wfrag, void* out, const float* bias, float* stats,
                                     const float* in_scale, const float* in_shift, int32_t act,
                                     void* stream) {
  if (!in_scale || !in_shift) return EMSA_E_ARG;
  return conv_rs_impl(dtype, g, in, wfrag, out, bias, stats, nullptr, nullptr, nullptr, 0, nullptr, 0,
                      act, nullptr, nullptr, nullptr, 0, stream, nullptr, in_scale, in_shift);
}

// Twin launch: the same conv geometry on two independent sets of tensors (the rgb | depth encoder
// blocks and the semantic | instance decoder blocks of the model have identical shapes) in ONE
// launch -- grid.y = 2, the second half of the workgroups reads the "1" operands.  Forward epilogue
// only (bias, folded BatchNorm scale / shift, residual, ReLU); each half's result is bit-identical
// to its own emsa_conv1d_rs_t launch.  bias / scale+shift / residual: given for both halves or neither.
extern "C" int emsa_conv1d_rs_pair_t(int32_t dtype, const EmsaConvGeom* g, const void* in0,
                                     const void* in1, const void* wfrag0, const void* wfrag1,
                                     void* out0, void* out1, const float* bias0, const float* bias1,
                                     const float* scale0, const float* scale1, const float* shift0,
                                     const float* shift1, const void* residual0,
                                     const void* residual1, int32_t ld_res, int32_t act,
                                     void* stream) {
  if (!in1 || !wfrag1 || !out1) return EMSA_E_ARG;
  if ((bias0 == nullptr) != (bias1 == nullptr) || (scale0 == nullptr) != (scale1 == nullptr) ||
      (shift0 == nullptr) != (shift1 == nullptr) || (residual0 == nullptr) != (residual1 == nullptr))
    return EMSA_E_ARG;
  if ((scale1 == nullptr) != (shift1 == nullptr)) return EMSA_E_ARG;
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  if (!al16(in1) || !al16(wfrag1) || !al16(out1) || !al16(bias1) || !al16(scale1) || !al16(shift1) ||
      !al16(residual1))
    return EMSA_E_SHAPE;
  if (out0 == out1) return EMSA_E_ARG;
  ConvRSArgs tw;
  tw.in2 = in1; tw.wf2 = wfrag1; tw.out2 = out1; tw.bias2 = bias1; tw.scale2 = scale1;
  tw.shift2 = shift1; tw.residual2 = residual1;
  return conv_rs_impl(dtype, g, in0, wfrag0, out0, bias0, nullptr, scale0, shift0, residual0, ld_res,
                      nullptr, 0, act, nullptr, nullptr, nullptr, 0, stream, &tw);
}

extern "C" int emsa_conv1d_rs_bnb_t(int32_t dtype, const EmsaConvGeom* g, const void* dy,
                                    const void* wfrag, void* out, const void* residual,
                                    int32_t ld_res, const void* t, int32_t ld_t,
                                    const float* bn_scale, const float* bn_shift,
                                    const float* bn_mean, const float* bn_invstd, float* partial,
                                    int32_t rows_alloc, void* stream) {
  if (!partial || !t) return EMSA_E_ARG;
  return conv_rs_impl(dtype, g, dy, wfrag, out, nullptr, nullptr, bn_scale, bn_shift, residual,
                      ld_res, t, ld_t, EMSA_ACT_NONE, bn_mean, bn_invstd, partial, rows_alloc, stream);
}

extern "C" int emsa_pack_weight_frag_t(int32_t dtype, const float* w_oihw, void* wf_fwd,
                                       void* wf_dgrad, int32_t cout, int32_t cin, void* stream) {
  if (!w_oihw || (!wf_fwd && !wf_dgrad)) return EMSA_E_ARG;
  if ((cout & 31) || (cin & 31)) return EMSA_E_SHAPE;
  const long total = 3L * cout * cin;
  int grid = (int)((total + 255) / 256);
  if (grid > 2048) grid = 2048;
  if (dtype == EMSA_DT_BF16)
    hipLaunchKernelGGL(pack_frag_kernel<emsa_bf16>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, (emsa_bf16*)wf_fwd, (emsa_bf16*)wf_dgrad, cout, cin);
  else if (dtype == EMSA_DT_F16)
    hipLaunchKernelGGL(pack_frag_kernel<emsa_f16>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, (emsa_f16*)wf_fwd, (emsa_f16*)wf_dgrad, cout, cin);
  else
    return EMSA_E_ARG;
  return emsa_launch_status();
}
