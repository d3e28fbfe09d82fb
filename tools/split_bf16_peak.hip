// Feasibility probe for fp32 convolutions on the bf16 matrix pipe ("what comes
// next"): an fp32 operand x is split into three bf16 terms x = h + m + l (8 + 8 + 8 mantissa bits);
// the six products hh, hm, mh, mm, hl, lh accumulated in fp32 reproduce an fp32 GEMM to ~1e-7
// (numpy check in DESIGN.md).  Six v_mfma_f32_32x32x16_bf16 cost 12 cycles per unit of K against 32
// for v_mfma_f32_32x32x2_f32 -- if the split (vector ALU) and the 3x LDS operand traffic keep up.
//
// This is the K loop of such a kernel without global memory: a 128 x 64 output tile per workgroup,
// 4 waves of 64 x 32, K step of HK channels; per step every thread splits 128*HK/256 fp32 values
// (held in registers, as if just loaded) into three bf16 LDS tiles, the weight tiles are pre-split
// (written as-is), then per 16 channels 9 ds_read_b128 feed 12 MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/split_peak tools/split_bf16_peak.hip && /tmp/split_peak
// prints fp32-EQUIVALENT TFLOP/s (2*M*N*K per step, the six-fold work not counted).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk(float a, float b) {
  const f32x2 v = {a, b};
  const bf16x2 h = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(unsigned, h);
}
// x0, x1 -> packed (h, m, l) pairs
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = pk(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
  m = pk(r0, r1);
  l = pk(r0 - __uint_as_float(m << 16), r1 - __uint_as_float(m & 0xFFFF0000u));
}

// MODE bit0: split + LDS writes + barriers each step; bit1: LDS operand reads (else registers);
// NPROD: products per (a, b) pair (6 = fp32-equivalent, 3 = hh + hm + mh)
template <int HK, int MODE, int NPROD, int WPC>
__global__ __launch_bounds__(256, WPC) void x6_like(float* out, int iters, float seed) {
  constexpr int BM = 128, BN = 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __bf16* const As = reinterpret_cast<__bf16*>(smem);          // [3][BM][HK]
  __bf16* const Bs = As + 3 * BM * HK;                          // [3][BN][HK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  constexpr int EPT = BM * HK / 256;                            // fp32 values per thread per step
  float x[EPT];
  for (int i = 0; i < EPT; ++i) x[i] = seed * (float)(tid * 7 + i) + 0.37f;
  // rows of 8 channels (16 B) per lane, swizzled by row like conv_h's tiles
  constexpr int CPR = HK / 8;                                   // 16-byte chunks per row
  for (int it = 0; it < iters; ++it) {
    if (MODE & 1) {
      __syncthreads();
#pragma unroll
      for (int e = 0; e < EPT; e += 8) {
        const int chunk = tid + (e / 8) * 256;                  // chunk index in the [BM][CPR] tile
        const int row = chunk / CPR, cc = (chunk % CPR) ^ (row & (CPR - 1));
        u32x4 h, m, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          unsigned a, b, c;
          split2(x[e + 2 * k], x[e + 2 * k + 1], a, b, c);
          h[k] = a; m[k] = b; l[k] = c;
        }
        *reinterpret_cast<u32x4*>(As + (0 * BM + row) * HK + cc * 8) = h;
        *reinterpret_cast<u32x4*>(As + (1 * BM + row) * HK + cc * 8) = m;
        *reinterpret_cast<u32x4*>(As + (2 * BM + row) * HK + cc * 8) = l;
      }
      // pre-split weights: 3 * BN * HK * 2 bytes per step, 16 bytes per lane
#pragma unroll
      for (int c = tid; c < 3 * BN * CPR; c += 256) {
        const u32x4 w = {(unsigned)c, (unsigned)it, 0x3f803f80u, 0x3f803f80u};
        *reinterpret_cast<u32x4*>(Bs + c * 8) = w;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < EPT; ++i) x[i] += 1.0f;               // "fresh data" for the next step
    }
#pragma unroll
    for (int ks = 0; ks < HK / 16; ++ks) {
      bf16x8 a[3][2], b[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (MODE & 2) {
            const int row = (wm * 2 + i) * 32 + l31, cc = (ks * 2 + lh) ^ (row & (CPR - 1));
            a[p][i] = *reinterpret_cast<const bf16x8*>(As + (p * BM + row) * HK + cc * 8);
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) a[p][i][k] = (__bf16)(seed + p + i);
          }
        }
        if (MODE & 2) {
          const int row = wn * 32 + l31, cc = (ks * 2 + lh) ^ (row & (CPR - 1));
          b[p] = *reinterpret_cast<const bf16x8*>(Bs + (p * BN + row) * HK + cc * 8);
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) b[p][k] = (__bf16)(seed - p);
        }
      }
      // small terms first: lh, hl, mm, mh, hm, hh
      constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
      for (int q = NPROD - 1; q >= 0; --q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][i], b[PB[q]], acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <typename F>
double time_ms(F f) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

template <int HK, int NPROD, int WPC>
void run(float* out, int cus) {
  const int blocks = cus * WPC, iters = 4000;
  const size_t lds = (size_t)3 * (128 + 64) * HK * 2;
  const double fl = (double)blocks * iters * 2.0 * 128 * 64 * HK;
  double t[3];
  int k = 0;
  auto go = [&](auto kern) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    t[k++] = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, iters, 0.001f); });
  };
  go(x6_like<HK, 0, NPROD, WPC>);
  go(x6_like<HK, 2, NPROD, WPC>);
  go(x6_like<HK, 3, NPROD, WPC>);
  printf("K step %2d  %d products  workgroups/CU %d (LDS %3zu KB):  MFMA only %6.1f   + LDS reads %6.1f   "
         "+ split, LDS writes, 2 barriers %6.1f   fp32-equivalent TFLOP/s\n",
         HK, NPROD, WPC, lds >> 10, fl / t[0] / 1e9, fl / t[1] / 1e9, fl / t[2] / 1e9);
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  printf("%s: %d CUs, clock %d MHz; fp32 MFMA peak of this part 157.3 TFLOP/s\n", pr.gcnArchName, cus,
         pr.clockRate / 1000);
  run<64, 6, 1>(out, cus);
  run<64, 6, 2>(out, cus);
  run<32, 6, 2>(out, cus);
  run<32, 6, 3>(out, cus);
  run<32, 6, 4>(out, cus);
  run<64, 3, 2>(out, cus);
  run<32, 3, 4>(out, cus);
  return 0;
}
