// Attainable fp32 MFMA rate on this GPU: registers only, no memory.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak
// variants: waves per SIMD (1..8), independent accumulators per wave (1, 2, 4), and a loop that
// mimics the weight-gradient kernel's inner loop (LDS reads + transform VALU between MFMA groups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_only(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// inner loop of conv_wgrad1d_kernel<..., WINO>: 6 LDS reads + transforms + 4 MFMAs per pixel pair
__global__ __launch_bounds__(256) void mfma_lds(float* out, int iters) {
  __shared__ float lds[66 * 64 + 64];
  for (int i = threadIdx.x; i < 66 * 64 + 64; i += 256) lds[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float* d = lds + l31;
  const float* x = lds + 32 * 64 + l31;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int row = 2 * (2 * kk + lh);
      const float e0 = d[row * 64], e1 = d[(row + 1) * 64];
      const float d0 = x[row * 64], d1 = x[(row + 1) * 64], d2 = x[(row + 2) * 64],
                  d3 = x[(row + 3) * 64];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0, d0 - d2, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0 + e1, d1 + d2, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0 - e1, d2 - d1, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(-e1, d1 - d3, acc[3], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[threadIdx.x] = s;
}


// does VALU work overlap the matrix pipe?  NV fp32 FMAs (independent of the MFMAs) per MFMA
template <int NV>
__global__ __launch_bounds__(256) void mfma_valu(float* out, int iters, float a, float b) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = a + i + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], b, a);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// K loop of conv1d_wino_kernel<64>: per step 16 MFMAs per wave fed by 8 ds_read_b128 (2 KB each) and
// 4 packed FMAs, then barrier, 6 ds_write_b128 per thread (the 24 KB tile), barrier.  MODE bit0:
// with the LDS writes + barriers, bit1: with the LDS reads (else operands stay in registers)
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256, 5) void wino_like(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[6144];       // 24 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 6144; i += 256) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const float* a0 = lds + (0 * 32 + l31) * 16;
  const float* a1 = lds + (2 * 32 + l31) * 16;
  const float* b = lds + 2048 + (wave * 64 + l31) * 16;
  float4 st = make_float4(1.f, 2.f, 3.f, 4.f);
  float4 x0 = make_float4(1.f, 2.f, 3.f, 4.f), x1 = x0, f0 = x0, f1 = x0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int co = ((lh + 2 * t) << 2) ^ (((l31 >> 2) & 3) << 2);
      if (MODE & 2) {
        x0 = *reinterpret_cast<const float4*>(a0 + co);
        x1 = *reinterpret_cast<const float4*>(a1 + co);
        f0 = *reinterpret_cast<const float4*>(b + co);
        f1 = *reinterpret_cast<const float4*>(b + 32 * 16 + co);
      }
      const f32x2 sg = {-1.f, -1.f};
      const f32x2 vlo = __builtin_elementwise_fma(f32x2{x1.x, x1.y}, sg, f32x2{x0.x, x0.y});
      const f32x2 vhi = __builtin_elementwise_fma(f32x2{x1.z, x1.w}, sg, f32x2{x0.z, x0.w});
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vlo.x, f0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vlo.x, f1.x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vlo.y, f0.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vlo.y, f1.y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vhi.x, f0.z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vhi.x, f1.z, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vhi.y, f0.w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vhi.y, f1.w, acc[1], 0, 0, 0);
    }
    if (MODE & 1) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 6; ++j)
        *reinterpret_cast<float4*>(lds + ((tid >> 2) + 64 * j) * 16 + (tid & 3) * 4) = st;
      __syncthreads();
    }
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[threadIdx.x] = s;
}


// K loop of a Winograd F(4,3) kernel (6 transformed components = 6 waves, 32 quads x 64 channels per
// workgroup, 16 channels per K step): per step and wave 16 MFMAs fed by 12 ds_read_b128 (each V_j is
// a 4-term combination of the raw rows d0..d5) and 16 packed FMAs; 36 KB of LDS per workgroup
// (6 ds_write_b128 per thread and step).  F(4,3) executes 6 products per 4 outputs x 3 taps = 1/2 of
// the direct-convolution multiplies (F(2,3): 2/3).  MODE as in wino_like.
template <int MODE, int WPE>
__global__ __launch_bounds__(384, WPE) void f43_like(float* out, int iters, float c0, float c1,
                                                      float c2, float c3) {
  __shared__ __attribute__((aligned(16))) float lds[9216];       // 36 KB: A [6][32][16], B [6][64][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 9216; i += 384) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // rows of this wave's component (wave-uniform): comps 0 / 5 use rows {0,2,4} / {1,3,5} (+ one
  // dummy), the others rows 1..4
  const int r0 = wave == 0 ? 0 : 1, r1 = (wave == 0 || wave == 5) ? r0 + 2 : 2,
            r2 = (wave == 0 || wave == 5) ? r0 + 4 : 3, r3 = 4;
  const float* a0 = lds + (r0 * 32 + l31) * 16;
  const float* a1 = lds + (r1 * 32 + l31) * 16;
  const float* a2 = lds + (r2 * 32 + l31) * 16;
  const float* a3 = lds + (r3 * 32 + l31) * 16;
  const float* b = lds + 3072 + (wave * 64 + l31) * 16;
  const f32x2 k0 = {c0, c0}, k1 = {c1, c1}, k2 = {c2, c2}, k3 = {c3, c3};
  float4 st = make_float4(1.f, 2.f, 3.f, 4.f);
  float4 x0 = st, x1 = st, x2 = st, x3 = st, f0 = st, f1 = st;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int co = ((lh + 2 * t) << 2) ^ (((l31 >> 2) & 3) << 2);
      if (MODE & 2) {
        x0 = *reinterpret_cast<const float4*>(a0 + co);
        x1 = *reinterpret_cast<const float4*>(a1 + co);
        x2 = *reinterpret_cast<const float4*>(a2 + co);
        x3 = *reinterpret_cast<const float4*>(a3 + co);
        f0 = *reinterpret_cast<const float4*>(b + co);
        f1 = *reinterpret_cast<const float4*>(b + 32 * 16 + co);
      }
      f32x2 vlo = f32x2{x3.x, x3.y} * k3, vhi = f32x2{x3.z, x3.w} * k3;
      vlo = __builtin_elementwise_fma(f32x2{x2.x, x2.y}, k2, vlo);
      vhi = __builtin_elementwise_fma(f32x2{x2.z, x2.w}, k2, vhi);
      vlo = __builtin_elementwise_fma(f32x2{x1.x, x1.y}, k1, vlo);
      vhi = __builtin_elementwise_fma(f32x2{x1.z, x1.w}, k1, vhi);
      vlo = __builtin_elementwise_fma(f32x2{x0.x, x0.y}, k0, vlo);
      vhi = __builtin_elementwise_fma(f32x2{x0.z, x0.w}, k0, vhi);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vlo.x, f0.x, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vlo.x, f1.x, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vlo.y, f0.y, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vlo.y, f1.y, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vhi.x, f0.z, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vhi.x, f1.z, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vhi.y, f0.w, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vhi.y, f1.w, acc[1], 0, 0, 0);
    }
    if (MODE & 1) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 6; ++j)
        *reinterpret_cast<float4*>(lds + ((tid >> 2) + 96 * j) * 16 + (tid & 3) * 4) = st;
      __syncthreads();
    }
    asm volatile("" ::: "memory");
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <class F>
double time_ms(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

int main() {
  float* out; hipMalloc(&out, 4096);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  printf("%s: %d CUs, clock %d MHz\n", pr.gcnArchName, cus, pr.clockRate / 1000);
  const int iters = 20000;            // x16 MFMAs per wave: ~20 ms per launch at full rate
  for (int wps : {1, 2, 4, 6}) {
    const int blocks = cus * wps;     // 256 threads = 4 waves = one wave per SIMD per block
    const double fl = (double)blocks * 4 * iters * 16 * 32 * 32 * 2 * 2;
    double m1 = time_ms([&] { hipLaunchKernelGGL(mfma_only<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    double m2 = time_ms([&] { hipLaunchKernelGGL(mfma_only<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    double m4 = time_ms([&] { hipLaunchKernelGGL(mfma_only<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    printf("waves/SIMD %d  mfma only: 1 acc %6.1f  2 acc %6.1f  4 acc %6.1f TFLOP/s\n", wps,
           fl / m1 / 1e9, fl / m2 / 1e9, fl / m4 / 1e9);
  }
  for (int wps : {1, 2, 3, 4, 6}) {
    const int blocks = cus * wps, it2 = 10000;
    const double fl = (double)blocks * 4 * it2 * 32 * 32 * 32 * 2 * 2;
    double m = time_ms([&] { hipLaunchKernelGGL(mfma_lds, dim3(blocks), dim3(256), 0, 0, out, it2); });
    printf("waves/SIMD %d  wgrad-like loop (LDS + transform + 4 MFMA): %6.1f TFLOP/s\n", wps,
           fl / m / 1e9);
  }
  for (int wps : {1, 3, 5}) {
    const int blocks = cus * wps;
    const double fl = (double)blocks * 4 * iters * 16 * 32 * 32 * 2 * 2;
    double m0 = time_ms([&] { hipLaunchKernelGGL(mfma_valu<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    double m2 = time_ms([&] { hipLaunchKernelGGL(mfma_valu<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    double m4 = time_ms([&] { hipLaunchKernelGGL(mfma_valu<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    double m8 = time_ms([&] { hipLaunchKernelGGL(mfma_valu<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    double m12 = time_ms([&] { hipLaunchKernelGGL(mfma_valu<12>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); });
    printf("waves/SIMD %d  FMAs per MFMA 0/2/4/8/12: %6.1f %6.1f %6.1f %6.1f %6.1f TFLOP/s\n", wps,
           fl / m0 / 1e9, fl / m2 / 1e9, fl / m4 / 1e9, fl / m8 / 1e9, fl / m12 / 1e9);
  }
  for (int wps : {1, 3, 5}) {
    const int blocks = cus * wps, it2 = 20000;
    const double fl = (double)blocks * 4 * it2 * 16 * 32 * 32 * 2 * 2;
    double m0 = time_ms([&] { hipLaunchKernelGGL(wino_like<0>, dim3(blocks), dim3(256), 0, 0, out, it2); });
    double m2 = time_ms([&] { hipLaunchKernelGGL(wino_like<2>, dim3(blocks), dim3(256), 0, 0, out, it2); });
    double m3 = time_ms([&] { hipLaunchKernelGGL(wino_like<3>, dim3(blocks), dim3(256), 0, 0, out, it2); });
    printf("workgroups/CU %d  Winograd K loop: MFMA+transform %6.1f  +LDS reads %6.1f  +LDS writes, 2 barriers %6.1f TFLOP/s\n",
           wps, fl / m0 / 1e9, fl / m2 / 1e9, fl / m3 / 1e9);
  }
  // F(4,3): executed MFMA rate of its K loop; x2 = direct-convolution equivalent (F(2,3): x1.5)
  for (int wps : {1, 2, 3, 4}) {
    const int blocks = cus * wps, it2 = 20000;
    const double fl = (double)blocks * 6 * it2 * 16 * 32 * 32 * 2 * 2;
    double m0 = time_ms([&] { hipLaunchKernelGGL((f43_like<0, 4>), dim3(blocks), dim3(384), 0, 0, out, it2, 4.f, -5.f, 1.f, 0.5f); });
    double m2 = time_ms([&] { hipLaunchKernelGGL((f43_like<2, 4>), dim3(blocks), dim3(384), 0, 0, out, it2, 4.f, -5.f, 1.f, 0.5f); });
    double m3 = time_ms([&] { hipLaunchKernelGGL((f43_like<3, 4>), dim3(blocks), dim3(384), 0, 0, out, it2, 4.f, -5.f, 1.f, 0.5f); });
    printf("workgroups/CU %d  F(4,3) K loop (6 waves): MFMA+transform %6.1f  +LDS reads %6.1f  +LDS writes, 2 barriers %6.1f TFLOP/s executed (x2 = direct-conv equivalent)\n",
           wps, fl / m0 / 1e9, fl / m2 / 1e9, fl / m3 / 1e9);
  }
  return 0;
}
