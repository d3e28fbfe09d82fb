// Shared helpers for the gfx950 kernels of libemsanet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "emsanet_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define EMSA_WAVE 64
#define EMSA_NXCD 8

static inline int emsa_launch_status() {
  return hipGetLastError() == hipSuccess ? EMSA_OK : EMSA_E_LAUNCH;
}

// Bijective XCD-aware block remap (MI355X: block b is dispatched to XCD b % 8, each XCD has its
// own 4 MiB L2): give every XCD one contiguous chunk of the tile space so that neighbouring
// tiles (which share input rows / weight panels) hit the same L2.
__device__ __forceinline__ int emsa_xcd_remap(int bid, int nwg) {
  const int q = nwg / EMSA_NXCD, r = nwg % EMSA_NXCD;
  const int xcd = bid % EMSA_NXCD, idx = bid / EMSA_NXCD;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float4 emsa_ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}
__device__ __forceinline__ void emsa_st4(float* p, float4 v) {
  *reinterpret_cast<float4*>(p) = v;
}
__device__ __forceinline__ float4 emsa_zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// ---- 16-bit activation storage (BASELINE configs 3 / 5: bf16 training, fp16|bf16 inference) ------
// Kernels templated on the STORAGE type T of their activation tensors read / write through these
// overloads; all arithmetic stays fp32 in registers.  Four consecutive channels = one 8-byte access
// for the 16-bit types (16 bytes for float); conversions are single v_cvt_pk instructions on gfx950.
typedef __bf16 emsa_bf16;
typedef _Float16 emsa_f16;
typedef __bf16 emsa_bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 emsa_f16x4 __attribute__((ext_vector_type(4)));
typedef float emsa_f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 emsa_ld4(const emsa_bf16* p) {
  const emsa_f32x4v v = __builtin_convertvector(*reinterpret_cast<const emsa_bf16x4*>(p), emsa_f32x4v);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 emsa_ld4(const emsa_f16* p) {
  const emsa_f32x4v v = __builtin_convertvector(*reinterpret_cast<const emsa_f16x4*>(p), emsa_f32x4v);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void emsa_st4(emsa_bf16* p, float4 v) {
  const emsa_f32x4v f = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<emsa_bf16x4*>(p) = __builtin_convertvector(f, emsa_bf16x4);
}
__device__ __forceinline__ void emsa_st4(emsa_f16* p, float4 v) {
  const emsa_f32x4v f = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<emsa_f16x4*>(p) = __builtin_convertvector(f, emsa_f16x4);
}
// scalar element access
__device__ __forceinline__ float emsa_ld1(const float* p) { return *p; }
__device__ __forceinline__ float emsa_ld1(const emsa_bf16* p) { return (float)*p; }
__device__ __forceinline__ float emsa_ld1(const emsa_f16* p) { return (float)*p; }
__device__ __forceinline__ void emsa_st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void emsa_st1(emsa_bf16* p, float v) { *p = (emsa_bf16)v; }
__device__ __forceinline__ void emsa_st1(emsa_f16* p, float v) { *p = (emsa_f16)v; }

// run `body` (a generic lambda taking a type tag) with the storage type selected by `dtype`
#define EMSA_DISPATCH_DTYPE(dtype, T, ...)                              \
  switch (dtype) {                                                      \
    case EMSA_DT_F32: { using T = float; __VA_ARGS__; } break;          \
    case EMSA_DT_BF16: { using T = emsa_bf16; __VA_ARGS__; } break;     \
    case EMSA_DT_F16: { using T = emsa_f16; __VA_ARGS__; } break;       \
    default: return EMSA_E_ARG;                                         \
  }

// Zero-fill by a KERNEL.  hipMemsetAsync is not used anywhere in this library: captured into a
// hipGraph as a memset node it was found to be clobbered when other (eager) work that itself issues
// memsets runs between two replays of the graph (ROCm 7.2: the gradients behind the pyramid-pooling
// bilinear backward turned to garbage from the second replay on; kernels never showed this).
static __global__ void emsa_zero_kernel(uint32_t* __restrict__ p, long n_words) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_words;
       i += (long)gridDim.x * blockDim.x)
    p[i] = 0u;
}
static inline void emsa_zero_async(void* p, size_t bytes, hipStream_t st) {
  const long words = (long)(bytes / 4);      // (every buffer zeroed here is a multiple of 4 bytes)
  long blocks = (words + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(emsa_zero_kernel, dim3((int)blocks), dim3(256), 0, st, (uint32_t*)p, words);
}

// counter-based hash shared with oracle/emsanet_oracle.py (_lowbias32)
__host__ __device__ __forceinline__ uint32_t emsa_lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
