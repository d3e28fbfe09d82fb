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

// counter-based hash shared with oracle/emsanet_oracle.py (_lowbias32)
__host__ __device__ __forceinline__ uint32_t emsa_lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
