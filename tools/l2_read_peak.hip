// What does the vector memory path of a CU deliver into registers / LDS when the data sits in L2?
// (The 16-bit conv and weight-gradient kernels stage 350-700 MB per launch through it and never
// exceed ~11 TB/s chip-wide = ~45 GB/s per CU)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_peak tools/l2_read_peak.hip && /tmp/l2_peak
// Every workgroup re-reads its own `region` bytes (16 B per lane, rows of `row_bytes`) `iters`
// times: region * workgroups per XCD < 4 MB keeps it in the XCD's L2 after the first pass.
//   mode 0: global_load_dwordx4 into registers, 8 loads in flight per lane
//   mode 1: buffer_load_dwordx4 ... lds (LDS-DMA), 8 in flight, vmcnt(0) + barrier per batch
//   mode 2: like 1, then the tile is read back from LDS (ds_read_b128) as an MFMA loop would
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void reader(const unsigned char* __restrict__ src, size_t region,
                                              int iters, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const unsigned char* base = src + (size_t)blockIdx.x * region;
  const __amdgpu_buffer_rsrc_t rs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, (int)region, 0x00020000);
  const int batches = (int)(region / (256 * 16 * 8));      // 32 KB per batch of 8 loads per lane
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it)
    for (int b = 0; b < batches; ++b) {
      const uint32_t off0 = (uint32_t)b * (256 * 16 * 8) + (uint32_t)tid * 16;
      if (MODE == 0) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off0 + k * 4096), 0, 0);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k];
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              rs, (__attribute__((address_space(3))) void*)(lds + k * 4096 + wave * 1024), 16,
              (int)(off0 + k * 4096), 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (MODE == 2) {
#pragma unroll
          for (int k = 0; k < 8; ++k) acc ^= *reinterpret_cast<const u32x4*>(lds + k * 4096 + tid * 16);
          __syncthreads();
        }
      }
    }
  if (MODE == 1) acc ^= *reinterpret_cast<const u32x4*>(lds + tid * 16);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[tid] = acc[0];
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

int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  unsigned char* src;
  unsigned* out;
  const size_t total = (size_t)1 << 30;
  hipMalloc(&src, total);
  hipMemset(src, 1, total);
  hipMalloc(&out, 4096);
  printf("%s, %d CUs.  TB/s delivered chip-wide (GB/s per CU)\n", pr.gcnArchName, cus);
  for (size_t region : {(size_t)32 << 10, (size_t)256 << 10}) {
    for (int wpc : {1, 2, 4, 5}) {
      const int blocks = cus * wpc;
      if ((size_t)blocks * region > total) continue;
      const int iters = (int)(((size_t)48 << 20) / region);      // 48 MB per workgroup
      const double bytes = (double)blocks * region * iters;
      double t[3];
      t[0] = time_ms([&] { hipLaunchKernelGGL(reader<0>, dim3(blocks), dim3(256), 0, 0, src, region, iters, out); });
      t[1] = time_ms([&] { hipLaunchKernelGGL(reader<1>, dim3(blocks), dim3(256), 32768, 0, src, region, iters, out); });
      t[2] = time_ms([&] { hipLaunchKernelGGL(reader<2>, dim3(blocks), dim3(256), 32768, 0, src, region, iters, out); });
      printf("region %4zu KB/workgroup (%6.1f MB total), %d workgroups/CU:  registers %5.1f (%4.0f)   LDS-DMA %5.1f (%4.0f)   "
             "LDS-DMA + ds_read %5.1f (%4.0f)\n",
             region >> 10, blocks * region / 1048576.0, wpc, bytes / t[0] / 1e9, bytes / t[0] / 1e6 / cus,
             bytes / t[1] / 1e9, bytes / t[1] / 1e6 / cus, bytes / t[2] / 1e9, bytes / t[2] / 1e6 / cus);
    }
  }
  return 0;
}
