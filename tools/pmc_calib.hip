// PMC calibration on known byte counts (VERDICT r3 item 5 ii): three kernels, each moving exactly
// 1 GiB with 16 bytes per lane -- pure read, pure write, copy -- four launches each, to be run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   and   rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
// (separate passes, MI355X_MICROARCH.md HBM section).  The buffers rotate over 4 GiB so that no launch
// finds its data in the 256 MiB Infinity Cache.  tools/pmc_traffic_json.py turns the three rows into
// the FETCH / WRITE correction factors it applies to the conv kernels of the same pass.
//   hipcc --offload-arch=gfx950 -O2 tools/pmc_calib.hip -o tools/bin/pmc_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void pmc_calib_read_1gib(const uint4* __restrict__ src, size_t n16, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = src[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) *sink = acc;       // (never true for the fill pattern: no stores)
}
__global__ void pmc_calib_write_1gib(uint4* __restrict__ dst, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = make_uint4((unsigned)i, 1u, 2u, 3u);
}
__global__ void pmc_calib_copy_1gib(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

int main() {
  const size_t GIB = (size_t)1 << 30, n16 = GIB / 16;
  uint4* buf[5];
  for (int i = 0; i < 5; ++i) {
    CK(hipMalloc(&buf[i], GIB));
    pmc_calib_write_1gib<<<4096, 256>>>(buf[i], n16);      // (also the first write launches)
  }
  unsigned* sink;
  CK(hipMalloc(&sink, 4));
  CK(hipDeviceSynchronize());
  for (int r = 0; r < 4; ++r) pmc_calib_read_1gib<<<4096, 256>>>(buf[r], n16, sink);
  CK(hipDeviceSynchronize());
  for (int r = 0; r < 4; ++r) pmc_calib_copy_1gib<<<4096, 256>>>(buf[r], buf[(r + 2) % 5], n16);
  CK(hipDeviceSynchronize());
  printf("pmc_calib: 5 write, 4 read, 4 copy launches of 1 GiB\n");
  return 0;
}
