// Stand-alone probe (ROCm 7.2, gfx950): is a hipMemsetAsync captured as a graph MEMSET NODE still
// correct at replay when eager hipMemsetAsync calls run between instantiation / replays?
//
//   hipcc --offload-arch=gfx950 -O3 tools/graph_memset_repro.hip -o repro_memset && ./repro_memset
//
// Why it matters here: torch's multi-block reductions (`t.mean()`, `torch.equal`) zero their
// semaphore / accumulator scratch with cudaMemsetAsync.  Inside a captured training step (a user
// loss written with torch ops) those become memset nodes; `GraphedTrainStep` produced a wrong LOSS
// (66.76 instead of 3.09, all model outputs bit-identical) exactly when eager torch reductions ran
// between capture and replay (tools/graph_step_debug.py, PRE_SD=1 POST_SD=1).  Round 2 had met the
// same number and mis-attributed it to a `blockDim.x` read.
//
// Each captured (memset 0, add-one kernel) pair must leave its buffer at exactly 1 after EVERY
// replay; a memset node that does not clear (or clears the wrong bytes) leaves 2, 3, ... or garbage.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                 \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                \
    }                                                                         \
  } while (0)

__global__ void add_one(unsigned* p, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    p[i] += 1u;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  // sizes in 4-byte words: torch's semaphores are a few bytes, accumulators KBs
  const long sizes[] = {1, 2, 4, 16, 64, 1024, 4096, 1 << 16, 1 << 20};
  const int ns = sizeof(sizes) / sizeof(sizes[0]);
  const int reps_per_size = 8;                 // several memset nodes of each size in one graph
  std::vector<unsigned*> bufs;
  std::vector<long> lens;
  for (int r = 0; r < reps_per_size; ++r)
    for (int i = 0; i < ns; ++i) {
      unsigned* p;
      CK(hipMalloc(&p, sizes[i] * 4));
      CK(hipMemset(p, 0x55, sizes[i] * 4));
      bufs.push_back(p);
      lens.push_back(sizes[i]);
    }
  unsigned* scratch;
  CK(hipMalloc(&scratch, 8u << 20));

  for (int variant = 0; variant < 3; ++variant) {
    // variant 0: nothing eager between instantiate and replay; 1: eager memsets (other buffers)
    // between instantiate and the first replay and between replays; 2: additionally eager memsets
    // BEFORE the capture (allocator / command-buffer state)
    if (variant == 2)
      for (int k = 0; k < 64; ++k) CK(hipMemsetAsync(scratch + 64 * k, k, 4 + 4 * (k % 5), st));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (size_t b = 0; b < bufs.size(); ++b) {
      CK(hipMemsetAsync(bufs[b], 0, lens[b] * 4, st));
      const int grid = (int)((lens[b] + 255) / 256 > 512 ? 512 : (lens[b] + 255) / 256);
      hipLaunchKernelGGL(add_one, dim3(grid), dim3(256), 0, st, bufs[b], lens[b]);
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    long bad_total = 0;
    for (int rep = 0; rep < 4; ++rep) {
      if (variant >= 1)
        for (int k = 0; k < 200; ++k) {
          CK(hipMemsetAsync(scratch + 1024 * (k % 100), 0xAB + k, 4 + 4 * (k % 7), st));
          hipLaunchKernelGGL(add_one, dim3(1), dim3(64), 0, st, scratch + 1024 * (k % 100), 8L);
        }
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      long bad = 0;
      std::vector<unsigned> host;
      for (size_t b = 0; b < bufs.size(); ++b) {
        host.resize(lens[b]);
        CK(hipMemcpy(host.data(), bufs[b], lens[b] * 4, hipMemcpyDeviceToHost));
        long wrong = 0;
        for (long i = 0; i < lens[b]; ++i) wrong += host[i] != 1u;
        if (wrong) {
          if (bad < 6)
            printf("  variant %d replay %d: buffer %zu (%ld words): %ld words != 1 (first word %u)\n",
                   variant, rep, b, lens[b], wrong, host[0]);
          ++bad;
        }
      }
      printf("variant %d replay %d: %ld of %zu memset-node buffers wrong\n", variant, rep, bad,
             bufs.size());
      bad_total += bad;
    }
    printf("VARIANT %d %s\n", variant, bad_total ? "MISMATCH" : "consistent");
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  }
  return 0;
}
