// Stand-alone probe for the round-2 finding: does a kernel see the same
// blockDim.x / gridDim.x under hipGraph replay as under an eager launch?
//
//   hipcc --offload-arch=gfx950 -O3 [-mcode-object-version=4] tools/graph_blockdim_repro.hip -o repro
//   ./repro [nodes]
//
// With code object v5 (hipcc's default) `blockDim` / `gridDim` are NOT read from the AQL dispatch
// packet: the compiler loads them from HIDDEN kernel arguments (hidden_group_size_x,
// hidden_block_count_x, hidden_remainder_x) that the HIP runtime writes behind the explicit
// arguments when it builds the kernarg block -- for a graph kernel node that block is built at
// capture / instantiate time and reused at every replay.  With -mcode-object-version=4 the same
// source reads workgroup_size_x / grid_size_x from the dispatch packet itself (s_load from the
// dispatch pointer), which cannot disagree with the launch.
//
// The probe mimics the shape of the failing case: ONE templated kernel with dynamic LDS, a
// __syncthreads, and a trailing loop stepping by blockDim.x, captured MANY times per graph with
// different grids / LDS sizes / block sizes, interleaved with other kernels, replayed while eager
// launches of the same kernels run between replays.  Every launch records what it saw.
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

struct Seen {
  unsigned bdx, gdx, lds_words, sum_lo;
};

// rec[slot]: what block 0 saw; `work`: c floats reduced through LDS like channel_dot_kernel does
template <int V>
__global__ void probe_kernel(const float* __restrict__ a, float* __restrict__ ws, Seen* rec, int slot,
                             int c, int lanes) {
  extern __shared__ __attribute__((aligned(16))) float cred[];
  const int cv = threadIdx.x % (c / V), rl = threadIdx.x / (c / V);
  if (rl < lanes) {
#pragma unroll
    for (int k = 0; k < V; ++k) cred[rl * c + cv * V + k] = a[(blockIdx.x * lanes + rl) * c + cv * V + k];
  }
  __syncthreads();
  unsigned n_written = 0;
  for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {     // <- the line of the round-2 bisection
    float t = 0.f;
    for (int k = 0; k < lanes; ++k) t += cred[k * c + ch];
    ws[(long)blockIdx.x * c + ch] = t;
    ++n_written;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    rec[slot].bdx = blockDim.x;
    rec[slot].gdx = gridDim.x;
    rec[slot].lds_words = __builtin_amdgcn_groupstaticsize();
    rec[slot].sum_lo = n_written;
  }
}

__global__ void filler_kernel(float* p, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    p[i] = p[i] * 0.5f + 1.f;
}

struct Launch {
  int block, grid, c, lanes, v;
};

static void issue(const Launch& l, const float* a, float* ws, Seen* rec, int slot, hipStream_t st) {
  const size_t lds = (size_t)l.lanes * l.c * sizeof(float);
  if (l.v == 4)
    hipLaunchKernelGGL((probe_kernel<4>), dim3(l.grid), dim3(l.block), lds, st, a, ws, rec, slot, l.c,
                       l.lanes);
  else
    hipLaunchKernelGGL((probe_kernel<8>), dim3(l.grid), dim3(l.block), lds, st, a, ws, rec, slot, l.c,
                       l.lanes);
}

int main(int argc, char** argv) {
  const int nodes = argc > 1 ? atoi(argv[1]) : 4000;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *a, *ws, *fill;
  Seen* rec;
  CK(hipMalloc(&a, 64u << 20));
  CK(hipMalloc(&ws, 64u << 20));
  CK(hipMalloc(&fill, 16u << 20));
  CK(hipMalloc(&rec, sizeof(Seen) * nodes * 2));
  CK(hipMemset(a, 0, 64u << 20));
  CK(hipMemset(rec, 0xFF, sizeof(Seen) * nodes * 2));
  std::vector<Launch> plan(nodes);
  const int blocks[4] = {256, 256, 128, 64}, cs[5] = {64, 128, 256, 512, 1024};
  for (int i = 0; i < nodes; ++i) {
    Launch l;
    l.block = blocks[i % 4];
    l.c = cs[(i / 3) % 5];
    l.v = (i % 7 == 0) ? 8 : 4;
    if (l.c / l.v > l.block) l.block = 256;
    if (l.c / l.v > l.block) l.c = 256 * l.v;
    l.lanes = l.block / (l.c / l.v);
    l.grid = 1 + (i * 37) % 640;
    plan[i] = l;
  }
  // eager reference pass
  for (int i = 0; i < nodes; ++i) issue(plan[i], a, ws, rec, i, st);
  CK(hipStreamSynchronize(st));
  // the same launches captured into one graph (+ filler kernels with other shapes in between)
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < nodes; ++i) {
    issue(plan[i], a, ws, rec, nodes + i, st);
    if (i % 5 == 0)
      hipLaunchKernelGGL(filler_kernel, dim3(1 + i % 300), dim3(i % 2 ? 256 : 512), 0, st, fill,
                         (long)(1 << 20));
  }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  std::vector<Seen> host(nodes * 2);
  long bad_total = 0;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipMemsetAsync(rec + nodes, 0xFF, sizeof(Seen) * nodes, st));
    CK(hipGraphLaunch(ge, st));
    // eager work of the same kernels between replays (the failing test ran an eager twin)
    for (int i = 0; i < nodes; i += 11) issue(plan[(i * 13) % nodes], a, ws, rec, (i * 13) % nodes, st);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(host.data(), rec, sizeof(Seen) * nodes * 2, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int i = 0; i < nodes; ++i) {
      const Seen &e = host[i], &r = host[nodes + i];
      const unsigned want_w = (plan[i].c + plan[i].block - 1) / plan[i].block;   // thread 0's trips
      const bool eager_ok = e.bdx == (unsigned)plan[i].block && e.gdx == (unsigned)plan[i].grid;
      const bool replay_ok = r.bdx == (unsigned)plan[i].block && r.gdx == (unsigned)plan[i].grid &&
                             r.sum_lo == want_w;
      if (!eager_ok || !replay_ok) {
        if (bad < 10)
          printf("  replay %d node %d: launched block %d grid %d | eager saw %u/%u | replay saw %u/%u "
                 "(trips %u, want %u)\n",
                 rep, i, plan[i].block, plan[i].grid, e.bdx, e.gdx, r.bdx, r.gdx, r.sum_lo, want_w);
        ++bad;
      }
    }
    printf("replay %d: %ld of %d kernel nodes saw a blockDim.x / gridDim.x different from their launch\n",
           rep, bad, nodes);
    bad_total += bad;
  }
  printf("RESULT %s\n", bad_total ? "MISMATCH" : "consistent");
  return bad_total ? 1 : 0;
}
