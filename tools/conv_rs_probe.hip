// Stand-alone probe of the register-stationary streaming conv (csrc/conv_rs.hip) against the
// generic 16-bit implicit GEMM (csrc/conv_h.hip) through the C-ABI of libemsanet_hip.so: results
// (outputs, BatchNorm statistics rows, fused residual / mask / BatchNorm-backward sums) and time per
// launch on the bs=32 640x480 layer shapes.  No torch: starts in a second on a fresh box.
//   hipcc --offload-arch=gfx950 -O2 -Iinclude tools/conv_rs_probe.hip -Lemsanet_amd/lib -lemsanet_hip \
//         -Wl,-rpath,'$ORIGIN/../../emsanet_amd/lib' -o tools/bin/conv_rs_probe
//   tools/bin/conv_rs_probe [batch] [check|time|all]
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "emsanet_hip.h"

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

typedef __bf16 bf16;

__global__ void fill_bf16(bf16* p, long n, uint32_t seed, float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    p[i] = (bf16)(((float)(x & 0xFFFF) / 32768.f - 1.f) * scale);
  }
}
__global__ void fill_f32(float* p, long n, uint32_t seed, float scale, float off) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u + seed;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    p[i] = ((float)(x & 0xFFFF) / 32768.f - 1.f) * scale + off;
  }
}
// max |a - b| and number of elements differing by more than tol (both bf16 tensors)
__global__ void cmp_bf16(const bf16* a, const bf16* b, long n, float tol, float* maxd, unsigned long long* bad) {
  float m = 0.f;
  unsigned long long c = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = fabsf((float)a[i] - (float)b[i]);
    const float ref = fabsf((float)b[i]);
    if (!(d <= tol * fmaxf(1.f, ref))) ++c;
    m = fmaxf(m, d == d ? d : 1e30f);
  }
  atomicMax((int*)maxd, __float_as_int(m));
  if (c) atomicAdd(bad, c);
}

struct Shape {
  const char* name;
  int c, h, w, kh, kw;
};

static EmsaConvGeom geom(int n, int h, int w, int c, int kh, int kw, bool dgrad) {
  EmsaConvGeom g;
  memset(&g, 0, sizeof(g));
  const int ph = kh / 2, pw = kw / 2;
  g.n_img = n; g.in_h = h; g.in_w = w; g.out_h = h; g.out_w = w;
  g.k_ch = c; g.n_ch = c; g.kh = kh; g.kw = kw;
  if (!dgrad) {
    g.mul_h = 1; g.off_h = -ph; g.step_h = 1; g.div_h = 1;
    g.mul_w = 1; g.off_w = -pw; g.step_w = 1; g.div_w = 1;
  } else {
    g.mul_h = 1; g.off_h = ph; g.step_h = -1; g.div_h = 1;
    g.mul_w = 1; g.off_w = pw; g.step_w = -1; g.div_w = 1;
  }
  g.in_img_stride = (int64_t)h * w * c; g.in_row_stride = (int64_t)w * c; g.in_px_stride = c;
  g.ld_out = c;
  return g;
}

// merge stats rows [3][rows][c] -> mean / var per channel (fp64 Chan)
static void merge_stats(const std::vector<float>& s, int rows, int c, std::vector<double>& mean,
                        std::vector<double>& var) {
  mean.assign(c, 0.0); var.assign(c, 0.0);
  for (int ch = 0; ch < c; ++ch) {
    double na = 0, ma = 0, qa = 0;
    for (int r = 0; r < rows; ++r) {
      const double nb = s[(2L * rows + r) * c + ch];
      if (nb <= 0) continue;
      const double mb = s[(0L * rows + r) * c + ch] / nb, qb = s[(1L * rows + r) * c + ch];
      const double nn = na + nb, d = mb - ma;
      qa += qb + d * d * na * nb / nn;
      ma += d * nb / nn;
      na = nn;
    }
    mean[ch] = ma; var[ch] = na > 0 ? qa / na : 0;
  }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 32;
  const char* what = argc > 2 ? argv[2] : "all";
  const bool do_check = strcmp(what, "time") != 0, do_time = strcmp(what, "check") != 0;
  const char* only = getenv("PROBE_SHAPE");
  printf("arch %s  batch %d\n", emsa_arch(), n);
  const Shape shapes[] = {
      {"1x3 c64 /4", 64, 120, 160, 1, 3},   {"3x1 c64 /4", 64, 120, 160, 3, 1},
      {"1x3 c128 /8", 128, 60, 80, 1, 3},   {"3x1 c128 /8", 128, 60, 80, 3, 1},
      {"1x3 c256 /16", 256, 30, 40, 1, 3},  {"3x1 c256 /16", 256, 30, 40, 3, 1},
      {"1x3 c512 /32", 512, 15, 20, 1, 3},  {"3x1 c512 /32", 512, 15, 20, 3, 1},
      // odd sizes (config 4: 960x736 -> /32 = 23 x 30; tails, partial tiles)
      {"1x3 c64 odd", 64, 23, 30, 1, 3},    {"3x1 c64 odd", 64, 23, 30, 3, 1},
      {"1x3 c512 odd", 512, 23, 30, 1, 3},  {"3x1 c512 odd", 512, 23, 30, 3, 1},
      {"3x1 c128 odd", 128, 46, 60, 3, 1},  {"1x3 c256 odd", 256, 46, 60, 1, 3},
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float* d_max;
  unsigned long long* d_bad;
  CK(hipMalloc(&d_max, 4));
  CK(hipMalloc(&d_bad, 8));
  int fails = 0;
  // phase table of a debug build (EMSA_RS_DBG=1), absent from the product library
  typedef int (*dbg_read_t)(long long*, int);
  dbg_read_t dbg_read = (dbg_read_t)dlsym(RTLD_DEFAULT, "emsa_conv1d_rs_dbg_read");
  if (do_time) printf("%-16s %-6s %9s %9s %8s %9s %9s\n", "shape", "kind", "old us", "new us", "x", "new TF/s", "new TB/s");
  for (const Shape& s : shapes) {
    if (only && !strstr(s.name, only)) continue;
    const bool odd = strstr(s.name, "odd") != nullptr;
    const int nb_img = odd ? (n > 5 ? 5 : n) : n;
    const long M = (long)nb_img * s.h * s.w, elems = M * s.c;
    // rotating buffer sets > Infinity Cache so that every timed launch reads from HBM
    const int sets = do_time && !odd ? (int)((900L << 20) / (elems * 4) + 2) : 1;
    std::vector<bf16*> X(sets), Y0(sets), Y1(sets);
    for (int i = 0; i < sets; ++i) {
      CK(hipMalloc(&X[i], elems * 2));
      CK(hipMalloc(&Y0[i], elems * 2));
      CK(hipMalloc(&Y1[i], elems * 2));
      fill_bf16<<<1024, 256, 0, st>>>(X[i], elems, 17 + i, 1.f);
    }
    bf16 *R, *Mk;
    CK(hipMalloc(&R, elems * 2));
    CK(hipMalloc(&Mk, elems * 2));
    fill_bf16<<<1024, 256, 0, st>>>(R, elems, 999, 1.f);
    fill_bf16<<<1024, 256, 0, st>>>(Mk, elems, 555, 1.f);
    float *W, *bias, *scale, *shift, *bmean, *binv;
    const long wn = 3L * s.c * s.c;
    CK(hipMalloc(&W, wn * 4));
    CK(hipMalloc(&bias, s.c * 4)); CK(hipMalloc(&scale, s.c * 4)); CK(hipMalloc(&shift, s.c * 4));
    CK(hipMalloc(&bmean, s.c * 4)); CK(hipMalloc(&binv, s.c * 4));
    fill_f32<<<256, 256, 0, st>>>(W, wn, 3, 1.f / sqrtf(3.f * s.c), 0.f);
    fill_f32<<<1, 256, 0, st>>>(bias, s.c, 4, 0.5f, 0.2f);
    fill_f32<<<1, 256, 0, st>>>(scale, s.c, 5, 0.5f, 1.0f);
    fill_f32<<<1, 256, 0, st>>>(shift, s.c, 6, 0.5f, 0.0f);
    fill_f32<<<1, 256, 0, st>>>(bmean, s.c, 7, 0.3f, 0.0f);
    fill_f32<<<1, 256, 0, st>>>(binv, s.c, 8, 0.2f, 1.0f);
    bf16 *wp, *wpd, *wf, *wfd;
    CK(hipMalloc(&wp, wn * 2)); CK(hipMalloc(&wpd, wn * 2)); CK(hipMalloc(&wf, wn * 2)); CK(hipMalloc(&wfd, wn * 2));
    if (emsa_pack_weight_t(EMSA_DT_BF16, W, wp, wpd, s.c, s.c, s.kh, s.kw, s.c, 0, s.c, 0, st) ||
        emsa_pack_weight_frag_t(EMSA_DT_BF16, W, wf, wfd, s.c, s.c, st)) {
      printf("pack failed\n");
      return 2;
    }
    for (int dg = 0; dg < 2; ++dg) {
      EmsaConvGeom g = geom(nb_img, s.h, s.w, s.c, s.kh, s.kw, dg == 1);
      const void* w_old = dg ? wpd : wp;
      const void* w_new = dg ? wfd : wf;
      if (!emsa_conv1d_rs_supported(EMSA_DT_BF16, &g)) {
        printf("%-16s %-6s NOT SUPPORTED by the rs kernel\n", s.name, dg ? "dgrad" : "fwd");
        ++fails;
        continue;
      }
      const int rows_old = emsa_conv_stats_rows_t(EMSA_DT_BF16, &g);
      const int rows_new = emsa_conv1d_rs_stats_rows(EMSA_DT_BF16, &g);
      if (do_check) {
        float *st_old, *st_new;
        CK(hipMalloc(&st_old, 3L * rows_old * s.c * 4));
        CK(hipMalloc(&st_new, 3L * rows_new * s.c * 4));
        // variants: 0 bias+relu, 1 bias+stats, 2 scale/shift+residual+relu, 3 mask (relu backward) + residual
        for (int v = 0; v < 5; ++v) {
          CK(hipMemsetAsync(Y0[0], 0xFF, elems * 2, st));
          CK(hipMemsetAsync(Y1[0], 0xEE, elems * 2, st));
          int r0 = 0, r1 = 0;
          float *p_old = nullptr, *p_new = nullptr;
          if (v == 0) {
            r0 = emsa_conv_igemm_t(EMSA_DT_BF16, &g, X[0], w_old, Y0[0], bias, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, EMSA_ACT_RELU, st);
            r1 = emsa_conv1d_rs_t(EMSA_DT_BF16, &g, X[0], w_new, Y1[0], bias, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, EMSA_ACT_RELU, st);
          } else if (v == 1) {
            r0 = emsa_conv_igemm_t(EMSA_DT_BF16, &g, X[0], w_old, Y0[0], bias, st_old, nullptr, nullptr, nullptr, 0, nullptr, 0, EMSA_ACT_NONE, st);
            r1 = emsa_conv1d_rs_t(EMSA_DT_BF16, &g, X[0], w_new, Y1[0], bias, st_new, nullptr, nullptr, nullptr, 0, nullptr, 0, EMSA_ACT_NONE, st);
          } else if (v == 2) {
            r0 = emsa_conv_igemm_t(EMSA_DT_BF16, &g, X[0], w_old, Y0[0], bias, nullptr, scale, shift, R, s.c, nullptr, 0, EMSA_ACT_RELU, st);
            r1 = emsa_conv1d_rs_t(EMSA_DT_BF16, &g, X[0], w_new, Y1[0], bias, nullptr, scale, shift, R, s.c, nullptr, 0, EMSA_ACT_RELU, st);
          } else if (v == 3) {
            r0 = emsa_conv_igemm_t(EMSA_DT_BF16, &g, X[0], w_old, Y0[0], nullptr, nullptr, nullptr, nullptr, R, s.c, Mk, s.c, EMSA_ACT_NONE, st);
            r1 = emsa_conv1d_rs_t(EMSA_DT_BF16, &g, X[0], w_new, Y1[0], nullptr, nullptr, nullptr, nullptr, R, s.c, Mk, s.c, EMSA_ACT_NONE, st);
          } else {
            CK(hipMalloc(&p_old, 2L * (rows_old + 16) * s.c * 4));
            CK(hipMalloc(&p_new, 2L * (rows_new + 16) * s.c * 4));
            r0 = emsa_conv_igemm_bnb_t(EMSA_DT_BF16, &g, X[0], w_old, Y0[0], R, s.c, Mk, s.c, scale, shift, bmean, binv, p_old, rows_old + 16, st);
            r1 = emsa_conv1d_rs_bnb_t(EMSA_DT_BF16, &g, X[0], w_new, Y1[0], R, s.c, Mk, s.c, scale, shift, bmean, binv, p_new, rows_new + 16, st);
          }
          if (r0 || r1) {
            printf("%-16s %-6s v%d: launch rc old %d new %d\n", s.name, dg ? "dgrad" : "fwd", v, r0, r1);
            ++fails;
            continue;
          }
          CK(hipMemsetAsync(d_max, 0, 4, st));
          CK(hipMemsetAsync(d_bad, 0, 8, st));
          cmp_bf16<<<1024, 256, 0, st>>>(Y1[0], Y0[0], elems, 0.02f, d_max, d_bad);
          float mx;
          unsigned long long bad;
          CK(hipMemcpyAsync(&mx, d_max, 4, hipMemcpyDeviceToHost, st));
          CK(hipMemcpyAsync(&bad, d_bad, 8, hipMemcpyDeviceToHost, st));
          CK(hipStreamSynchronize(st));
          char extra[160] = "";
          bool ok = bad == 0;
          if (v == 1) {
            std::vector<float> h0(3L * rows_old * s.c), h1(3L * rows_new * s.c);
            CK(hipMemcpy(h0.data(), st_old, h0.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h1.data(), st_new, h1.size() * 4, hipMemcpyDeviceToHost));
            std::vector<double> m0, v0, m1, v1;
            merge_stats(h0, rows_old, s.c, m0, v0);
            merge_stats(h1, rows_new, s.c, m1, v1);
            double em = 0, ev = 0;
            for (int c = 0; c < s.c; ++c) {
              em = fmax(em, fabs(m0[c] - m1[c]) / fmax(1e-3, sqrt(v0[c])));
              ev = fmax(ev, fabs(v0[c] - v1[c]) / fmax(1e-6, v0[c]));
            }
            snprintf(extra, sizeof extra, " stats: mean err %.2e sigma, var rel err %.2e", em, ev);
            ok = ok && em < 1e-4 && ev < 1e-4;
          }
          if (v == 4) {
            std::vector<float> h0(2L * (rows_old + 16) * s.c), h1(2L * (rows_new + 16) * s.c);
            CK(hipMemcpy(h0.data(), p_old, h0.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h1.data(), p_new, h1.size() * 4, hipMemcpyDeviceToHost));
            double worst = 0;
            for (int k = 0; k < 2; ++k)
              for (int c = 0; c < s.c; ++c) {
                double a = 0, b = 0, mag = 0;
                for (int r = 0; r < rows_old; ++r) { a += h0[((long)k * (rows_old + 16) + r) * s.c + c]; mag += fabs(h0[((long)k * (rows_old + 16) + r) * s.c + c]); }
                for (int r = 0; r < rows_new; ++r) b += h1[((long)k * (rows_new + 16) + r) * s.c + c];
                worst = fmax(worst, fabs(a - b) / fmax(1.0, mag));
              }
            snprintf(extra, sizeof extra, " bnb sums rel err %.2e", worst);
            ok = ok && worst < 2e-3;      // (the two kernels round g to bf16 identically; sums of ~1e5 terms)
            CK(hipFree(p_old));
            CK(hipFree(p_new));
          }
          printf("%-16s %-6s v%d: max|d| %.4f  bad %llu / %ld%s  %s\n", s.name, dg ? "dgrad" : "fwd", v, mx, bad, elems, extra, ok ? "ok" : "FAIL");
          if (!ok) ++fails;
        }
        CK(hipFree(st_old));
        CK(hipFree(st_new));
      }
      if (do_time && !odd) {
        float t_old = 0, t_new = 0;
        for (int which = 0; which < 2; ++which) {
          const int iters = 40;
          for (int i = -5; i < iters; ++i) {
            if (i == 0) CK(hipEventRecord(e0, st));
            const int k = (i + 5) % sets;
            if (which == 0)
              emsa_conv_igemm_t(EMSA_DT_BF16, &g, X[k], w_old, Y0[k], bias, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, EMSA_ACT_RELU, st);
            else
              emsa_conv1d_rs_t(EMSA_DT_BF16, &g, X[k], w_new, Y1[k], bias, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, EMSA_ACT_RELU, st);
          }
          CK(hipEventRecord(e1, st));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          (which ? t_new : t_old) = ms * 1000.f / iters;
        }
        if (dbg_read) {
          std::vector<long long> tb(8 * 2048);
          if (dbg_read(tb.data(), 8 * 2048) == 0) {
            const int wgs = 2048;
            double ph[8] = {0};
            int used = 0;
            for (int b = 0; b < wgs; ++b) {
              if (tb[b * 8 + 7] <= 0 || tb[b * 8 + 7] > 100000) continue;
              ++used;
              for (int k = 0; k < 8; ++k) ph[k] += (double)tb[b * 8 + k];
            }
            if (used)
              printf("   phases (shader-clock ticks per workgroup, mean of %d; tiles/wg %.1f): prologue %.0f | per tile: wait %.0f dma+addr %.0f mfma %.0f stage %.0f out %.0f | total %.0f\n",
                     used, ph[7] / used, ph[0] / used, ph[1] / ph[7], ph[2] / ph[7], ph[3] / ph[7],
                     ph[4] / ph[7], ph[5] / ph[7], ph[6] / used);
          }
        }
        const double fl = 2.0 * M * s.c * s.c * 3, by = 4.0 * elems + wn * 2.0;
        printf("%-16s %-6s %9.1f %9.1f %8.2f %9.1f %9.2f\n", s.name, dg ? "dgrad" : "fwd", t_old, t_new,
               t_old / t_new, fl / t_new / 1e6, by / t_new / 1e6);
      }
    }
    for (int i = 0; i < sets; ++i) { CK(hipFree(X[i])); CK(hipFree(Y0[i])); CK(hipFree(Y1[i])); }
    CK(hipFree(R)); CK(hipFree(Mk)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(scale)); CK(hipFree(shift));
    CK(hipFree(bmean)); CK(hipFree(binv)); CK(hipFree(wp)); CK(hipFree(wpd)); CK(hipFree(wf)); CK(hipFree(wfd));
  }
  printf("%s (%d failures)\n", fails ? "FAILED" : "ALL OK", fails);
  return fails ? 1 : 0;
}
