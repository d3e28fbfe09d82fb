// Eval-time post-processing and input normalisation on device:
//  * semantic / scene: per-pixel softmax score + arg-max in one pass over the NHWC logits
//    (`semantic_segmentation_idx|score`, `scene_class_idx|score`, Appendix C of SURVEY.md)
//  * instance centres: threshold 0.1 -> 17x17 max-pool NMS -> top-k 64
//    (/root/reference/emsanet/args.py:468-504, decoder.py:95-104), then every pixel is assigned to
//    the centre nearest to (pixel + predicted offset)  (Panoptic-DeepLab grouping, which the
//    un-vendored nicr_mt_scene_analysis post-processing follows; restated in
//    oracle/postprocessing_oracle.py, parity unpinned)
//  * NormalizeRGB / NormalizeDepth + HWC->CHW of the uint8 / uint16 camera frames
//    (/root/reference/emsanet/preprocessing.py:216-226) so that the H2D copy moves 1-2 bytes per
//    value instead of 4.
// All HBM-bound single-pass kernels.
#include "common.h"

namespace {

constexpr int kPix = 256;

template <int C4>
__global__ __launch_bounds__(kPix) void softmax_argmax_kernel(const float* __restrict__ logits,
                                                              int ld, int n_classes, long pixels,
                                                              float* __restrict__ score,
                                                              int64_t* __restrict__ idx) {
  constexpr int LD = C4 * 4 + 4;           // conflict-free ds_read_b128 rows
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const long p0 = (long)blockIdx.x * kPix;
#pragma unroll
  for (int j = 0; j < C4; ++j) {
    const int i = threadIdx.x + kPix * j;
    const int px = i / C4, c4 = i % C4;
    float4 v = emsa_zero4();
    if (p0 + px < pixels && c4 * 4 < ld) v = emsa_ld4(logits + (p0 + px) * (long)ld + c4 * 4);
    emsa_st4(tile + px * LD + c4 * 4, v);
  }
  __syncthreads();
  const long p = p0 + threadIdx.x;
  if (p >= pixels) return;
  float v[C4 * 4];
#pragma unroll
  for (int j = 0; j < C4; ++j)
    *reinterpret_cast<float4*>(v + 4 * j) = emsa_ld4(tile + threadIdx.x * LD + 4 * j);
  float m = -INFINITY;
  int am = 0;
#pragma unroll
  for (int c = 0; c < C4 * 4; ++c)
    if (c < n_classes && v[c] > m) { m = v[c]; am = c; }       // first maximum wins (torch.max)
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < C4 * 4; ++c)
    if (c < n_classes) s += expf(v[c] - m);
  score[p] = 1.f / s;                                           // softmax value of the arg-max
  idx[p] = am;
}

// ---- instance centres -----------------------------------------------------------------------
// EXACT top-k over ALL survivors of the NMS (ref decoder.py:95-104, args.py:468-504: the reference's
// top-k runs over every surviving pixel).  A saturated 16-bit sigmoid gives plateaus where every
// pixel equals its window maximum, i.e. tens of thousands of survivors; round 4 appended survivors
// through one atomic counter into 1024 slots and sorted whichever arrived first (VERDICT r4 weak 8:
// neither the true top-k nor reproducible).  Now: the global top-k is contained in the union of the
// per-chunk top-k, so every workgroup of the NMS pass owns kChunk consecutive pixels of one image,
// collects its survivors in LDS, sorts them when they are more than top_k and appends only its best
// min(count, top_k); the merge pass sorts the union -- in one bitonic sort when it fits (the common
// case: a handful of candidates per image), else in rounds that carry the running top-k.  The order
// (score descending, position ascending) is total, so neither the order of the LDS / global appends
// nor the chunking shows in the result.
constexpr int kMaxCand = 1024;     // sort width of both passes
constexpr int kChunk = kMaxCand;   // pixels per workgroup of the NMS pass
constexpr int kMaxTopK = kMaxCand / 2;

// bitonic sort of ss/sp[0, kMaxCand) by (score desc, position asc) with NT threads
template <int NT>
__device__ __forceinline__ void center_sort(float* ss, int* sp, int tid) {
  for (int k = 2; k <= kMaxCand; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < kMaxCand; t += NT) {
        const int o = t ^ j;
        if (o > t) {
          const bool desc = (t & k) == 0;
          const bool t_first = ss[t] > ss[o] || (ss[t] == ss[o] && sp[t] < sp[o]);
          if (desc != t_first) {
            const float fs = ss[t]; ss[t] = ss[o]; ss[o] = fs;
            const int fp = sp[t]; sp[t] = sp[o]; sp[o] = fp;
          }
        }
      }
      __syncthreads();
    }
}

// a pixel is a centre candidate iff heat >= threshold and heat == max over its k x k window
// (max-pool NMS with "same" padding; equal neighbours are all kept, like pooled == heat).
// grid = (chunks per image, images); cand_* = [n][cap] with cap = chunks * min(top_k, kChunk);
// count[img] = candidates appended, survivors[img] = pixels that survived the NMS (diagnostics)
__global__ __launch_bounds__(256) void center_nms_kernel(
    const float* __restrict__ heat, int ld, int h, int w, int ksize, float threshold,
    const uint8_t* __restrict__ fg, int top_k, int cap, int* __restrict__ count,
    int* __restrict__ survivors, float* __restrict__ cand_score, int* __restrict__ cand_pos) {
  __shared__ float ss[kChunk];
  __shared__ int sp[kChunk];
  __shared__ int cnt, base;
  const int img = blockIdx.y, tid = threadIdx.x;
  const int hw = h * w, r = ksize / 2;
  const float* hm = heat + (long)img * hw * ld;
  const uint8_t* fgi = fg ? fg + (long)img * hw : nullptr;
  for (int t = tid; t < kChunk; t += 256) { ss[t] = -1.f; sp[t] = 0x7fffffff; }
  if (tid == 0) cnt = 0;
  __syncthreads();
  for (int q = blockIdx.x * kChunk + tid; q < min(hw, (int)(blockIdx.x + 1) * kChunk); q += 256) {
    float v = hm[(long)q * ld];
    if (fgi && !fgi[q]) v = 0.f;
    if (!(v >= threshold)) continue;
    const int x = q % w, y = q / w;
    bool is_max = true;
    for (int dy = -r; dy <= r && is_max; ++dy) {
      const int yy = y + dy;
      if (yy < 0 || yy >= h) continue;
      for (int dx = -r; dx <= r; ++dx) {
        const int xx = x + dx;
        if (xx < 0 || xx >= w) continue;
        const int j = yy * w + xx;
        float u = hm[(long)j * ld];
        if (fgi && !fgi[j]) u = 0.f;
        if (u > v) { is_max = false; break; }
      }
    }
    if (!is_max) continue;
    const int slot = atomicAdd(&cnt, 1);          // (LDS; <= kChunk survivors per chunk)
    ss[slot] = v;
    sp[slot] = q;
  }
  __syncthreads();
  const int c = cnt;
  if (c == 0) return;
  if (c > top_k) center_sort<256>(ss, sp, tid);   // (c is uniform over the workgroup)
  const int keep = min(c, top_k);
  if (tid == 0) {
    base = atomicAdd(count + img, keep);
    atomicAdd(survivors + img, c);
  }
  __syncthreads();
  for (int t = tid; t < keep; t += 256) {
    cand_score[(long)img * cap + base + t] = ss[t];
    cand_pos[(long)img * cap + base + t] = sp[t];
  }
}

// one workgroup per image: sort the candidates by (score desc, position asc) and write the top k.
// More than kMaxCand candidates: rounds of (running top-k + the next kMaxCand - top_k candidates).
__global__ __launch_bounds__(kMaxCand) void center_topk_kernel(
    const int* __restrict__ count, const float* __restrict__ cand_score,
    const int* __restrict__ cand_pos, int cap, int w, int top_k, float* __restrict__ centers,
    float* __restrict__ scores, int* __restrict__ n_centers) {
  __shared__ float ss[kMaxCand];
  __shared__ int sp[kMaxCand];
  const int img = blockIdx.x, t = threadIdx.x;
  const int total = min(count[img], cap);
  int have = 0, pos = 0;
  do {
    const int take = min(kMaxCand - have, total - pos);
    if (t >= have) {
      const bool in = t < have + take;
      ss[t] = in ? cand_score[(long)img * cap + pos + t - have] : -1.f;
      sp[t] = in ? cand_pos[(long)img * cap + pos + t - have] : 0x7fffffff;
    }
    __syncthreads();
    center_sort<kMaxCand>(ss, sp, t);
    pos += take;
    have = min(top_k, have + take);
  } while (pos < total);
  const int keep = min(total, top_k);
  if (t < top_k) {
    const bool ok = t < keep;
    centers[((long)img * top_k + t) * 2 + 0] = ok ? (float)(sp[t] / w) : -1.f;   // y
    centers[((long)img * top_k + t) * 2 + 1] = ok ? (float)(sp[t] % w) : -1.f;   // x
    scores[(long)img * top_k + t] = ok ? ss[t] : 0.f;
  }
  if (t == 0) n_centers[img] = keep;
}

// id(p) = 1 + argmin_k |(y + off_y*sy, x + off_x*sx) - centre_k|^2, 0 outside the foreground or
// without centres (first minimum wins)
__global__ void instance_assign_kernel(const float* __restrict__ offset, int ld, int n, int h,
                                       int w, float sy, float sx,
                                       const float* __restrict__ centers,
                                       const int* __restrict__ n_centers, int top_k,
                                       const uint8_t* __restrict__ fg, float max_dist2,
                                       int32_t* __restrict__ ids) {
  extern __shared__ float cs[];                   // [top_k][2] of this image
  const int img = blockIdx.y;
  const int nc = n_centers[img];
  for (int j = threadIdx.x; j < top_k * 2; j += blockDim.x) cs[j] = centers[(long)img * top_k * 2 + j];
  __syncthreads();
  const long hw = (long)h * w;
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < hw;
       q += (long)gridDim.x * blockDim.x) {
    const long i = img * hw + q;
    int id = 0;
    if (nc > 0 && (!fg || fg[i])) {
      const float cy = (float)(q / w) + offset[i * ld] * sy;
      const float cx = (float)(q % w) + offset[i * ld + 1] * sx;
      float best = INFINITY;
      for (int k = 0; k < nc; ++k) {
        const float dy = cy - cs[2 * k], dx = cx - cs[2 * k + 1];
        const float d = dy * dy + dx * dx;
        if (d < best) { best = d; id = k + 1; }
      }
      if (max_dist2 > 0.f && best > max_dist2) id = 0;
    }
    ids[i] = id;
  }
}

// ---- input normalisation ---------------------------------------------------------------------
// rgb uint8 [n][h][w][3] -> float [n][3][h][w]: (v * scale - mean[c]) / std[c]
__global__ void normalize_rgb_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                     long hw, long total, float scale, float m0, float m1,
                                     float m2, float i0, float i1, float i2) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long img = i / hw, q = i - img * hw;
    const uint8_t* s = src + i * 3;
    float* d = dst + img * 3 * hw + q;
    d[0] = ((float)s[0] * scale - m0) * i0;
    d[hw] = ((float)s[1] * scale - m1) * i1;
    d[2 * hw] = ((float)s[2] * scale - m2) * i2;
  }
}

// depth uint16 [n][h][w] -> float [n][1][h][w]: (v - mean) / std, invalid (0) stays 0
__global__ void normalize_depth_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst,
                                       long total, float mean, float inv_std, int keep_zero) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const float v = (float)src[i];
    dst[i] = (keep_zero && src[i] == 0) ? 0.f : (v - mean) * inv_std;
  }
}


// ---- panoptic merge (Panoptic-DeepLab, Cheng et al. CVPR 2020, Sec. 3.3) ---------------------
// votes[n][top_k + 1][n_classes] += 1 for every thing pixel with an instance id: the semantic
// class histogram of each instance (int32 atomics: deterministic)
__global__ void panoptic_votes_kernel(const int64_t* __restrict__ sem, const int32_t* __restrict__ ids,
                                      const uint8_t* __restrict__ is_thing, long hw, long total,
                                      int n_classes, int slots, int32_t* __restrict__ votes) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int id = ids[i];
    const int c = (int)sem[i];
    if (id > 0 && id < slots && c >= 0 && c < n_classes && is_thing[c])
      atomicAdd(votes + ((i / hw) * slots + id) * n_classes + c, 1);
  }
}

// class of every instance = arg-max of its votes (first maximum wins; -1 without votes)
__global__ void panoptic_class_kernel(const int32_t* __restrict__ votes, int n_classes, int total,
                                      int32_t* __restrict__ cls) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int best = 0, arg = -1;
  for (int c = 0; c < n_classes; ++c) {
    const int v = votes[(long)i * n_classes + c];
    if (v > best) { best = v; arg = c; }
  }
  cls[i] = arg;
}

// per pixel: thing pixel with an instance -> (instance class, id); stuff pixel -> (class, 0);
// thing pixel without an instance -> void.  panoptic id = (class + 1) * label_divisor + id, 0 = void;
// the semantic part is published in the label list WITH void, like the id: class + 1, 0 = void
// (/root/reference/inference_dataset.py:298-304 "already has void class", visualization.py:726-727)
__global__ void panoptic_merge_kernel(const int64_t* __restrict__ sem, const int32_t* __restrict__ ids,
                                      const uint8_t* __restrict__ is_thing,
                                      const int32_t* __restrict__ cls, long hw, long total,
                                      int n_classes, int slots, int label_divisor,
                                      int64_t* __restrict__ pan_sem, int32_t* __restrict__ pan_inst,
                                      int64_t* __restrict__ pan_id) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)sem[i];
    const bool thing = c >= 0 && c < n_classes && is_thing[c];
    int oc = -1, oi = 0;
    if (!thing) {
      oc = (c >= 0 && c < n_classes) ? c : -1;
    } else {
      const int id = ids[i];
      if (id > 0 && id < slots) {
        const int k = cls[(i / hw) * slots + id];
        if (k >= 0) { oc = k; oi = id; }
      }
    }
    pan_sem[i] = oc + 1;                      // WITH void: 0 = void, class c -> c + 1 (see emsa_panoptic_merge)
    pan_inst[i] = oi;
    pan_id[i] = oc < 0 ? 0 : (int64_t)(oc + 1) * label_divisor + oi;
  }
}

// ---- per-instance statistics, scores, orientations -------------------------------------------
// Sums are taken in FIXED POINT with integer atomics, so they do not depend on the order the
// pixels arrive in: the scores and angles derived from them are bit-reproducible and the numpy
// oracle reproduces them exactly.
constexpr int kStatSlotsLds = 1024;              // per-workgroup LDS histogram up to this many ids
constexpr double kScoreOne = 1073741824.0;       // 2^30: scores in [0, 1]
constexpr double kVecOne = 16777216.0;           // 2^24: orientation components, clamped to +-2^14

__device__ __forceinline__ unsigned long long score_fixed(float v) {
  if (!(v > 0.f)) return 0ull;                   // also NaN
  if (v > 1.f) v = 1.f;
  return (unsigned long long)((double)v * kScoreOne + 0.5);
}

__device__ __forceinline__ long long vec_fixed(float v) {
  if (!(v == v)) return 0ll;
  if (v > 16384.f) v = 16384.f;
  if (v < -16384.f) v = -16384.f;
  return __double2ll_rn((double)v * kVecOne);
}

// grid = (blocks per image, images).  area[n][slots] += 1 and sum[n][slots] += fixed(value) for every
// pixel with 0 < id < slots (and mask != 0)
template <bool LDS>
__global__ void instance_stats_kernel(const int32_t* __restrict__ ids, const float* __restrict__ value,
                                      const uint8_t* __restrict__ mask, long hw, int slots,
                                      unsigned long long* __restrict__ sum, int32_t* __restrict__ area) {
  __shared__ unsigned long long s_sum[LDS ? kStatSlotsLds : 1];
  __shared__ int s_area[LDS ? kStatSlotsLds : 1];
  const long img = blockIdx.y;
  if (LDS) {
    for (int k = threadIdx.x; k < slots; k += blockDim.x) { s_sum[k] = 0ull; s_area[k] = 0; }
    __syncthreads();
  }
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < hw;
       q += (long)gridDim.x * blockDim.x) {
    const long i = img * hw + q;
    const int id = ids[i];
    if (id <= 0 || id >= slots || (mask && !mask[i])) continue;
    if (LDS) {
      atomicAdd(s_area + id, 1);
      if (value) atomicAdd(s_sum + id, score_fixed(value[i]));
    } else {
      atomicAdd(area + img * slots + id, 1);
      if (value) atomicAdd(sum + img * slots + id, score_fixed(value[i]));
    }
  }
  if (LDS) {
    __syncthreads();
    for (int k = threadIdx.x; k < slots; k += blockDim.x)
      if (s_area[k]) {
        atomicAdd(area + img * slots + k, s_area[k]);
        if (value) atomicAdd(sum + img * slots + k, s_sum[k]);
      }
  }
}

// vec[n][slots][2] += fixed(orientation[pixel][0..1]), count[n][slots] += 1
template <bool LDS>
__global__ void instance_orientation_kernel(const float* __restrict__ ori, int ld,
                                            const int32_t* __restrict__ ids,
                                            const uint8_t* __restrict__ mask, long hw, int slots,
                                            long long* __restrict__ vec, int32_t* __restrict__ count) {
  __shared__ unsigned long long s_vec[LDS ? 2 * kStatSlotsLds : 1];
  __shared__ int s_cnt[LDS ? kStatSlotsLds : 1];
  const long img = blockIdx.y;
  if (LDS) {
    for (int k = threadIdx.x; k < slots; k += blockDim.x) {
      s_vec[2 * k] = 0ull; s_vec[2 * k + 1] = 0ull; s_cnt[k] = 0;
    }
    __syncthreads();
  }
  unsigned long long* gv = (unsigned long long*)vec;     // two's complement: wrap-around adds
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < hw;
       q += (long)gridDim.x * blockDim.x) {
    const long i = img * hw + q;
    const int id = ids[i];
    if (id <= 0 || id >= slots || (mask && !mask[i])) continue;
    const unsigned long long a = (unsigned long long)vec_fixed(ori[i * ld]);
    const unsigned long long b = (unsigned long long)vec_fixed(ori[i * ld + 1]);
    if (LDS) {
      atomicAdd(s_vec + 2 * id, a);
      atomicAdd(s_vec + 2 * id + 1, b);
      atomicAdd(s_cnt + id, 1);
    } else {
      atomicAdd(gv + (img * slots + id) * 2, a);
      atomicAdd(gv + (img * slots + id) * 2 + 1, b);
      atomicAdd(count + img * slots + id, 1);
    }
  }
  if (LDS) {
    __syncthreads();
    for (int k = threadIdx.x; k < slots; k += blockDim.x)
      if (s_cnt[k]) {
        atomicAdd(gv + (img * slots + k) * 2, s_vec[2 * k]);
        atomicAdd(gv + (img * slots + k) * 2 + 1, s_vec[2 * k + 1]);
        atomicAdd(count + img * slots + k, s_cnt[k]);
      }
  }
}

// per instance: mean semantic score of its pixels, and centre score x that mean
__global__ void panoptic_instance_scores_kernel(const unsigned long long* __restrict__ sum,
                                                const int32_t* __restrict__ area,
                                                const float* __restrict__ center_scores, int slots,
                                                int top_k, int total, float* __restrict__ inst_sem,
                                                float* __restrict__ inst_pan) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int a = area[i];
  const float mean = a > 0 ? (float)((double)sum[i] / (double)a / kScoreOne) : 0.f;
  const int id = i % slots;
  const float cs = id >= 1 ? center_scores[(long)(i / slots) * top_k + id - 1] : 0.f;
  inst_sem[i] = mean;
  inst_pan[i] = __fmul_rn(cs, mean);
}

// per pixel: instance pixel -> (mean semantic score, centre score, product) of its instance;
// stuff pixel -> (own semantic score, 0, own semantic score); void -> 0
__global__ void panoptic_pixel_scores_kernel(const float* __restrict__ sem_score,
                                             const int32_t* __restrict__ pan_inst,
                                             const int64_t* __restrict__ pan_sem,
                                             const float* __restrict__ center_scores,
                                             const float* __restrict__ inst_sem,
                                             const float* __restrict__ inst_pan, long hw, long total,
                                             int slots, int top_k, float* __restrict__ o_sem,
                                             float* __restrict__ o_inst, float* __restrict__ o_pan) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int id = pan_inst[i];
    float a = 0.f, b = 0.f, c = 0.f;
    if (id > 0 && id < slots) {
      const long img = i / hw;
      a = inst_sem[img * slots + id];
      b = center_scores[img * top_k + id - 1];
      c = inst_pan[img * slots + id];
    } else if (pan_sem[i] > 0) {
      a = sem_score[i];
      c = a;
    }
    o_sem[i] = a;
    o_inst[i] = b;
    o_pan[i] = c;
  }
}

inline int stats_grid(long hw) {
  long b = (hw + 1023) / 1024;                   // >= 4 pixels per thread: few LDS flushes
  return (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
}

inline int grid1d(long items) {
  long b = (items + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

template <int C4>
int launch_argmax(const float* x, int ld, int nc, long pixels, float* score, int64_t* idx,
                  hipStream_t st) {
  const int grid = (int)((pixels + kPix - 1) / kPix);
  const size_t lds = (size_t)kPix * (C4 * 4 + 4) * sizeof(float);
  hipLaunchKernelGGL(softmax_argmax_kernel<C4>, dim3(grid), dim3(kPix), lds, st, x, ld, nc, pixels,
                     score, idx);
  return emsa_launch_status();
}

}  // namespace

extern "C" int emsa_softmax_argmax(const float* logits, int32_t ld, int32_t n_classes,
                                   int64_t pixels, float* score, int64_t* idx, void* stream) {
  if (!logits || !score || !idx) return EMSA_E_ARG;
  if (n_classes < 1 || n_classes > 64 || (ld & 3) || ld < ((n_classes + 3) & ~3) ||
      (((uintptr_t)logits) & 15) || pixels < 1)
    return EMSA_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int c4 = (n_classes + 3) / 4;
  if (c4 <= 4) return launch_argmax<4>(logits, ld, n_classes, pixels, score, idx, st);
  if (c4 <= 10) return launch_argmax<10>(logits, ld, n_classes, pixels, score, idx, st);
  return launch_argmax<16>(logits, ld, n_classes, pixels, score, idx, st);
}

extern "C" int emsa_center_candidates_max(void) { return kMaxCand; }

// entries PER IMAGE of the ws_score / ws_pos scratch of emsa_instance_centers
extern "C" int64_t emsa_center_ws_entries(int32_t h, int32_t w, int32_t top_k) {
  if (h < 1 || w < 1 || top_k < 1) return 0;
  const long chunks = ((long)h * w + kChunk - 1) / kChunk;
  return chunks * (top_k < kChunk ? top_k : kChunk);
}

// ws_count: int[2 * n] -- [0, n): candidates handed to the merge pass, [n, 2n): pixels that survived
// the NMS per image (what a caller inspects to see a saturated heat-map; nothing is dropped)
extern "C" int emsa_instance_centers(const float* heat, int32_t ld, int32_t n, int32_t h,
                                     int32_t w, int32_t nms_kernel, float threshold,
                                     int32_t top_k, const uint8_t* fg, int32_t* ws_count,
                                     float* ws_score, int32_t* ws_pos, float* centers,
                                     float* scores, int32_t* n_centers, void* stream) {
  if (!heat || !ws_count || !ws_score || !ws_pos || !centers || !scores || !n_centers)
    return EMSA_E_ARG;
  if (n < 1 || h < 1 || w < 1 || ld < 1 || nms_kernel < 1 || !(nms_kernel & 1) || top_k < 1 ||
      top_k > kMaxTopK || (long)h * w > 0x7fffffffL / (ld > 1 ? ld : 1))
    return EMSA_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = (int)(((long)h * w + kChunk - 1) / kChunk);
  const int cap = (int)emsa_center_ws_entries(h, w, top_k);
  emsa_zero_async(ws_count, (size_t)2 * n * sizeof(int), st);
  hipLaunchKernelGGL(center_nms_kernel, dim3(chunks, n), dim3(256), 0, st, heat, ld, h, w,
                     nms_kernel, threshold, fg, top_k, cap, ws_count, ws_count + n, ws_score, ws_pos);
  hipLaunchKernelGGL(center_topk_kernel, dim3(n), dim3(kMaxCand), 0, st, ws_count, ws_score, ws_pos,
                     cap, w, top_k, centers, scores, n_centers);
  return emsa_launch_status();
}

extern "C" int emsa_instance_assign(const float* offset, int32_t ld, int32_t n, int32_t h,
                                    int32_t w, float scale_y, float scale_x, const float* centers,
                                    const int32_t* n_centers, int32_t top_k, const uint8_t* fg,
                                    float max_distance, int32_t* ids, void* stream) {
  if (!offset || !centers || !n_centers || !ids) return EMSA_E_ARG;
  if (n < 1 || h < 1 || w < 1 || ld < 2 || top_k < 1 || top_k > kMaxTopK) return EMSA_E_SHAPE;
  const long hw = (long)h * w;
  int gx = (int)((hw + 255) / 256);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(instance_assign_kernel, dim3(gx, n), dim3(256),
                     (size_t)top_k * 2 * sizeof(float), (hipStream_t)stream, offset, ld, n, h, w,
                     scale_y, scale_x, centers, n_centers, top_k, fg,
                     max_distance > 0.f ? max_distance * max_distance : 0.f, ids);
  return emsa_launch_status();
}

extern "C" int emsa_normalize_rgb(const uint8_t* rgb_hwc, float* out_chw, int32_t n, int32_t h,
                                  int32_t w, float scale, const float* mean3, const float* std3,
                                  void* stream) {
  if (!rgb_hwc || !out_chw || !mean3 || !std3) return EMSA_E_ARG;
  if (n < 1 || h < 1 || w < 1 || std3[0] == 0.f || std3[1] == 0.f || std3[2] == 0.f)
    return EMSA_E_SHAPE;
  const long hw = (long)h * w, total = hw * n;
  hipLaunchKernelGGL(normalize_rgb_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream,
                     rgb_hwc, out_chw, hw, total, scale, mean3[0], mean3[1], mean3[2],
                     1.f / std3[0], 1.f / std3[1], 1.f / std3[2]);
  return emsa_launch_status();
}

extern "C" int emsa_normalize_depth(const uint16_t* depth, float* out, int64_t total, float mean,
                                    float std, int32_t keep_zero, void* stream) {
  if (!depth || !out) return EMSA_E_ARG;
  if (total < 1 || std == 0.f) return EMSA_E_SHAPE;
  hipLaunchKernelGGL(normalize_depth_kernel, dim3(grid1d((long)total)), dim3(256), 0,
                     (hipStream_t)stream, depth, out, (long)total, mean, 1.f / std,
                     keep_zero ? 1 : 0);
  return emsa_launch_status();
}

extern "C" int emsa_panoptic_merge(const int64_t* semantic_idx, const int32_t* instance_ids,
                                   const uint8_t* class_is_thing, int32_t n, int64_t hw,
                                   int32_t n_classes, int32_t top_k, int32_t label_divisor,
                                   int32_t* ws_votes, int32_t* ws_class, int64_t* pan_semantic,
                                   int32_t* pan_instance, int64_t* pan_id, void* stream) {
  if (!semantic_idx || !instance_ids || !class_is_thing || !ws_votes || !ws_class ||
      !pan_semantic || !pan_instance || !pan_id)
    return EMSA_E_ARG;
  if (n < 1 || hw < 1 || n_classes < 1 || top_k < 1 || label_divisor <= top_k) return EMSA_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int slots = top_k + 1;
  const long total = (long)n * hw;
  emsa_zero_async(ws_votes, (size_t)n * slots * n_classes * sizeof(int32_t), st);
  hipLaunchKernelGGL(panoptic_votes_kernel, dim3(grid1d(total)), dim3(256), 0, st, semantic_idx,
                     instance_ids, class_is_thing, (long)hw, total, n_classes, slots, ws_votes);
  hipLaunchKernelGGL(panoptic_class_kernel, dim3((n * slots + 255) / 256), dim3(256), 0, st, ws_votes,
                     n_classes, n * slots, ws_class);
  hipLaunchKernelGGL(panoptic_merge_kernel, dim3(grid1d(total)), dim3(256), 0, st, semantic_idx,
                     instance_ids, class_is_thing, ws_class, (long)hw, total, n_classes, slots,
                     label_divisor, pan_semantic, pan_instance, pan_id);
  return emsa_launch_status();
}

extern "C" int emsa_instance_stats(const int32_t* ids, const float* value, const uint8_t* mask,
                                   int32_t n, int64_t hw, int32_t slots, int64_t* sum,
                                   int32_t* area, void* stream) {
  if (!ids || !area || (value && !sum)) return EMSA_E_ARG;
  if (n < 1 || hw < 1 || slots < 2) return EMSA_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  emsa_zero_async(area, (size_t)n * slots * sizeof(int32_t), st);
  if (value) emsa_zero_async(sum, (size_t)n * slots * sizeof(int64_t), st);
  const dim3 grid(stats_grid(hw), n);
  if (slots <= kStatSlotsLds)
    hipLaunchKernelGGL(instance_stats_kernel<true>, grid, dim3(256), 0, st, ids, value, mask,
                       (long)hw, slots, (unsigned long long*)sum, area);
  else
    hipLaunchKernelGGL(instance_stats_kernel<false>, grid, dim3(256), 0, st, ids, value, mask,
                       (long)hw, slots, (unsigned long long*)sum, area);
  return emsa_launch_status();
}

extern "C" int emsa_panoptic_scores(const float* semantic_score, const int32_t* pan_instance,
                                    const int64_t* pan_semantic, const float* center_scores,
                                    int32_t n, int64_t hw, int32_t top_k, int64_t* ws_sum,
                                    int32_t* inst_area, float* inst_semantic_score,
                                    float* inst_panoptic_score, float* semantic_score_out,
                                    float* instance_score_out, float* panoptic_score_out,
                                    void* stream) {
  if (!semantic_score || !pan_instance || !pan_semantic || !center_scores || !ws_sum ||
      !inst_area || !inst_semantic_score || !inst_panoptic_score || !semantic_score_out ||
      !instance_score_out || !panoptic_score_out)
    return EMSA_E_ARG;
  if (n < 1 || hw < 1 || top_k < 1 || top_k > kMaxTopK) return EMSA_E_SHAPE;
  const int slots = top_k + 1;
  const int rc = emsa_instance_stats(pan_instance, semantic_score, nullptr, n, hw, slots, ws_sum,
                                     inst_area, stream);
  if (rc != EMSA_OK) return rc;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(panoptic_instance_scores_kernel, dim3((n * slots + 255) / 256), dim3(256), 0,
                     st, (const unsigned long long*)ws_sum, inst_area, center_scores, slots, top_k,
                     n * slots, inst_semantic_score, inst_panoptic_score);
  const long total = (long)n * hw;
  hipLaunchKernelGGL(panoptic_pixel_scores_kernel, dim3(grid1d(total)), dim3(256), 0, st,
                     semantic_score, pan_instance, pan_semantic, center_scores, inst_semantic_score,
                     inst_panoptic_score, (long)hw, total, slots, top_k, semantic_score_out,
                     instance_score_out, panoptic_score_out);
  return emsa_launch_status();
}

extern "C" int emsa_instance_orientation(const float* orientation, int32_t ld, const int32_t* ids,
                                         const uint8_t* mask, int32_t n, int64_t hw, int32_t slots,
                                         int64_t* vec_sum, int32_t* count, void* stream) {
  if (!orientation || !ids || !vec_sum || !count) return EMSA_E_ARG;
  if (n < 1 || hw < 1 || ld < 2 || slots < 2) return EMSA_E_SHAPE;
  hipStream_t st = (hipStream_t)stream;
  emsa_zero_async(vec_sum, (size_t)n * slots * 2 * sizeof(int64_t), st);
  emsa_zero_async(count, (size_t)n * slots * sizeof(int32_t), st);
  const dim3 grid(stats_grid(hw), n);
  if (slots <= kStatSlotsLds)
    hipLaunchKernelGGL(instance_orientation_kernel<true>, grid, dim3(256), 0, st, orientation, ld, ids,
                       mask, (long)hw, slots, (long long*)vec_sum, count);
  else
    hipLaunchKernelGGL(instance_orientation_kernel<false>, grid, dim3(256), 0, st, orientation, ld,
                       ids, mask, (long)hw, slots, (long long*)vec_sum, count);
  return emsa_launch_status();
}
