// Inference post-processing of the GFL / LD head on gfx950: everything between
// the head's raw maps and the (k, 5) detections, on the device, NCHW-direct.
//
// Replaces (reference file:line)
//   GFLHead._get_bboxes             dense_heads/gfl_head.py:354-451
//     sigmoid scores, Integral * stride, per-level top-nms_pre by max class
//     score, distance2bbox (core/bbox/transforms.py:119-156) + clamp, rescale
//   multiclass_nms (type 'nms')     core/post_processing/bbox_nms.py:70-195
//   mmcv.ops.batched_nms / nms      (mmcv-full 1.2.x compiled op; greedy NMS on
//     class-shifted boxes, IoU > thr suppresses; its >= 10000-box per-class
//     branch yields the same detections as the single pass run here)
//
// HBM/latency-bound integer + fp32 work, no GEMM shape anywhere:
//   1. keys      one thread per (image, level, anchor): max class logit ->
//                64-bit key (score bits << 32 | ~anchor) so ONE descending sort
//                orders by score and breaks ties lower-index-first
//   2. top-k     per (image, level) with more than nms_pre anchors: radix select
//                of the nms_pre-th key + LDS sort of the selected keys (one
//                workgroup; a global-memory bitonic sort is the fallback)
//   3. decode    one thread per selected anchor: 80 sigmoid scores, 4 x 17-bin
//                softmax expectations, box, clamp, rescale; (anchor, class)
//                pairs above score_thr are appended (atomic counter) as keys
//                (score bits << 32 | ~pair index): order fixed by the next sort
//   4. sort      the 4096 best candidate keys per image (radix select + LDS
//                sort); the full bitonic sort only if NMS runs out of them
//   5. nms       one workgroup per image walks the sorted candidates in chunks
//                of 256 against the kept list (<= max_per_img, in LDS) and
//                stops at max_per_img: O(candidates examined x kept), not
//                O(candidates^2)
#include <hip/hip_runtime.h>

#include "ld_launch.h"
#include <stdint.h>
#include <stdlib.h>

#include "../../include/ld_hip.h"
#include "ld_math.h"

namespace {

using ld::sigmoidf_;
using ld::softmax_expect;

constexpr int kSortThreads = 1024;
constexpr int kNmsThreads = 256;
constexpr int kMaxKeep = 1024;  // max_per_img supported by the LDS kept list

struct Plan {
  int N, L, C, Ktot;
  int H[LD_MAX_LEVELS], W[LD_MAX_LEVELS], stride[LD_MAX_LEVELS];
  int A[LD_MAX_LEVELS];      // anchors of the level
  int K[LD_MAX_LEVELS];      // selected anchors of the level
  int koff[LD_MAX_LEVELS];   // first slot of the level among the Ktot slots
  int pad[LD_MAX_LEVELS];    // pow2 sort length, 0 = level is not sorted
  int keyoff[LD_MAX_LEVELS]; // first key of the level inside an image's keys
  int keys_per_img;
  int cand_cap;              // candidate capacity per image (pow2)
  int prob;                  // the class maps hold probabilities (GFocalHead)
  int has_ctr;               // score factor sigmoid(centerness) (ATSS / FCOS heads)
  int points;                // FCOS points (x, y) * s + s / 2 instead of anchor centres
  int B;                     // anchors per cell (RetinaGFLHead: rows = cell * B + b,
                             // channels = b * C + c / b * 68 + side * 17 + k)
};

inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

__device__ __forceinline__ unsigned long long make_key(float score, unsigned idx) {
  // score > 0: its bit pattern is monotonic; ~idx makes the lower index win ties
  return ((unsigned long long)__float_as_uint(score) << 32) |
         (unsigned long long)(0xFFFFFFFFu - idx);
}

// ---- 1. keys ---------------------------------------------------------------
__global__ __launch_bounds__(256) void infer_keys_kernel(Plan p, ld_maps_t cls,
                                                        ld_maps_t ctr,
                                                        unsigned long long* keys) {
  const int l = blockIdx.y, n = blockIdx.z;
  if (p.pad[l] == 0) return;
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= p.pad[l]) return;
  unsigned long long key = 0ull;
  if (a < p.A[l]) {
    const int cell = a / p.B, b = a - cell * p.B;
    const float* base = cls.ptr[l] + (size_t)n * cls.stride_n[l] +
                        (size_t)b * p.C * cls.stride_c[l] + cell;
    float m = base[0];
    for (int c = 1; c < p.C; ++c) m = fmaxf(m, base[(size_t)c * cls.stride_c[l]]);
    float sc = p.prob ? m : sigmoidf_(m);
    // (scores * centerness[..., None]).max(-1), atss_gfl_head.py:509: the
    // product is monotonic in the score, so the max commutes with it
    if (p.has_ctr)
      sc = sc * sigmoidf_(ctr.ptr[l][(size_t)n * ctr.stride_n[l] + cell]);
    key = make_key(sc, (unsigned)a);
  }
  keys[(size_t)n * p.keys_per_img + p.keyoff[l] + a] = key;
}

// ---- 2./4. bitonic sort, descending, one workgroup per segment ---------------
__device__ void bitonic_desc(unsigned long long* a, int n) {
  for (int k = 2; k <= n; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long x = a[i], y = a[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) {
            a[i] = y;
            a[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  }
}

// Workgroup-wide: the min(k, n) largest of in[0..n) in descending order -> out
// (k <= kSelN).  n <= kSelN: stage everything in LDS and sort there.  Otherwise
// an MSB-first radix select over the eight key bytes finds the k-th largest key
// (keys are unique: score bits | ~index), the k keys >= it are gathered into
// LDS and sorted there -- 8 counting passes + one 4096-wide LDS sort instead of
// ~120 bitonic passes over global memory.  `out` may alias `in`: every read of
// `in` precedes the first write of `out`.
constexpr int kSelN = 4096;

struct SelectLds {
  unsigned long long keys[kSelN];
  int hist[256];
  unsigned long long prefix;
  int rem, cnt;
};

__device__ int select_sort_desc(const unsigned long long* in, int n, int k,
                                unsigned long long* out, SelectLds& s) {
  const int T = blockDim.x, t = threadIdx.x;
  int m;
  if (n <= kSelN) {
    for (int i = t; i < n; i += T) s.keys[i] = in[i];
    m = n;
  } else {
    if (t == 0) {
      s.prefix = 0ull;
      s.rem = k;
    }
    __syncthreads();
    for (int b = 7; b >= 0; --b) {
      for (int i = t; i < 256; i += T) s.hist[i] = 0;
      __syncthreads();
      const unsigned long long prefix = s.prefix;
      const int shift = 8 * b;
      const unsigned long long himask = b == 7 ? 0ull : (~0ull << (8 * (b + 1)));
      for (int i = t; i < n; i += T) {
        const unsigned long long key = in[i];
        if ((key & himask) == (prefix & himask))
          atomicAdd(&s.hist[(int)((key >> shift) & 0xFFull)], 1);
      }
      __syncthreads();
      if (t == 0) {
        // largest byte value v with #(byte > v) < rem <= #(byte >= v)
        int cum = 0, v = 255;
        const int rem = s.rem;
        for (; v > 0; --v) {
          if (cum + s.hist[v] >= rem) break;
          cum += s.hist[v];
        }
        s.prefix = prefix | ((unsigned long long)v << shift);
        s.rem = rem - cum;
      }
      __syncthreads();
    }
    const unsigned long long kth = s.prefix;
    if (t == 0) s.cnt = 0;
    __syncthreads();
    for (int i = t; i < n; i += T) {
      const unsigned long long key = in[i];
      if (key >= kth) {
        const int pos = atomicAdd(&s.cnt, 1);
        if (pos < kSelN) s.keys[pos] = key;
      }
    }
    __syncthreads();
    m = min(s.cnt, kSelN);
  }
  int len = 2;
  while (len < m) len <<= 1;
  for (int i = m + t; i < len; i += T) s.keys[i] = 0ull;
  __syncthreads();
  bitonic_desc(s.keys, len);
  const int w = min(k, m);
  for (int i = t; i < w; i += T) out[i] = s.keys[i];
  __syncthreads();
  return w;
}

__global__ __launch_bounds__(kSortThreads) void infer_topk_select_kernel(
    Plan p, unsigned long long* keys) {
  __shared__ SelectLds s;
  int l = -1, seen = 0;
  for (int i = 0; i < p.L; ++i)
    if (p.pad[i] > 0) {
      if (seen == (int)blockIdx.x) l = i;
      ++seen;
    }
  if (l < 0) return;
  unsigned long long* seg = keys + (size_t)blockIdx.y * p.keys_per_img + p.keyoff[l];
  select_sort_desc(seg, p.A[l], p.K[l], seg, s);
}

// the kSelN best candidates of an image, sorted, into their own buffer (the
// full list stays intact for the rare fallback)
__global__ __launch_bounds__(kSortThreads) void infer_cand_select_kernel(
    Plan p, const unsigned long long* cand, const int* cand_count,
    unsigned long long* cand_top) {
  __shared__ SelectLds s;
  const int n = blockIdx.x;
  const int M = min(cand_count[n], p.cand_cap);
  select_sort_desc(cand + (size_t)n * p.cand_cap, M, kSelN,
                   cand_top + (size_t)n * kSelN, s);
}

__global__ __launch_bounds__(kSortThreads) void infer_topk_sort_kernel(
    Plan p, unsigned long long* keys) {
  // blockIdx.x enumerates the sorted levels, blockIdx.y the images
  int l = -1, seen = 0;
  for (int i = 0; i < p.L; ++i)
    if (p.pad[i] > 0) {
      if (seen == (int)blockIdx.x) l = i;
      ++seen;
    }
  if (l < 0) return;
  bitonic_desc(keys + (size_t)blockIdx.y * p.keys_per_img + p.keyoff[l], p.pad[l]);
}

// ---- 3. decode -----------------------------------------------------------------
__global__ __launch_bounds__(256) void infer_decode_kernel(
    Plan p, ld_maps_t cls, ld_maps_t reg, ld_maps_t ctr,
    const unsigned long long* keys, const float* img_hw, const float* scale_factors,
    float score_thr, float* boxes,
    float* scores, unsigned long long* cand, int* cand_count, unsigned* max_coord,
    float* factors) {
  const int n = blockIdx.y;
  const int slot = blockIdx.x * 256 + threadIdx.x;
  if (slot >= p.Ktot) return;
  int l = 0;
  for (int i = 1; i < p.L; ++i)
    if (slot >= p.koff[i]) l = i;
  const int r = slot - p.koff[l];
  int a = r;
  if (p.pad[l] > 0) {
    const unsigned long long key = keys[(size_t)n * p.keys_per_img + p.keyoff[l] + r];
    a = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
  }
  const int W = p.W[l], s = p.stride[l];
  const int cell = a / p.B, b = a - cell * p.B;
  const int y = cell / W, x = cell - y * W;
  // anchor centre (AnchorGenerator, center_offset 0: the cell's top-left corner)
  // FCOS points: + stride // 2 (fcos_gfl_head.py:548-558)
  const int half = p.points ? s / 2 : 0;
  const float cx = (float)(x * s + half), cy = (float)(y * s + half);
  // Integral * stride (gfl_head.py:32-44, :405)
  const float* rbase = reg.ptr[l] + (size_t)n * reg.stride_n[l] +
                       (size_t)b * 68 * reg.stride_c[l] + cell;
  float d[4];
#pragma unroll
  for (int side = 0; side < 4; ++side) {
    float sv[17], pv[17];
#pragma unroll
    for (int k = 0; k < 17; ++k)
      sv[k] = rbase[(size_t)(side * 17 + k) * reg.stride_c[l]];
    d[side] = softmax_expect<17>(sv, pv) * (float)s;
  }
  // distance2bbox + clamp to the image (transforms.py:135-154)
  const float Hi = img_hw[n * 2 + 0], Wi = img_hw[n * 2 + 1];
  float bx[4] = {cx - d[0], cy - d[1], cx + d[2], cy + d[3]};
  bx[0] = fminf(fmaxf(bx[0], 0.f), Wi);
  bx[1] = fminf(fmaxf(bx[1], 0.f), Hi);
  bx[2] = fminf(fmaxf(bx[2], 0.f), Wi);
  bx[3] = fminf(fmaxf(bx[3], 0.f), Hi);
  if (scale_factors) {
#pragma unroll
    for (int k = 0; k < 4; ++k) bx[k] = bx[k] / scale_factors[n * 4 + k];
  }
  float* bo = boxes + ((size_t)n * p.Ktot + slot) * 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) bo[k] = bx[k];
  // scores + candidates
  const float* cbase = cls.ptr[l] + (size_t)n * cls.stride_n[l] +
                       (size_t)b * p.C * cls.stride_c[l] + cell;
  float* so = scores + ((size_t)n * p.Ktot + slot) * p.C;
  bool any = false;
  // multiclass_nms(score_factors=centerness): the factor multiplies the score
  // AFTER the threshold test (bbox_nms.py:114-123)
  const float fac =
      p.has_ctr ? sigmoidf_(ctr.ptr[l][(size_t)n * ctr.stride_n[l] + cell]) : 1.0f;
  if (factors) factors[(size_t)n * p.Ktot + slot] = fac;
  for (int c = 0; c < p.C; ++c) {
    const float raw = cbase[(size_t)c * cls.stride_c[l]];
    const float sc = p.prob ? raw : sigmoidf_(raw);
    so[c] = sc;
    if (sc > score_thr) {
      any = true;
      const int pos = atomicAdd(&cand_count[n], 1);
      if (pos < p.cand_cap)
        cand[(size_t)n * p.cand_cap + pos] =
            make_key(p.has_ctr ? sc * fac : sc, (unsigned)(slot * p.C + c));
    }
  }
  if (any) {
    // boxes.max() over the candidate boxes (batched_nms); coordinates are >= 0
    const float m = fmaxf(fmaxf(bx[0], bx[1]), fmaxf(bx[2], bx[3]));
    atomicMax(&max_coord[n], __float_as_uint(m));
  }
}

// ---- 4. candidate sort ----------------------------------------------------------
__global__ __launch_bounds__(kSortThreads) void infer_cand_sort_kernel(
    Plan p, unsigned long long* cand, const int* cand_count) {
  const int n = blockIdx.x;
  const int M = min(cand_count[n], p.cand_cap);
  int len = 2;
  while (len < M) len <<= 1;
  unsigned long long* a = cand + (size_t)n * p.cand_cap;
  for (int i = M + threadIdx.x; i < len; i += blockDim.x) a[i] = 0ull;
  __syncthreads();
  bitonic_desc(a, len);
}

// ---- 5. greedy NMS, stops at max_per_img ----------------------------------------
// `cand` holds, per image (stride `cand_stride`), the best min(count, limit)
// candidates in descending order.  exhausted[n] = the list ran out before
// max_keep detections although more candidates exist (fast path only).
// DIOU = the reference's 'voting_cluster_diounms' branch (bbox_nms.py:141-176):
// overlap measure IoU - D^0.8 (D = squared centre distance / squared diagonal
// of the enclosing box, bbox_nms.py:35-67 with beta = 0.8) on boxes shifted by
// 4000 * label exactly as the reference shifts them (the fp32 rounding of the
// shifted coordinates is part of its decisions).  Cluster-NMS iterated to
// convergence keeps exactly the boxes of the greedy pass below; keep_rank[q] =
// position of kept box q in the sorted candidate list, for the voting pass.
__device__ __forceinline__ float diou_measure(const float (&a)[4], float area_a,
                                              const float (&b)[4], float area_b) {
  const float w = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]), 0.f);
  const float h = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]), 0.f);
  const float inter = w * h;
  const float uni = area_a + area_b - inter;
  const float dx = (b[2] + b[0]) / 2 - (a[2] + a[0]) / 2;
  const float dy = (b[3] + b[1]) / 2 - (a[3] + a[1]) / 2;
  const float cw = fmaxf(a[2], b[2]) - fminf(a[0], b[0]);
  const float ch = fmaxf(a[3], b[3]) - fminf(a[1], b[1]);
  const float D = (dx * dx + dy * dy) / (cw * cw + ch * ch + 1e-7f);
  return inter / uni - powf(D, 0.8f);
}

template <bool DIOU>
__global__ __launch_bounds__(kNmsThreads) void infer_nms_kernel(
    Plan p, const unsigned long long* cand, size_t cand_stride, int limit,
    const int* cand_count, const unsigned* max_coord, const float* boxes,
    float iou_thr, int max_keep, float* dets, long long* labels, int* counts,
    int* exhausted, int* keep_rank) {
  __shared__ float k_box[kMaxKeep][4];  // class-shifted coordinates
  __shared__ float k_area[kMaxKeep];
  __shared__ int k_label[kMaxKeep];
  __shared__ int s_alive[kNmsThreads];
  __shared__ int s_nkept;
  const int n = blockIdx.x, t = threadIdx.x;
  const int Mall = min(cand_count[n], p.cand_cap);
  const int M = min(Mall, limit);
  const unsigned long long* keys = cand + (size_t)n * cand_stride;
  const float shift_unit = DIOU ? 4000.0f : __uint_as_float(max_coord[n]) + 1.0f;
  if (t == 0) s_nkept = 0;
  __syncthreads();
  for (int base = 0; base < M; base += kNmsThreads) {
    if (s_nkept >= max_keep) break;  // uniform: read after a barrier
    const int i = base + t;
    bool alive = i < M;
    float sc = 0.f, ob[4] = {0.f, 0.f, 0.f, 0.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};
    float area = 0.f;
    int label = -1;
    if (alive) {
      const unsigned long long key = keys[i];
      sc = __uint_as_float((unsigned)(key >> 32));
      const unsigned pidx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
      const int slot = (int)(pidx / (unsigned)p.C);
      label = (int)(pidx - (unsigned)slot * (unsigned)p.C);
      const float* bo = boxes + ((size_t)n * p.Ktot + slot) * 4;
      const float off = (float)label * shift_unit;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        ob[k] = bo[k];
        sb[k] = ob[k] + off;
      }
      area = (sb[2] - sb[0]) * (sb[3] - sb[1]);
      // against everything kept so far
      const int nk = s_nkept;
      for (int q = 0; q < nk && alive; ++q) {
        if (k_label[q] != label) continue;
        float ovr;
        if (DIOU) {
          ovr = diou_measure(k_box[q], k_area[q], sb, area);
        } else {
          const float w =
              fmaxf(fminf(sb[2], k_box[q][2]) - fmaxf(sb[0], k_box[q][0]), 0.f);
          const float h =
              fmaxf(fminf(sb[3], k_box[q][3]) - fmaxf(sb[1], k_box[q][1]), 0.f);
          const float inter = w * h;
          ovr = inter / (k_area[q] + area - inter);
        }
        if (ovr > iou_thr) alive = false;
      }
    }
    s_alive[t] = alive ? 1 : 0;
    __syncthreads();
    // resolve the chunk in score order: the next alive candidate is kept and
    // suppresses the later ones of its class
    for (int u = 0; u < kNmsThreads; ++u) {
      if (!s_alive[u]) continue;           // uniform (LDS flag after a barrier)
      if (s_nkept >= max_keep) break;      // uniform
      const int q = s_nkept;
      if (t == u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) k_box[q][k] = sb[k];
        k_area[q] = area;
        k_label[q] = label;
        float* d = dets + ((size_t)n * max_keep + q) * 5;
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = ob[k];
        d[4] = sc;
        labels[(size_t)n * max_keep + q] = label;
        if (DIOU) keep_rank[(size_t)n * max_keep + q] = i;
      }
      __syncthreads();
      if (t == 0) s_nkept = q + 1;
      if (t > u && s_alive[t] && label == k_label[q]) {
        float ovr;
        if (DIOU) {
          ovr = diou_measure(k_box[q], k_area[q], sb, area);
        } else {
          const float w =
              fmaxf(fminf(sb[2], k_box[q][2]) - fmaxf(sb[0], k_box[q][0]), 0.f);
          const float h =
              fmaxf(fminf(sb[3], k_box[q][3]) - fmaxf(sb[1], k_box[q][1]), 0.f);
          const float inter = w * h;
          ovr = inter / (k_area[q] + area - inter);
        }
        if (ovr > iou_thr) s_alive[t] = 0;
      }
      __syncthreads();
    }
    __syncthreads();
  }
  __syncthreads();
  if (t == 0) {
    counts[n] = s_nkept;
    if (exhausted) exhausted[n] = (s_nkept < max_keep && Mall > M) ? 1 : 0;
  }
}

// ---- 6. score voting (bbox_nms.py:160-163) ----------------------------------
// For kept box i (rank r_i in the sorted list):
//   new_box = sum_j w_ij * box_j / sum_j w_ij over ALL candidates j,
//   w_ij = exp(-(1 - B_ij)^2 / 0.025) * score_j,
//   B_ij = DIoU(i, j) if j >= r_i (upper triangle incl. the diagonal), same
//          class-shifted geometry, and DIoU > 0.7; else 0  -- so every other
//          candidate still enters with the factor exp(-40), as in the
//          reference's dense matrix product.
// One workgroup per (kept box, image); fp32 sums like torch.mm / sum.
__global__ __launch_bounds__(256) void infer_vote_kernel(
    Plan p, const unsigned long long* cand, const int* cand_count, const float* boxes,
    int max_keep, const int* counts, const int* keep_rank, float* dets) {
  __shared__ float red[5][256];
  const int n = blockIdx.y, q = blockIdx.x, t = threadIdx.x;
  if (q >= counts[n]) return;
  const int M = min(cand_count[n], p.cand_cap);
  const unsigned long long* keys = cand + (size_t)n * p.cand_cap;
  auto fetch = [&](int j, float (&ob)[4], float (&sb)[4], float& sc, int& label) {
    const unsigned long long key = keys[j];
    sc = __uint_as_float((unsigned)(key >> 32));
    const unsigned pidx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
    const int slot = (int)(pidx / (unsigned)p.C);
    label = (int)(pidx - (unsigned)slot * (unsigned)p.C);
    const float* bo = boxes + ((size_t)n * p.Ktot + slot) * 4;
    const float off = (float)label * 4000.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      ob[k] = bo[k];
      sb[k] = ob[k] + off;
    }
  };
  const int ri = keep_rank[(size_t)n * max_keep + q];
  float iob[4], isb[4], isc;
  int ilabel;
  fetch(ri, iob, isb, isc, ilabel);
  const float iarea = (isb[2] - isb[0]) * (isb[3] - isb[1]);
  float sw = 0.f, sx[4] = {0.f, 0.f, 0.f, 0.f};
  for (int j = t; j < M; j += 256) {
    float ob[4], sb[4], sc;
    int label;
    fetch(j, ob, sb, sc, label);
    float B = 0.f;
    if (j >= ri && label == ilabel) {
      const float area = (sb[2] - sb[0]) * (sb[3] - sb[1]);
      const float d = diou_measure(isb, iarea, sb, area);
      if (d > 0.7f) B = d;
    }
    const float om = 1.0f - B;
    const float w = expf(-(om * om) / 0.025f) * sc;
    sw += w;
#pragma unroll
    for (int k = 0; k < 4; ++k) sx[k] += w * ob[k];
  }
  red[4][t] = sw;
#pragma unroll
  for (int k = 0; k < 4; ++k) red[k][t] = sx[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
#pragma unroll
      for (int k = 0; k < 5; ++k) red[k][t] += red[k][t + s];
    }
    __syncthreads();
  }
  if (t < 4) dets[((size_t)n * max_keep + q) * 5 + t] = red[t][0] / red[4][0];
}

int make_plan(const ld_geom_t* g, int num_classes, int nms_pre, Plan* p,
              int num_base = 1) {
  if (!g || g->num_levels < 1 || g->num_levels > LD_MAX_LEVELS || g->num_imgs < 1 ||
      num_classes < 1 || num_base < 1 || num_base > 64)
    return LD_EINVAL;
  p->B = num_base;
  p->N = g->num_imgs;
  p->L = g->num_levels;
  p->C = num_classes;
  p->prob = p->has_ctr = p->points = 0;
  int koff = 0, keyoff = 0;
  for (int l = 0; l < p->L; ++l) {
    p->H[l] = g->lv[l].H;
    p->W[l] = g->lv[l].W;
    p->stride[l] = g->lv[l].stride;
    p->A[l] = g->lv[l].H * g->lv[l].W * num_base;
    if (p->A[l] < 1) return LD_EINVAL;
    const bool sorted = nms_pre > 0 && p->A[l] > nms_pre;
    p->K[l] = sorted ? nms_pre : p->A[l];
    p->koff[l] = koff;
    koff += p->K[l];
    p->pad[l] = sorted ? next_pow2(p->A[l]) : 0;
    p->keyoff[l] = keyoff;
    keyoff += p->pad[l];
  }
  for (int l = p->L; l < LD_MAX_LEVELS; ++l) {
    p->H[l] = p->W[l] = p->stride[l] = p->A[l] = p->K[l] = p->pad[l] = 0;
    p->koff[l] = koff;
    p->keyoff[l] = keyoff;
  }
  p->Ktot = koff;
  p->keys_per_img = keyoff;
  const long long cap = (long long)koff * num_classes;
  if (cap > (1LL << 30)) return LD_EUNSUPPORTED;
  p->cand_cap = next_pow2((int)(cap < 2 ? 2 : cap));
  return 0;
}

struct Offsets {
  size_t keys, boxes, scores, cand, cand_top, count, maxc, flags, rank, total;
};

Offsets layout(const Plan& p) {
  Offsets o;
  size_t at = 0;
  auto take = [&](size_t bytes) {
    const size_t r = at;
    at += (bytes + 255) / 256 * 256;
    return r;
  };
  o.keys = take((size_t)p.N * (p.keys_per_img > 0 ? p.keys_per_img : 1) * 8);
  o.cand = take((size_t)p.N * p.cand_cap * 8);
  o.boxes = take((size_t)p.N * p.Ktot * 4 * sizeof(float));
  o.scores = take((size_t)p.N * p.Ktot * p.C * sizeof(float));
  o.cand_top = take((size_t)p.N * kSelN * 8);
  o.count = take((size_t)p.N * sizeof(int));
  o.maxc = take((size_t)p.N * sizeof(unsigned));
  o.flags = take((size_t)p.N * sizeof(int));
  o.rank = take((size_t)p.N * kMaxKeep * sizeof(int));
  o.total = at;
  return o;
}

}  // namespace

extern "C" size_t ld_get_bboxes_workspace_bytes(const ld_geom_t* g, int num_classes,
                                                int nms_pre) {
  Plan p;
  if (make_plan(g, num_classes, nms_pre, &p) != 0) return 0;
  return layout(p).total;
}

extern "C" size_t ld_get_bboxes_ex_workspace_bytes(const ld_geom_t* g, int num_classes,
                                                   int num_base, int nms_pre) {
  Plan p;
  if (make_plan(g, num_classes, nms_pre, &p, num_base) != 0) return 0;
  return layout(p).total;
}

static int get_bboxes_impl(const ld_geom_t* g, const ld_maps_t* cls,
                           const ld_maps_t* reg, int num_classes, int reg_max,
                           const float* img_hw, const float* scale_factors,
                           int nms_pre, float score_thr, float iou_thr,
                           int max_per_img, float* dets, int64_t* labels,
                           int32_t* counts, void* workspace, size_t workspace_bytes,
                           ld_stream_t stream_, bool voting, bool prob = false,
                           const ld_maps_t* ctr = nullptr, bool points = false,
                           int num_base = 1, float* pre_boxes = nullptr,
                           float* pre_scores = nullptr, float* pre_factors = nullptr) {
  Plan p;
  if (int e = make_plan(g, num_classes, nms_pre, &p, num_base)) return e;
  p.prob = prob ? 1 : 0;
  p.has_ctr = ctr ? 1 : 0;
  p.points = points ? 1 : 0;
  if (ctr && voting) return LD_EUNSUPPORTED;
  ld_maps_t ctr_maps = ctr ? *ctr : ld_maps_t{};
  const bool pre_only = pre_boxes != nullptr;
  if (!cls || !reg || !img_hw) return LD_EINVAL;
  if (pre_only ? !pre_scores : (!dets || !labels || !counts)) return LD_EINVAL;
  if (pre_only) max_per_img = 1;  // unused
  if (reg_max != 16) return LD_EUNSUPPORTED;  // 17-bin Integral only
  if (max_per_img < 1 || max_per_img > kMaxKeep) return LD_EUNSUPPORTED;
  // a negative threshold would suppress ACROSS classes in the reference (IoU 0
  // of class-shifted boxes > thr); only same-class pairs are compared here
  if (!(iou_thr >= 0.0f)) return LD_EINVAL;
  const Offsets o = layout(p);
  if (!workspace || workspace_bytes < o.total) return LD_ENOSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  char* ws = (char*)workspace;
  unsigned long long* keys = (unsigned long long*)(ws + o.keys);
  unsigned long long* cand = (unsigned long long*)(ws + o.cand);
  float* boxes = (float*)(ws + o.boxes);
  float* scores = (float*)(ws + o.scores);
  int* count = (int*)(ws + o.count);
  unsigned* maxc = (unsigned*)(ws + o.maxc);
  hipError_t err;
  if ((err = ldrec::memset_async(count, 0, (size_t)p.N * sizeof(int), stream)))
    return (int)err;
  if ((err = ldrec::memset_async(maxc, 0, (size_t)p.N * sizeof(unsigned), stream)))
    return (int)err;
  unsigned long long* cand_top = (unsigned long long*)(ws + o.cand_top);
  int* flags = (int*)(ws + o.flags);
  int nsorted = 0, maxpad = 0;
  for (int l = 0; l < p.L; ++l)
    if (p.pad[l] > 0) {
      ++nsorted;
      if (p.pad[l] > maxpad) maxpad = p.pad[l];
    }
  // LD_INFER_SORT=global: the plain global-memory bitonic sorts everywhere
  const char* env = getenv("LD_INFER_SORT");
  const bool force_global = voting || (env && env[0] == 'g');
  const bool fast_topk = !force_global && nms_pre <= kSelN;
  if (nsorted > 0) {
    LD_LAUNCH(infer_keys_kernel, dim3((maxpad + 255) / 256, p.L, p.N),
                       dim3(256), 0, stream, p, *cls, ctr_maps, keys);
    if (fast_topk)
      LD_LAUNCH(infer_topk_select_kernel, dim3(nsorted, p.N),
                         dim3(kSortThreads), 0, stream, p, keys);
    else
      LD_LAUNCH(infer_topk_sort_kernel, dim3(nsorted, p.N),
                         dim3(kSortThreads), 0, stream, p, keys);
  }
  LD_LAUNCH(infer_decode_kernel, dim3((p.Ktot + 255) / 256, p.N), dim3(256), 0,
                     stream, p, *cls, *reg, ctr_maps, keys, img_hw, scale_factors, score_thr,
                     boxes, scores, cand, count, maxc, pre_factors);
  if (pre_only) {
    // with_nms=False: the per-level selection + decode is the result
    if ((err = hipMemcpyAsync(pre_boxes, boxes, (size_t)p.N * p.Ktot * 4 * sizeof(float),
                              hipMemcpyDeviceToDevice, stream)))
      return (int)err;
    if ((err = hipMemcpyAsync(pre_scores, scores,
                              (size_t)p.N * p.Ktot * p.C * sizeof(float),
                              hipMemcpyDeviceToDevice, stream)))
      return (int)err;
    return (int)hipGetLastError();
  }
  bool need_global = force_global;
  if (!force_global) {
    // fast path: NMS over the kSelN best candidates of every image
    LD_LAUNCH(infer_cand_select_kernel, dim3(p.N), dim3(kSortThreads), 0,
                       stream, p, cand, count, cand_top);
    int limit = kSelN;  // LD_INFER_LIMIT: test hook to provoke the fallback
    if (const char* lim = getenv("LD_INFER_LIMIT")) {
      const int v = atoi(lim);
      if (v >= 1 && v <= kSelN) limit = v;
    }
    LD_LAUNCH(infer_nms_kernel<false>, dim3(p.N), dim3(kNmsThreads), 0, stream,
                       p, cand_top, (size_t)kSelN, limit, count, maxc, boxes, iou_thr,
                       max_per_img, dets, (long long*)labels, counts, flags,
                       (int*)nullptr);
    // rare: the best kSelN ran out before max_per_img detections -> redo the
    // NMS over the fully sorted list (the caller reads `counts` next anyway)
    int host_flags[64];
    if (p.N > 64) {
      need_global = true;
    } else {
      if ((err = hipMemcpyAsync(host_flags, flags, (size_t)p.N * sizeof(int),
                                hipMemcpyDeviceToHost, stream)))
        return (int)err;
      if ((err = hipStreamSynchronize(stream))) return (int)err;
      for (int n = 0; n < p.N; ++n)
        if (host_flags[n]) need_global = true;
    }
  }
  if (need_global) {
    LD_LAUNCH(infer_cand_sort_kernel, dim3(p.N), dim3(kSortThreads), 0, stream,
                       p, cand, count);
    if (voting) {
      int* rank = (int*)(ws + o.rank);
      LD_LAUNCH(infer_nms_kernel<true>, dim3(p.N), dim3(kNmsThreads), 0, stream,
                         p, cand, (size_t)p.cand_cap, p.cand_cap, count, maxc, boxes,
                         iou_thr, max_per_img, dets, (long long*)labels, counts,
                         (int*)nullptr, rank);
      LD_LAUNCH(infer_vote_kernel, dim3(max_per_img, p.N), dim3(256), 0, stream,
                         p, cand, count, boxes, max_per_img, counts, rank, dets);
    } else {
      LD_LAUNCH(infer_nms_kernel<false>, dim3(p.N), dim3(kNmsThreads), 0,
                         stream, p, cand, (size_t)p.cand_cap, p.cand_cap, count, maxc,
                         boxes, iou_thr, max_per_img, dets, (long long*)labels, counts,
                         (int*)nullptr, (int*)nullptr);
    }
  }
  return (int)hipGetLastError();
}

extern "C" int ld_get_bboxes(const ld_geom_t* g, const ld_maps_t* cls,
                             const ld_maps_t* reg, int num_classes, int reg_max,
                             const float* img_hw, const float* scale_factors,
                             int nms_pre, float score_thr, float iou_thr,
                             int max_per_img, float* dets, int64_t* labels,
                             int32_t* counts, void* workspace, size_t workspace_bytes,
                             ld_stream_t stream) {
  return get_bboxes_impl(g, cls, reg, num_classes, reg_max, img_hw, scale_factors,
                         nms_pre, score_thr, iou_thr, max_per_img, dets, labels,
                         counts, workspace, workspace_bytes, stream, false);
}

extern "C" int ld_get_bboxes_voting(const ld_geom_t* g, const ld_maps_t* cls,
                                    const ld_maps_t* reg, int num_classes,
                                    int reg_max, const float* img_hw,
                                    const float* scale_factors, int nms_pre,
                                    float score_thr, float iou_thr, int max_per_img,
                                    float* dets, int64_t* labels, int32_t* counts,
                                    void* workspace, size_t workspace_bytes,
                                    ld_stream_t stream) {
  return get_bboxes_impl(g, cls, reg, num_classes, reg_max, img_hw, scale_factors,
                         nms_pre, score_thr, iou_thr, max_per_img, dets, labels,
                         counts, workspace, workspace_bytes, stream, true);
}

extern "C" int ld_get_bboxes_ex(const ld_geom_t* g, const ld_maps_t* cls,
                                const ld_maps_t* reg, const ld_maps_t* ctr,
                                int num_classes, int num_base, int reg_max,
                                const float* img_hw, const float* scale_factors,
                                int nms_pre, float score_thr, float iou_thr,
                                int max_per_img, int flags, float* dets,
                                int64_t* labels, int32_t* counts, void* workspace,
                                size_t workspace_bytes, ld_stream_t stream) {
  if (flags & ~(LD_INFER_VOTING | LD_INFER_PROB | LD_INFER_POINTS)) return LD_EINVAL;
  return get_bboxes_impl(g, cls, reg, num_classes, reg_max, img_hw, scale_factors,
                         nms_pre, score_thr, iou_thr, max_per_img, dets, labels,
                         counts, workspace, workspace_bytes, stream,
                         (flags & LD_INFER_VOTING) != 0, (flags & LD_INFER_PROB) != 0,
                         ctr, (flags & LD_INFER_POINTS) != 0, num_base);
}

extern "C" int ld_get_bboxes_num_selected(const ld_geom_t* g, int num_base, int nms_pre) {
  Plan p;
  if (make_plan(g, 1, nms_pre, &p, num_base) != 0) return -1;
  return p.Ktot;
}

extern "C" int ld_get_bboxes_pre_nms(const ld_geom_t* g, const ld_maps_t* cls,
                                     const ld_maps_t* reg, const ld_maps_t* ctr,
                                     int num_classes, int num_base, int reg_max,
                                     const float* img_hw, const float* scale_factors,
                                     int nms_pre, int flags, float* boxes, float* scores,
                                     float* factors, void* workspace,
                                     size_t workspace_bytes, ld_stream_t stream) {
  if (flags & ~(LD_INFER_PROB | LD_INFER_POINTS)) return LD_EINVAL;
  if (!boxes || !scores || (factors && !ctr)) return LD_EINVAL;
  return get_bboxes_impl(g, cls, reg, num_classes, reg_max, img_hw, scale_factors,
                         nms_pre, 0.0f, 0.5f, 1, nullptr, nullptr, nullptr, workspace,
                         workspace_bytes, stream, false, (flags & LD_INFER_PROB) != 0, ctr,
                         (flags & LD_INFER_POINTS) != 0, num_base, boxes, scores, factors);
}
