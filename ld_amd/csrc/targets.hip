// ATSS target assignment + VLR region + IM region for a whole batch, on gfx950.
//
// Replaces (reference, mmdet 2.10 fork):
//   ATSSAssigner.assign            mmdet/core/bbox/assigners/atss_assigner.py:33-181
//   ATSSAssigner.get_vlr_region    ...atss_assigner.py:183-298
//   LDHead.get_im_region           mmdet/models/dense_heads/ld_head.py:580-611
//   LDHead._get_target_single      ...ld_head.py:449-577 (+ unmap, misc.py:32-42)
//   PseudoSampler.sample           mmdet/core/bbox/samplers/pseudo_sampler.py:24-41
//
// Design (MI355X-first, not a translation of the ~40 ATen launches per image):
//   kernel A  one 256-thread workgroup per (image, GT): streams every valid
//             anchor of every level once, keeps a register-resident sorted
//             top-k per thread, merges them with k rounds of wave-shuffle
//             arg-min, derives the mean+std IoU threshold and publishes the
//             positive candidates with a 64-bit atomicMax (IoU, lowest GT
//             index) per anchor.  Anchors are never materialised: anchor a of
//             level l is the square of side 8*stride centred on (x*s, y*s).
//   kernel B  one thread per (image, anchor): loops over the image's GTs held
//             in LDS, evaluates the IoU / IoF-"diou" tile ONCE (the reference
//             evaluates it five times) and writes the dense label / weight /
//             box-target / VLR / IM maps in level-major layout, i.e. already in
//             the form `images_to_levels` would produce.
// All integer results (labels, positive indices, masks) are bit-exact w.r.t.
// the reference CPU path on tie-free inputs; fp32 op order follows the
// reference (no FMA contraction in this file).
#include <hip/hip_runtime.h>

#include "ld_launch.h"
#include <float.h>
#include <stdlib.h>
#include <limits.h>

#include "../../include/ld_hip.h"
#include "ld_math.h"

namespace {

using ld::Box;

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;
constexpr int kMaxCand = LD_MAX_LEVELS * 16;

struct DistIdx {
  float d;
  int i;
};

__device__ __forceinline__ bool lex_less(float d0, int i0, float d1, int i1) {
  return (d0 < d1) || (d0 == d1 && i0 < i1);
}

__device__ __forceinline__ DistIdx wave_argmin(DistIdx v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    float od = __shfl_xor(v.d, off);
    int oi = __shfl_xor(v.i, off);
    if (lex_less(od, oi, v.d, v.i)) {
      v.d = od;
      v.i = oi;
    }
  }
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

// explicit anchors (reference-style API: any (A,4) box list split into levels by
// count) or implicit ones generated from the pyramid geometry
__device__ __forceinline__ Box get_anchor(const float* __restrict__ anchors, int a,
                                          int x, int y, int stride, float half) {
  if (anchors) {
    const float4 q = reinterpret_cast<const float4*>(anchors)[a];
    return Box{q.x, q.y, q.z, q.w};
  }
  return ld::anchor_box(x, y, stride, half);
}

__device__ __forceinline__ int level_of(const ld_geom_t& g, int a) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < LD_MAX_LEVELS; ++i)
    if (i < g.num_levels && a >= g.lv[i].offset) l = i;
  return l;
}

// ---------------------------------------------------------------- kernel A --
// workspace: keys (N*A u64), thr (N*max_gt f32), colmax (N*max_gt f32)
template <int KMAX>
__global__ __launch_bounds__(kBlock) void atss_select_kernel(
    ld_geom_t geom, int topk, const float* __restrict__ anchors,
    const float* __restrict__ gt_bboxes, const int32_t* __restrict__ num_gt,
    int max_gt, const int32_t* __restrict__ valid_hw, int wmul,
    unsigned long long* __restrict__ keys, float* __restrict__ thr_out,
    float* __restrict__ colmax_out) {
  // wmul: the valid width counts positions; a multi-anchor level is laid out
  // as H x (W * wmul) with the base anchor fastest (ld_retina_targets)
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  if (g >= num_gt[n]) return;
  const float* gp = gt_bboxes + ((size_t)n * max_gt + g) * 4;
  const Box gt{gp[0], gp[1], gp[2], gp[3]};
  const float gcx = (gt.x1 + gt.x2) / 2.0f, gcy = (gt.y1 + gt.y2) / 2.0f;

  __shared__ DistIdx s_wave[kWaves];
  __shared__ float s_wmax[kWaves];
  __shared__ int s_cand[kMaxCand];
  __shared__ float s_ciou[kMaxCand];
  __shared__ float s_thr;
  int ncand = 0;  // uniform across the block
  float my_colmax = 0.0f;

  for (int l = 0; l < geom.num_levels; ++l) {
    const ld_level_t lv = geom.lv[l];
    const int vh = valid_hw[((size_t)n * geom.num_levels + l) * 2 + 0];
    const int vw = valid_hw[((size_t)n * geom.num_levels + l) * 2 + 1] * wmul;
    const int nvalid = vh * vw;
    const int k = min(topk, nvalid);
    const float half = 0.5f * (float)(geom.anchor_scale * lv.stride);
    float ld_[KMAX];
    int li_[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      ld_[j] = FLT_MAX;
      li_[j] = INT_MAX;
    }
    for (int v = tid; v < nvalid; v += kBlock) {
      const int y = v / vw, x = v - y * vw;
      const int a = lv.offset + y * lv.W + x;
      const Box ab = get_anchor(anchors, a, x, y, lv.stride, half);
      const float acx = (ab.x1 + ab.x2) / 2.0f, acy = (ab.y1 + ab.y2) / 2.0f;
      float d = ld::centre_dist(acx, acy, gcx, gcy);
      my_colmax = fmaxf(my_colmax, ld::iou_pair(ab, gt));
      int ai = a;
      if (lex_less(d, ai, ld_[KMAX - 1], li_[KMAX - 1])) {
#pragma unroll
        for (int j = 0; j < KMAX; ++j) {
          if (lex_less(d, ai, ld_[j], li_[j])) {
            float td = ld_[j];
            int ti = li_[j];
            ld_[j] = d;
            li_[j] = ai;
            d = td;
            ai = ti;
          }
        }
      }
    }
    // merge: k rounds of block-wide lexicographic arg-min over the list heads
    for (int r = 0; r < k; ++r) {
      DistIdx w = wave_argmin(DistIdx{ld_[0], li_[0]});
      if ((tid & 63) == 0) s_wave[tid >> 6] = w;
      __syncthreads();
      DistIdx best = s_wave[0];
#pragma unroll
      for (int q = 1; q < kWaves; ++q)
        if (lex_less(s_wave[q].d, s_wave[q].i, best.d, best.i)) best = s_wave[q];
      if (li_[0] == best.i && best.i != INT_MAX) {  // the unique owner pops
#pragma unroll
        for (int j = 0; j < KMAX - 1; ++j) {
          ld_[j] = ld_[j + 1];
          li_[j] = li_[j + 1];
        }
        ld_[KMAX - 1] = FLT_MAX;
        li_[KMAX - 1] = INT_MAX;
      }
      if (tid == 0) s_cand[ncand + r] = best.i;
      __syncthreads();
    }
    ncand += k;
  }
  // column max IoU over all valid anchors (for the IM region)
  {
    float m = wave_max(my_colmax);
    if ((tid & 63) == 0) s_wmax[tid >> 6] = m;
  }
  __syncthreads();
  // IoU of the candidates
  Box cab{0, 0, 0, 0};
  float ciou = 0.0f;
  if (tid < ncand) {
    const int a = s_cand[tid];
    const int l = level_of(geom, a);
    const ld_level_t lv = geom.lv[l];
    const int r = a - lv.offset;
    const int y = r / lv.W, x = r - y * lv.W;
    cab = get_anchor(anchors, a, x, y, lv.stride,
                     0.5f * (float)(geom.anchor_scale * lv.stride));
    ciou = ld::iou_pair(cab, gt);
    s_ciou[tid] = ciou;
  }
  __syncthreads();
  if (tid == 0) {
    // mean + unbiased std (atss_assigner.py:126-131); float64 accumulate,
    // one rounding each -- matches torch's fp32 mean/std on every fixture
    double sum = 0.0;
    for (int i = 0; i < ncand; ++i) sum += (double)s_ciou[i];
    const double mean = sum / (double)ncand;
    double ss = 0.0;
    for (int i = 0; i < ncand; ++i) {
      const double dd = (double)s_ciou[i] - mean;
      ss += dd * dd;
    }
    const double sd = sqrt(ss / (double)(ncand - 1));  // NaN when ncand == 1
    const float thr = (float)mean + (float)sd;
    s_thr = thr;
    thr_out[(size_t)n * max_gt + g] = thr;
    float cm = s_wmax[0];
#pragma unroll
    for (int q = 1; q < kWaves; ++q) cm = fmaxf(cm, s_wmax[q]);
    colmax_out[(size_t)n * max_gt + g] = cm;
  }
  __syncthreads();
  if (tid < ncand) {
    const float thr = s_thr;
    const float acx = (cab.x1 + cab.x2) / 2.0f, acy = (cab.y1 + cab.y2) / 2.0f;
    const float l_ = acx - gt.x1, t_ = acy - gt.y1;
    const float r_ = gt.x2 - acx, b_ = gt.y2 - acy;
    const float mn = fminf(fminf(l_, t_), fminf(r_, b_));
    if (ciou >= thr && mn > 0.01f) {
      // highest IoU wins, ties -> lowest GT index (torch.max(dim=1) on CPU)
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(ciou) << 32) |
          (unsigned long long)(0xFFFFFFFFu - (unsigned)g);
      atomicMax(keys + (size_t)n * geom.num_anchors + s_cand[tid], key);
    }
  }
}

// ------------------------------------------------- kernel A, windowed (round 6) --
// The scan above streams all ~22 400 anchors through ONE workgroup per (image, GT):
// 131 us per C2 step for 14 workgroups on 256 CUs (profiles/r05_rocprof_kernel_
// stats_fp32.csv), a latency chain, not a bandwidth problem.  With IMPLICIT anchors
// (one square per cell, centred on (x s, y s)) the k <= 9 nearest centres of a level
// lie in a small window around the cell nearest to the GT centre, so each level
// needs (2 R + 1)^2 = 289 candidates, not up to 16 800:
//   * a valid region of at least 3 x 3 cells holds 9 cells within 3.6 cell sides
//     of any point inside it, every cell outside the window is > 7 away;
//   * a strip (1 or 2 valid rows / columns) holds its 9 nearest within +-8 cells
//     along the strip -- hence R = 8 -- and a level with fewer than 9 valid cells
//     (k = nvalid) is at most 8 x 8: the window covers it whole.
// The column maximum of the IoU (IM region) is attained within one cell of the
// nearest centre (the overlap of two axis-aligned boxes of fixed sizes is a
// product of two trapezoids of the centre offset; where it is flat the computed
// values are identical bit for bit), i.e. inside the same window.
// One WAVE per level: every lane evaluates <= 5 candidates with the same
// arithmetic as the scan, keeps them sorted, and k rounds of wave-shuffle arg-min
// (distance, then anchor index: the scan's order) pop the winners -- no block
// barrier until the candidates of all levels are in LDS, from where the second
// half (IoU threshold, 64-bit atomicMax publish) is the scan kernel's, verbatim.
// Results are bit-identical to the scan (tests/test_gpu_atss.py runs both).
constexpr int kWinR = 8;
constexpr int kWinK = 5;  // ceil(17 * 17 / 64)

__global__ __launch_bounds__(kBlock) void atss_select_window_kernel(
    ld_geom_t geom, int topk, const float* __restrict__ gt_bboxes,
    const int32_t* __restrict__ num_gt, int max_gt, const int32_t* __restrict__ valid_hw,
    unsigned long long* __restrict__ keys, float* __restrict__ thr_out,
    float* __restrict__ colmax_out) {
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  if (g >= num_gt[n]) return;
  const float* gp = gt_bboxes + ((size_t)n * max_gt + g) * 4;
  const Box gt{gp[0], gp[1], gp[2], gp[3]};
  const float gcx = (gt.x1 + gt.x2) / 2.0f, gcy = (gt.y1 + gt.y2) / 2.0f;
  const int lane = tid & 63, wave = tid >> 6;

  __shared__ float s_wmax[kWaves];
  __shared__ int s_cand[kMaxCand];
  __shared__ float s_ciou[kMaxCand];
  __shared__ float s_thr;
  float my_colmax = 0.0f;
  int ncand = 0;  // total over the levels (uniform)
  for (int l = 0; l < geom.num_levels; ++l) {
    const ld_level_t lv = geom.lv[l];
    const int vh = valid_hw[((size_t)n * geom.num_levels + l) * 2 + 0];
    const int vw = valid_hw[((size_t)n * geom.num_levels + l) * 2 + 1];
    const int nvalid = vh * vw;
    const int k = min(topk, nvalid);
    const int base = ncand;  // this level's slots in s_cand: the scan's order
    ncand += k;
    if ((l % kWaves) != wave || k == 0) continue;
    const float half = 0.5f * (float)(geom.anchor_scale * lv.stride);
    const float inv = 1.0f / (float)lv.stride;
    const int xc = min(max((int)floorf(gcx * inv + 0.5f), 0), vw - 1);
    const int yc = min(max((int)floorf(gcy * inv + 0.5f), 0), vh - 1);
    const int x0 = max(xc - kWinR, 0), x1 = min(xc + kWinR, vw - 1);
    const int y0 = max(yc - kWinR, 0), y1 = min(yc + kWinR, vh - 1);
    const int ww = x1 - x0 + 1, wn = ww * (y1 - y0 + 1);
    float ld_[kWinK];
    int li_[kWinK];
#pragma unroll
    for (int j = 0; j < kWinK; ++j) {
      ld_[j] = FLT_MAX;
      li_[j] = INT_MAX;
    }
    for (int c = lane; c < wn; c += 64) {
      const int wy = c / ww, wx = c - wy * ww;
      const int x = x0 + wx, y = y0 + wy;
      const int a = lv.offset + y * lv.W + x;
      const Box ab = ld::anchor_box(x, y, lv.stride, half);
      const float acx = (ab.x1 + ab.x2) / 2.0f, acy = (ab.y1 + ab.y2) / 2.0f;
      float d = ld::centre_dist(acx, acy, gcx, gcy);
      my_colmax = fmaxf(my_colmax, ld::iou_pair(ab, gt));
      int ai = a;
#pragma unroll
      for (int j = 0; j < kWinK; ++j) {
        if (lex_less(d, ai, ld_[j], li_[j])) {
          const float td = ld_[j];
          const int ti = li_[j];
          ld_[j] = d;
          li_[j] = ai;
          d = td;
          ai = ti;
        }
      }
    }
    for (int r = 0; r < k; ++r) {
      const DistIdx best = wave_argmin(DistIdx{ld_[0], li_[0]});
      if (li_[0] == best.i && best.i != INT_MAX) {  // the unique owner pops
#pragma unroll
        for (int j = 0; j < kWinK - 1; ++j) {
          ld_[j] = ld_[j + 1];
          li_[j] = li_[j + 1];
        }
        ld_[kWinK - 1] = FLT_MAX;
        li_[kWinK - 1] = INT_MAX;
      }
      if (lane == 0) s_cand[base + r] = best.i;
    }
  }
  {
    const float m = wave_max(my_colmax);
    if (lane == 0) s_wmax[wave] = m;
  }
  __syncthreads();
  // ---- from here on: the scan kernel's second half
  Box cab{0, 0, 0, 0};
  float ciou = 0.0f;
  if (tid < ncand) {
    const int a = s_cand[tid];
    const int l = level_of(geom, a);
    const ld_level_t lv = geom.lv[l];
    const int r = a - lv.offset;
    const int y = r / lv.W, x = r - y * lv.W;
    cab = ld::anchor_box(x, y, lv.stride, 0.5f * (float)(geom.anchor_scale * lv.stride));
    ciou = ld::iou_pair(cab, gt);
    s_ciou[tid] = ciou;
  }
  __syncthreads();
  if (tid == 0) {
    double sum = 0.0;
    for (int i = 0; i < ncand; ++i) sum += (double)s_ciou[i];
    const double mean = sum / (double)ncand;
    double ss = 0.0;
    for (int i = 0; i < ncand; ++i) {
      const double dd = (double)s_ciou[i] - mean;
      ss += dd * dd;
    }
    const double sd = sqrt(ss / (double)(ncand - 1));  // NaN when ncand == 1
    const float thr = (float)mean + (float)sd;
    s_thr = thr;
    thr_out[(size_t)n * max_gt + g] = thr;
    float cm = s_wmax[0];
#pragma unroll
    for (int q = 1; q < kWaves; ++q) cm = fmaxf(cm, s_wmax[q]);
    colmax_out[(size_t)n * max_gt + g] = cm;
  }
  __syncthreads();
  if (tid < ncand) {
    const float thr = s_thr;
    const float acx = (cab.x1 + cab.x2) / 2.0f, acy = (cab.y1 + cab.y2) / 2.0f;
    const float l_ = acx - gt.x1, t_ = acy - gt.y1;
    const float r_ = gt.x2 - acx, b_ = gt.y2 - acy;
    const float mn = fminf(fminf(l_, t_), fminf(r_, b_));
    if (ciou >= thr && mn > 0.01f) {
      const unsigned long long key =
          ((unsigned long long)__float_as_uint(ciou) << 32) |
          (unsigned long long)(0xFFFFFFFFu - (unsigned)g);
      atomicMax(keys + (size_t)n * geom.num_anchors + s_cand[tid], key);
    }
  }
}

// ---------------------------------------------------------------- kernel B --
constexpr int kGtChunk = 128;

__global__ __launch_bounds__(kBlock) void atss_dense_kernel(
    ld_geom_t geom, int num_classes, int im_center_inside,
    const float* __restrict__ anchors,
    const float* __restrict__ gt_bboxes, const int64_t* __restrict__ gt_labels,
    const int32_t* __restrict__ num_gt, int max_gt,
    const int32_t* __restrict__ valid_hw,
    const unsigned long long* __restrict__ keys, const float* __restrict__ thr,
    const float* __restrict__ colmax, int64_t* __restrict__ labels,
    float* __restrict__ label_weights, float* __restrict__ bbox_targets,
    float* __restrict__ vlr, float* __restrict__ im, int32_t* __restrict__ counts,
    int64_t* __restrict__ gt_inds_out, float* __restrict__ max_overlaps_out) {
  const int n = blockIdx.y, tid = threadIdx.x;
  const int a = blockIdx.x * kBlock + tid;
  const int A = geom.num_anchors, L = geom.num_levels, N = geom.num_imgs;
  const int G = num_gt[n];
  __shared__ float4 s_gt[kGtChunk];
  __shared__ float s_thr[kGtChunk];
  __shared__ float s_cm[kGtChunk];

  bool valid = false;
  Box ab{0, 0, 0, 0};
  int l = 0;
  if (a < A) {
    l = level_of(geom, a);
    const ld_level_t lv = geom.lv[l];
    const int r = a - lv.offset;
    const int y = r / lv.W, x = r - y * lv.W;
    const int vh = valid_hw[((size_t)n * L + l) * 2 + 0];
    const int vw = valid_hw[((size_t)n * L + l) * 2 + 1];
    valid = (y < vh) && (x < vw);
    ab = get_anchor(anchors, a, x, y, lv.stride,
                    0.5f * (float)(geom.anchor_scale * lv.stride));
  }
  float vmax = ld::kNegInf;
  bool is_im = false;
  for (int g0 = 0; g0 < G; g0 += kGtChunk) {
    const int gc = min(kGtChunk, G - g0);
    __syncthreads();
    if (tid < gc) {
      const float* gp = gt_bboxes + ((size_t)n * max_gt + g0 + tid) * 4;
      s_gt[tid] = make_float4(gp[0], gp[1], gp[2], gp[3]);
      s_thr[tid] = thr[(size_t)n * max_gt + g0 + tid];
      s_cm[tid] = colmax[(size_t)n * max_gt + g0 + tid];
    }
    __syncthreads();
    if (valid) {
      for (int j = 0; j < gc; ++j) {
        const float4 q = s_gt[j];
        const Box gt{q.x, q.y, q.z, q.w};
        const float iou = ld::iou_pair(ab, gt);
        const float di = ld::diou_pair(ab, gt);
        const float t = s_thr[j];
        // atss_assigner.py:271-272
        if ((di < t) && (di >= 0.25f * t)) vmax = fmaxf(vmax, iou);
        if (im_center_inside) {
          // 'fitnet' / 'decouple' / 'gibox' region: anchor centre strictly
          // inside a GT box (ld_head.py:598-606)
          const float cx = (ab.x2 + ab.x1) / 2, cy = (ab.y2 + ab.y1) / 2;
          if (cx > gt.x1 && cx < gt.x2 && cy > gt.y1 && cy < gt.y2) is_im = true;
        } else if (iou > 0.5f * s_cm[j]) {  // 'finegrained', ld_head.py:594-596
          is_im = true;
        }
      }
    }
  }
  if (a >= A) return;
  const size_t o = (size_t)n * A + a;
  int64_t lab = num_classes, gind = 0;
  float lw = 0.0f, bt[4] = {0, 0, 0, 0}, vl = 0.0f, imv = 0.0f;
  float mov = ld::kNegInf;
  if (valid) {
    lw = 1.0f;  // pos_weight <= 0 -> 1.0 for positives, negatives 1.0
    const unsigned long long key = keys[o];
    if (key != 0ull) {
      const int g = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
      gind = g + 1;
      mov = __uint_as_float((unsigned)(key >> 32));
      const float* gp = gt_bboxes + ((size_t)n * max_gt + g) * 4;
      bt[0] = gp[0];
      bt[1] = gp[1];
      bt[2] = gp[2];
      bt[3] = gp[3];
      lab = gt_labels[(size_t)n * max_gt + g];
      atomicAdd(counts + n, 1);
      atomicAdd(counts + N + l, 1);
    }
    vl = (vmax != ld::kNegInf) ? vmax : 0.0f;
    imv = is_im ? 1.0f : 0.0f;
    if (is_im) atomicAdd(counts + N + L + l, 1);
  }
  labels[o] = lab;
  label_weights[o] = lw;
  reinterpret_cast<float4*>(bbox_targets)[o] =
      make_float4(bt[0], bt[1], bt[2], bt[3]);
  vlr[o] = vl;
  im[o] = imv;
  if (gt_inds_out) gt_inds_out[o] = gind;
  if (max_overlaps_out) max_overlaps_out[o] = mov;
}

__global__ void atss_counts_finalize(int N, int L, int32_t* counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int s = 0;
    for (int i = 0; i < N; ++i) s += max(counts[i], 1);  // ld_head.py:428
    counts[N + 2 * L] = s;
  }
}

// ------------------------------------------- MaxIoU targets, B anchors/cell --
// LDRetinaHead._get_targets_single (ld_retina.py:256-362): MaxIoUAssigner
// (max_iou_assigner.py:131-212, match_low_quality + gt_max_assign_all, no
// ignore boxes) + PseudoSampler + the head's own get_vlr_region
// (ld_retina.py:478-603, the ATSS VLR formula over all B anchors of a cell).
// Thread = one anchor in the reference's order (level, position, base anchor);
// outputs go to the PSEUDO-IMAGE layout (n * B + b, level offset + position):
// an (N, B * C, H, W) head map is the (N * B, C, H, W) map of N * B images
// with one anchor per cell, which is what the loss-block kernels sweep.
struct MaxIouCfg {
  float pos_thr, neg_thr, min_pos;
  int num_base;
};

__global__ __launch_bounds__(kBlock) void maxiou_dense_kernel(
    ld_geom_t geom /* flattened: W * B wide levels */, int num_classes, MaxIouCfg cfg,
    const float* __restrict__ anchors, const float* __restrict__ gt_bboxes,
    const int64_t* __restrict__ gt_labels, const int32_t* __restrict__ num_gt, int max_gt,
    const int32_t* __restrict__ valid_hw, const float* __restrict__ thr,
    const float* __restrict__ colmax, int64_t* __restrict__ labels,
    float* __restrict__ label_weights, float* __restrict__ bbox_targets,
    float* __restrict__ vlr, float* __restrict__ im, int32_t* __restrict__ counts,
    int64_t* __restrict__ gt_inds_out) {
  const int n = blockIdx.y, tid = threadIdx.x;
  const int a = blockIdx.x * kBlock + tid;
  const int AB = geom.num_anchors, L = geom.num_levels, B = cfg.num_base;
  const int NB = geom.num_imgs * B, A = AB / B;
  const int G = num_gt[n];
  __shared__ float4 s_gt[kGtChunk];
  __shared__ float s_thr[kGtChunk];
  __shared__ float s_cm[kGtChunk];
  bool valid = false;
  Box ab{0, 0, 0, 0};
  int l = 0, p = 0, b = 0;
  if (a < AB) {
    l = level_of(geom, a);
    const ld_level_t lv = geom.lv[l];
    const int r = a - lv.offset;
    p = r / B;
    b = r - p * B;
    const int W = lv.W / B;
    const int y = p / W, x = p - y * W;
    const int vh = valid_hw[((size_t)n * L + l) * 2 + 0];
    const int vw = valid_hw[((size_t)n * L + l) * 2 + 1];
    valid = (y < vh) && (x < vw);
    const float4 q = reinterpret_cast<const float4*>(anchors)[a];
    ab = Box{q.x, q.y, q.z, q.w};
  }
  float vmax = ld::kNegInf, mx = ld::kNegInf;
  int arg = 0, lowq = -1;
  for (int g0 = 0; g0 < G; g0 += kGtChunk) {
    const int gc = min(kGtChunk, G - g0);
    __syncthreads();
    if (tid < gc) {
      const float* gp = gt_bboxes + ((size_t)n * max_gt + g0 + tid) * 4;
      s_gt[tid] = make_float4(gp[0], gp[1], gp[2], gp[3]);
      s_thr[tid] = thr[(size_t)n * max_gt + g0 + tid];
      s_cm[tid] = colmax[(size_t)n * max_gt + g0 + tid];
    }
    __syncthreads();
    if (valid) {
      for (int j = 0; j < gc; ++j) {
        const float4 q = s_gt[j];
        const Box gt{q.x, q.y, q.z, q.w};
        const float iou = ld::iou_pair(ab, gt);
        const float di = ld::diou_pair(ab, gt);
        const float t = s_thr[j];
        if ((di < t) && (di >= 0.25f * t)) vmax = fmaxf(vmax, iou);
        if (iou > mx) {  // first maximum: torch.max(dim=0) on CPU
          mx = iou;
          arg = g0 + j;
        }
        // every anchor that attains a gt's best IoU is its match; the loop
        // over gts lets a later one override (max_iou_assigner.py:196-204)
        if (s_cm[j] >= cfg.min_pos && iou == s_cm[j]) lowq = g0 + j;
      }
    }
  }
  if (a >= AB) return;
  const int lvoff = geom.lv[l].offset / B;
  const size_t o = ((size_t)n * B + b) * A + lvoff + p;
  int64_t lab = num_classes, gind = -1;
  float lw = 0.0f, bt[4] = {0, 0, 0, 0}, vl = 0.0f;
  if (valid) {
    if (G == 0) {
      gind = 0;
    } else if (lowq >= 0) {
      gind = lowq + 1;
    } else if (mx >= cfg.pos_thr) {
      gind = arg + 1;
    } else if (mx >= 0.0f && mx < cfg.neg_thr) {
      gind = 0;
    }
    if (gind > 0) {
      const int g = (int)gind - 1;
      const float* gp = gt_bboxes + ((size_t)n * max_gt + g) * 4;
      bt[0] = gp[0];
      bt[1] = gp[1];
      bt[2] = gp[2];
      bt[3] = gp[3];
      lab = gt_labels[(size_t)n * max_gt + g];
      atomicAdd(counts + n * B + b, 1);
      atomicAdd(counts + NB + l, 1);
    }
    lw = gind >= 0 ? 1.0f : 0.0f;  // the band between the thresholds is ignored
    vl = (vmax != ld::kNegInf) ? vmax : 0.0f;
  }
  labels[o] = lab;
  label_weights[o] = lw;
  reinterpret_cast<float4*>(bbox_targets)[o] = make_float4(bt[0], bt[1], bt[2], bt[3]);
  vlr[o] = vl;
  im[o] = 0.0f;
  if (gt_inds_out) gt_inds_out[o] = gind;
}

__global__ void maxiou_counts_finalize(int N, int B, int L, int32_t* counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int s = 0;
    for (int i = 0; i < N; ++i) {  // per REAL image: sum over its B pseudo-images
      int p = 0;
      for (int b = 0; b < B; ++b) p += counts[i * B + b];
      s += max(p, 1);  // ld_retina.py:449
    }
    counts[N * B + 2 * L] = s;
  }
}

// ------------------------------------------------------ FCOS point targets ---
// LDFCOSHead._get_target_single (ld_fcos_head.py:261-353) for every point of
// every image: the object with the smallest area among those whose (centre-
// sampled) box contains the point and whose largest side distance falls in the
// level's regress range; ties keep the lower gt index (torch.min's first
// minimum).  Points inside some gt box but assigned to none are the head's
// "remain" set (label num_classes + 1 in the reference): reported as background
// with vlr = 1.  bbox_targets = (left, top, right, bottom) distances in pixels.
struct FcosCfg {
  float lo[LD_MAX_LEVELS], hi[LD_MAX_LEVELS];
  float radius;
  int center_sampling;
};

__global__ __launch_bounds__(kBlock) void fcos_dense_kernel(
    ld_geom_t geom, int num_classes, FcosCfg cfg, const float* __restrict__ gt_bboxes,
    const int64_t* __restrict__ gt_labels, const int32_t* __restrict__ num_gt,
    int max_gt, int64_t* __restrict__ labels, float* __restrict__ label_weights,
    float* __restrict__ bbox_targets, float* __restrict__ vlr, float* __restrict__ im,
    int32_t* __restrict__ counts) {
  const int n = blockIdx.y;
  const int a = blockIdx.x * kBlock + threadIdx.x;
  const int A = geom.num_anchors, N = geom.num_imgs;
  if (a >= A) return;
  const int l = level_of(geom, a);
  const ld_level_t lv = geom.lv[l];
  const int r = a - lv.offset;
  const int y = r / lv.W, x = r - y * lv.W;
  // fcos_gfl_head.py:548-558: point = index * stride + stride // 2
  const float px = (float)(x * lv.stride + lv.stride / 2);
  const float py = (float)(y * lv.stride + lv.stride / 2);
  const int G = num_gt[n];
  const float INF = 1e8f;
  float best = INF;
  int arg = 0;
  bool in_some = false;
  const float sr = (float)lv.stride * cfg.radius;
  for (int g = 0; g < G; ++g) {
    const float* gp = gt_bboxes + ((size_t)n * max_gt + g) * 4;
    const float x0 = gp[0], y0 = gp[1], x1 = gp[2], y1 = gp[3];
    const float le = px - x0, ri = x1 - px, to = py - y0, bo = y1 - py;
    const float mn = fminf(fminf(le, to), fminf(ri, bo));
    const float mx = fmaxf(fmaxf(le, to), fmaxf(ri, bo));
    bool inside;
    if (cfg.center_sampling) {
      const float cx = (x0 + x1) / 2, cy = (y0 + y1) / 2;
      const float xm = cx - sr, ym = cy - sr, xM = cx + sr, yM = cy + sr;
      const float c0 = xm > x0 ? xm : x0, c1 = ym > y0 ? ym : y0;
      const float c2 = xM > x1 ? x1 : xM, c3 = yM > y1 ? y1 : yM;
      inside = fminf(fminf(px - c0, py - c1), fminf(c2 - px, c3 - py)) > 0.0f;
    } else {
      inside = mn > 0.0f;
    }
    const bool in_range = mx >= cfg.lo[l] && mx <= cfg.hi[l];
    float area = (x1 - x0) * (y1 - y0);
    if (!inside || !in_range) area = INF;
    if (area < best) {
      best = area;
      arg = g;
    }
    in_some = in_some || mn > 0.0f;
  }
  const size_t o = (size_t)n * A + a;
  int64_t lab = num_classes;
  float bt[4] = {0.f, 0.f, 0.f, 0.f}, vl = 0.0f;
  if (G > 0) {
    const float* gp = gt_bboxes + ((size_t)n * max_gt + arg) * 4;
    bt[0] = px - gp[0];
    bt[1] = py - gp[1];
    bt[2] = gp[2] - px;
    bt[3] = gp[3] - py;
    if (best != INF) {
      lab = gt_labels[(size_t)n * max_gt + arg];
      atomicAdd(counts + n, 1);
      atomicAdd(counts + N + l, 1);
    } else if (in_some) {
      vl = 1.0f;
    }
  }
  labels[o] = lab;
  label_weights[o] = 1.0f;  // FCOS has no valid-region mask and no loss weights
  reinterpret_cast<float4*>(bbox_targets)[o] = make_float4(bt[0], bt[1], bt[2], bt[3]);
  vlr[o] = vl;
  im[o] = 0.0f;
}

__global__ void fcos_counts_finalize(int N, int L, int32_t* counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int s = 0;
    for (int i = 0; i < N; ++i) s += counts[i];  // ld_fcos_head.py:186-189
    counts[N + 2 * L] = s;
  }
}

__global__ void grid_anchors_kernel(ld_geom_t geom, float* __restrict__ out) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= geom.num_anchors) return;
  const int l = level_of(geom, a);
  const ld_level_t lv = geom.lv[l];
  const int r = a - lv.offset;
  const int y = r / lv.W, x = r - y * lv.W;
  const Box b = ld::anchor_box(x, y, lv.stride,
                               0.5f * (float)(geom.anchor_scale * lv.stride));
  reinterpret_cast<float4*>(out)[a] = make_float4(b.x1, b.y1, b.x2, b.y2);
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int check_geom(const ld_geom_t* g) {
  if (!g || g->num_levels < 1 || g->num_levels > LD_MAX_LEVELS ||
      g->num_imgs < 1 || g->num_anchors < 1)
    return LD_EINVAL;
  int off = 0;
  for (int l = 0; l < g->num_levels; ++l) {
    if (g->lv[l].offset != off || g->lv[l].H < 1 || g->lv[l].W < 1)
      return LD_EINVAL;
    off += g->lv[l].H * g->lv[l].W;
  }
  return off == g->num_anchors ? 0 : LD_EINVAL;
}

}  // namespace

extern "C" size_t ld_atss_targets_workspace_bytes(const ld_geom_t* geom,
                                                   int max_gt) {
  if (check_geom(geom) != 0 || max_gt < 0) return 0;
  size_t keys = align_up((size_t)geom->num_imgs * geom->num_anchors * 8, 256);
  size_t per_gt = align_up((size_t)geom->num_imgs * (max_gt > 0 ? max_gt : 1) * 4, 256);
  return keys + 2 * per_gt;
}

extern "C" int ld_atss_targets(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                               const float* gt_bboxes, const int64_t* gt_labels,
                               const int32_t* num_gt, int max_gt,
                               const int32_t* valid_hw, int64_t* labels,
                               float* label_weights, float* bbox_targets,
                               float* vlr, float* im, int32_t* counts,
                               void* workspace, size_t workspace_bytes,
                               ld_stream_t stream_) {
  return ld_atss_targets_ex(geom, hp, nullptr, gt_bboxes, gt_labels, num_gt, max_gt,
                            valid_hw, labels, label_weights, bbox_targets, vlr, im,
                            counts, nullptr, nullptr, workspace, workspace_bytes,
                            stream_);
}

extern "C" int ld_atss_targets_ex(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                                  const float* anchors, const float* gt_bboxes,
                                  const int64_t* gt_labels, const int32_t* num_gt,
                                  int max_gt, const int32_t* valid_hw,
                                  int64_t* labels, float* label_weights,
                                  float* bbox_targets, float* vlr, float* im,
                                  int32_t* counts, int64_t* gt_inds,
                                  float* max_overlaps, void* workspace,
                                  size_t workspace_bytes, ld_stream_t stream_) {
  if (int e = check_geom(geom)) return e;
  if (!hp || !labels || !label_weights || !bbox_targets || !vlr || !im ||
      !counts || !num_gt || !valid_hw || max_gt < 0)
    return LD_EINVAL;
  if (max_gt > 0 && (!gt_bboxes || !gt_labels)) return LD_EINVAL;
  if (hp->topk < 1 || hp->topk > 16) return LD_EUNSUPPORTED;
  const size_t need = ld_atss_targets_workspace_bytes(geom, max_gt);
  if (!workspace || workspace_bytes < need) return LD_ENOSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int N = geom->num_imgs, A = geom->num_anchors, L = geom->num_levels;
  char* ws = (char*)workspace;
  const size_t keys_b = align_up((size_t)N * A * 8, 256);
  const size_t per_gt = align_up((size_t)N * (max_gt > 0 ? max_gt : 1) * 4, 256);
  unsigned long long* keys = (unsigned long long*)ws;
  float* thr = (float*)(ws + keys_b);
  float* colmax = (float*)(ws + keys_b + per_gt);
  hipError_t err;
  if ((err = ldrec::memset_async(keys, 0, (size_t)N * A * 8, stream))) return (int)err;
  if ((err = ldrec::memset_async(counts, 0, sizeof(int32_t) * (N + 2 * L + 1), stream)))
    return (int)err;
  if (max_gt > 0) {
    dim3 grid(max_gt, N);
    // implicit anchors: the windowed search (LD_ATSS_SELECT=scan: the full scan)
    const char* sel = getenv("LD_ATSS_SELECT");  // read per call: tests switch it
    const bool window = !(sel && sel[0] == 's');
    if (window && !anchors && hp->topk <= 9 && kMaxCand >= geom->num_levels * hp->topk)
      LD_LAUNCH(atss_select_window_kernel, grid, dim3(kBlock), 0, stream, *geom,
                hp->topk, gt_bboxes, num_gt, max_gt, valid_hw, keys, thr, colmax);
    else if (hp->topk <= 9)
      LD_LAUNCH(atss_select_kernel<9>, grid, dim3(kBlock), 0, stream,
                         *geom, hp->topk, anchors, gt_bboxes, num_gt, max_gt,
                         valid_hw, 1, keys, thr, colmax);
    else
      LD_LAUNCH(atss_select_kernel<16>, grid, dim3(kBlock), 0, stream,
                         *geom, hp->topk, anchors, gt_bboxes, num_gt, max_gt,
                         valid_hw, 1, keys, thr, colmax);
  }
  dim3 gridb((A + kBlock - 1) / kBlock, N);
  LD_LAUNCH(atss_dense_kernel, gridb, dim3(kBlock), 0, stream, *geom,
                     hp->num_classes, (hp->flags & LD_IM_CENTER_INSIDE) ? 1 : 0, anchors,
                     gt_bboxes, gt_labels, num_gt,
                     max_gt > 0 ? max_gt : 1, valid_hw, keys, thr, colmax,
                     labels, label_weights, bbox_targets, vlr, im, counts, gt_inds,
                     max_overlaps);
  LD_LAUNCH(atss_counts_finalize, dim3(1), dim3(64), 0, stream, N, L,
                     counts);
  return (int)hipGetLastError();
}

namespace {
// the reference-ordered anchor list of a B-anchor head as single-anchor levels
// of width W * B
ld_geom_t flatten_geom(const ld_geom_t& g, int B) {
  ld_geom_t f = g;
  for (int l = 0; l < g.num_levels; ++l) {
    f.lv[l].W = g.lv[l].W * B;
    f.lv[l].offset = g.lv[l].offset * B;
  }
  f.num_anchors = g.num_anchors * B;
  return f;
}
}  // namespace

extern "C" size_t ld_retina_targets_workspace_bytes(const ld_geom_t* geom, int num_base,
                                                     int max_gt) {
  if (check_geom(geom) != 0 || num_base < 1 || max_gt < 0) return 0;
  const ld_geom_t f = flatten_geom(*geom, num_base);
  return ld_atss_targets_workspace_bytes(&f, max_gt);
}

extern "C" int ld_retina_targets(const ld_geom_t* geom, int num_base, const float* anchors,
                                 int num_classes, float pos_iou_thr, float neg_iou_thr,
                                 float min_pos_iou, int topk, const float* gt_bboxes,
                                 const int64_t* gt_labels, const int32_t* num_gt,
                                 int max_gt, const int32_t* valid_hw, int64_t* labels,
                                 float* label_weights, float* bbox_targets, float* vlr,
                                 float* im, int32_t* counts, int64_t* gt_inds,
                                 void* workspace, size_t workspace_bytes,
                                 ld_stream_t stream_) {
  if (int e = check_geom(geom)) return e;
  if (num_base < 1 || num_base > 16 || !anchors || num_classes < 1 || !labels ||
      !label_weights || !bbox_targets || !vlr || !im || !counts || !num_gt ||
      !valid_hw || max_gt < 0)
    return LD_EINVAL;
  if (max_gt > 0 && (!gt_bboxes || !gt_labels)) return LD_EINVAL;
  if (topk < 1 || topk > 9) return LD_EUNSUPPORTED;
  const size_t need = ld_retina_targets_workspace_bytes(geom, num_base, max_gt);
  if (!workspace || workspace_bytes < need) return LD_ENOSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const ld_geom_t f = flatten_geom(*geom, num_base);
  const int N = f.num_imgs, AB = f.num_anchors, L = f.num_levels;
  char* ws = (char*)workspace;
  const size_t keys_b = align_up((size_t)N * AB * 8, 256);
  const size_t per_gt = align_up((size_t)N * (max_gt > 0 ? max_gt : 1) * 4, 256);
  unsigned long long* keys = (unsigned long long*)ws;
  float* thr = (float*)(ws + keys_b);
  float* colmax = (float*)(ws + keys_b + per_gt);
  hipError_t err;
  if ((err = ldrec::memset_async(keys, 0, (size_t)N * AB * 8, stream))) return (int)err;
  if ((err = ldrec::memset_async(counts, 0, sizeof(int32_t) * (N * num_base + 2 * L + 1),
                            stream)))
    return (int)err;
  // per gt: the VLR threshold (mean + std IoU of the topk closest anchors of every
  // level) and its best IoU over all valid anchors
  if (max_gt > 0)
    LD_LAUNCH(atss_select_kernel<9>, dim3(max_gt, N), dim3(kBlock), 0, stream, f,
                       topk, anchors, gt_bboxes, num_gt, max_gt, valid_hw, num_base, keys,
                       thr, colmax);
  MaxIouCfg cfg{pos_iou_thr, neg_iou_thr, min_pos_iou, num_base};
  LD_LAUNCH(maxiou_dense_kernel, dim3((AB + kBlock - 1) / kBlock, N),
                     dim3(kBlock), 0, stream, f, num_classes, cfg, anchors, gt_bboxes,
                     gt_labels, num_gt, max_gt > 0 ? max_gt : 1, valid_hw, thr, colmax,
                     labels, label_weights, bbox_targets, vlr, im, counts, gt_inds);
  LD_LAUNCH(maxiou_counts_finalize, dim3(1), dim3(64), 0, stream, N, num_base, L,
                     counts);
  return (int)hipGetLastError();
}

extern "C" int ld_fcos_targets(const ld_geom_t* geom, int num_classes,
                               const float* regress_ranges, int center_sampling,
                               float center_sample_radius, const float* gt_bboxes,
                               const int64_t* gt_labels, const int32_t* num_gt,
                               int max_gt, int64_t* labels, float* label_weights,
                               float* bbox_targets, float* vlr, float* im,
                               int32_t* counts, ld_stream_t stream_) {
  if (int e = check_geom(geom)) return e;
  if (!regress_ranges || !labels || !label_weights || !bbox_targets || !vlr || !im ||
      !counts || !num_gt || max_gt < 0 || num_classes < 1)
    return LD_EINVAL;
  if (max_gt > 0 && (!gt_bboxes || !gt_labels)) return LD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  const int N = geom->num_imgs, A = geom->num_anchors, L = geom->num_levels;
  FcosCfg cfg;
  for (int l = 0; l < LD_MAX_LEVELS; ++l) {
    cfg.lo[l] = l < L ? regress_ranges[2 * l] : 0.0f;
    cfg.hi[l] = l < L ? regress_ranges[2 * l + 1] : 0.0f;
  }
  cfg.radius = center_sample_radius;
  cfg.center_sampling = center_sampling;
  hipError_t err;
  if ((err = ldrec::memset_async(counts, 0, sizeof(int32_t) * (N + 2 * L + 1), stream)))
    return (int)err;
  LD_LAUNCH(fcos_dense_kernel, dim3((A + kBlock - 1) / kBlock, N), dim3(kBlock),
                     0, stream, *geom, num_classes, cfg, gt_bboxes, gt_labels, num_gt,
                     max_gt > 0 ? max_gt : 1, labels, label_weights, bbox_targets, vlr,
                     im, counts);
  LD_LAUNCH(fcos_counts_finalize, dim3(1), dim3(64), 0, stream, N, L, counts);
  return (int)hipGetLastError();
}

extern "C" int ld_grid_anchors(const ld_geom_t* geom, float* anchors,
                               ld_stream_t stream) {
  if (int e = check_geom(geom)) return e;
  if (!anchors) return LD_EINVAL;
  LD_LAUNCH(grid_anchors_kernel,
                     dim3((geom->num_anchors + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, *geom, anchors);
  return (int)hipGetLastError();
}
