// Fused LD loss block for gfx950: forward + gradient of every term of
// LDHead.loss_single (mmdet/models/dense_heads/ld_head.py:116-282), for all
// FPN levels and all images of the rank's batch, read NCHW-direct.
//
// Reference ops replaced (each was 5-15 ATen launches per level):
//   Integral                      gfl_head.py:32-44
//   distance2bbox / bbox2distance core/bbox/transforms.py:119-180
//   bbox_overlaps(aligned)        iou2d_calculator.py:111-171
//   GIoULoss                      losses/iou_loss.py:85-102,334-360
//   DistributionFocalLoss         losses/gfocal_loss.py:53-74
//   QualityFocalLoss              losses/gfocal_loss.py:8-50
//   KnowledgeDistillationKLDivLoss losses/kd_loss.py:10-88  (LD, VLR-LD, KD)
//   IMLoss                        losses/kd_loss.py:91-120
//
// Layout: thread <-> anchor (one spatial cell of one level of one image); the
// channel stride of an NCHW map is H*W, so the 64 lanes of a wavefront read 64
// consecutive floats of every channel they touch: perfectly coalesced with no
// NCHW->NHWC permute (the reference materialises six permuted copies per
// level, ld_head.py:143-154).  Per-anchor softmax / KL over 17 bins runs
// serially in registers; cross-lane traffic is only the final wave reduction
// of the loss partial sums, which are written per block and summed in a fixed
// order by the finalise kernel (bitwise run-to-run reproducible, no float
// atomics).
#include <hip/hip_runtime.h>

#include "ld_launch.h"

#include "../../include/ld_hip.h"
#include "ld_math.h"

namespace {

using ld::Box;
constexpr int K17 = ld::kRegBins;
constexpr int kWave = 64;

// loss partial-sum slots (workspace rows)
enum { S_BBOX = 0, S_DFL, S_LD, S_VLR, S_CLS, S_KD, S_IM, S_WSUM, S_COUNT };

struct BlockMap {  // level-aligned chunking of the anchor axis
  int32_t blk_start[LD_MAX_LEVELS + 1];
  int32_t blocks_per_img;
};

BlockMap make_block_map(const ld_geom_t& g, int per_block) {
  BlockMap m;
  int s = 0;
  for (int l = 0; l < LD_MAX_LEVELS + 1; ++l) m.blk_start[l] = 0;
  for (int l = 0; l < g.num_levels; ++l) {
    m.blk_start[l] = s;
    s += (g.lv[l].H * g.lv[l].W + per_block - 1) / per_block;
  }
  for (int l = g.num_levels; l < LD_MAX_LEVELS + 1; ++l) m.blk_start[l] = s;
  m.blocks_per_img = s;
  return m;
}

__device__ __forceinline__ int block_level(const BlockMap& m, int nl, int b) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < LD_MAX_LEVELS; ++i)
    if (i < nl && b >= m.blk_start[i]) l = i;
  return l;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// divisor of loss_bbox / loss_dfl: LDHead sum(weight_targets) + 1e-6
// (ld_head.py:362-365); LDATSSHead sum(centerness targets), 1 if < EPS
// (ld_atss.py:245-249)
__device__ __forceinline__ float avg_divisor(const ld_loss_hp_t& hp, const float* norm) {
  if (hp.flags & LD_LOSS_ATSS) return norm[1] < 1e-12f ? 1.0f : norm[1];
  // LDRetinaHead: loss_bbox(avg_factor = num_total_samples), ld_retina.py:104-108
  if (hp.flags & LD_LOSS_RETINA) return fmaxf(norm[0], 1.0f);
  return norm[1] + 1e-6f;
}

struct Cell {  // where this thread sits
  int n, l, r, a, x, y;
  bool active;
  size_t o;  // n * A + a
};

__device__ __forceinline__ Cell locate(const ld_geom_t& g, const BlockMap& bm) {
  Cell c;
  c.n = blockIdx.y;
  c.l = block_level(bm, g.num_levels, blockIdx.x);
  const ld_level_t lv = g.lv[c.l];
  c.r = (blockIdx.x - bm.blk_start[c.l]) * kWave + threadIdx.x;
  c.active = c.r < lv.H * lv.W;
  c.a = lv.offset + c.r;
  c.y = c.r / lv.W;
  c.x = c.r - c.y * lv.W;
  c.o = (size_t)c.n * g.num_anchors + c.a;
  return c;
}

__device__ __forceinline__ const float* chan_ptr(const ld_maps_t& m, const Cell& c,
                                                 int ch) {
  return m.ptr[c.l] + (size_t)c.n * m.stride_n[c.l] + (size_t)ch * m.stride_c[c.l] + c.r;
}
__device__ __forceinline__ float* chan_ptr_w(const ld_maps_t& m, const Cell& c,
                                             int ch) {
  return m.ptr[c.l] + (size_t)c.n * m.stride_n[c.l] + (size_t)ch * m.stride_c[c.l] + c.r;
}

__device__ __forceinline__ void load_side(const ld_maps_t& m, const Cell& c,
                                          int side, float* v) {
#pragma unroll
  for (int k = 0; k < K17; ++k) v[k] = *chan_ptr(m, c, side * K17 + k);
}

// ------------------------------------------------------------- prepass ----
// target box in stride units: xyxy / stride (anchor heads) or the point +-
// (l, t, r, b) / stride (FCOS: distance2bbox(points / stride, targets / stride),
// ld_fcos_head.py:82-87)
__device__ __forceinline__ Box target_box(bool fcos, float cx, float cy, const float4& t,
                                          float stride) {
  if (fcos)
    return Box{cx - t.x / stride, cy - t.y / stride, cx + t.z / stride,
               cy + t.w / stride};
  return Box{t.x / stride, t.y / stride, t.z / stride, t.w / stride};
}

__global__ __launch_bounds__(kWave) void loss_prepass_kernel(
    ld_geom_t geom, ld_loss_hp_t hp, BlockMap bm, ld_maps_t cls, ld_maps_t reg,
    const int64_t* __restrict__ labels, const float* __restrict__ bbox_targets,
    const float* __restrict__ vlr, float* __restrict__ weight_targets, float* __restrict__ score,
    float* __restrict__ partial) {
  const Cell c = locate(geom, bm);
  const bool fcos = (hp.flags & LD_LOSS_FCOS) != 0;
  float wt = 0.0f, sc = 0.0f;
  if (c.active) {
    const int64_t lab = labels[c.o];
    const bool is_pos = lab >= 0 && lab < hp.num_classes;
    if (is_pos || (fcos && vlr != nullptr && vlr[c.o] > 0.0f)) {
      const int CC = hp.cls_channels > 0 ? hp.cls_channels : hp.num_classes;
      float m = *chan_ptr(cls, c, 0);
      for (int ch = 1; ch < CC; ++ch) m = fmaxf(m, *chan_ptr(cls, c, ch));
      // LDHead: max_c sigmoid(x_c) == sigmoid(max_c x_c) (ld_head.py:198-199);
      // LDv2Head: the map already holds probabilities (ld_gflv2.py:200);
      // LDFCOSHead also needs it on the "remain" points (ld_fcos_head.py:122)
      wt = (hp.flags & LD_LOSS_PROB_CLS) ? m : ld::sigmoidf_(m);
    }
    if (is_pos) {
      float e[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v[K17], p[K17];
        load_side(reg, c, s, v);
        e[s] = ld::softmax_expect<K17>(v, p);
      }
      const float stride = (float)geom.lv[c.l].stride;
      // anchor centre / stride == (x, y) exactly (ld_head.py:196); FCOS points
      // sit at (x + 0.5, y + 0.5) strides (stride // 2 / stride, even strides)
      const float cx = (float)c.x + (fcos ? 0.5f : 0.0f);
      const float cy = (float)c.y + (fcos ? 0.5f : 0.0f);
      const Box box{cx - e[0], cy - e[1], cx + e[2], cy + e[3]};
      const float4 t = reinterpret_cast<const float4*>(bbox_targets)[c.o];
      const Box tgt = target_box(fcos, cx, cy, t, stride);
      sc = ld::iou_pair(box, tgt);
      if (fcos) {
        // fcos_gfl_head.py:705-721 on the (l, t, r, b) targets
        sc = sqrtf((fminf(t.x, t.z) / fmaxf(t.x, t.z)) *
                   (fminf(t.y, t.w) / fmaxf(t.y, t.w)));
      } else if (hp.flags & LD_LOSS_ATSS) {
        // centerness target (atss_gfl_head.py:312-331), image pixels; the
        // anchor centre is (x, y) * stride exactly
        const float ax = cx * stride, ay = cy * stride;
        const float l_ = ax - t.x, t_ = ay - t.y, r_ = t.z - ax, b_ = t.w - ay;
        sc = sqrtf((fminf(l_, r_) / fmaxf(l_, r_)) * (fminf(t_, b_) / fmaxf(t_, b_)));
      }
    }
    weight_targets[c.o] = wt;
    score[c.o] = sc;
  }
  const float s = wave_sum((hp.flags & LD_LOSS_ATSS) ? sc : wt);
  if (threadIdx.x == 0)
    partial[(size_t)S_WSUM * gridDim.y * bm.blocks_per_img +
            (size_t)c.n * bm.blocks_per_img + blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void loss_norm_kernel(
    int nparts, const float* __restrict__ partial_wsum,
    const int32_t* __restrict__ counts, int idx_total_pos,
    float* __restrict__ norm) {
  __shared__ float s[256];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += partial_wsum[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    norm[0] = (float)counts[idx_total_pos];
    norm[1] = s[0];
  }
}

// ===================== main block: four dense launches =======================
// Work is split so that EVERY launch is a dense, coalesced sweep with
// 256-thread workgroups and one thread per (anchor, channel group):
//   loss_pos_kernel        the few positive anchors only: Integral of the four
//                          sides -> GIoU loss + d/dE, softmax statistics of the
//                          80 class logits for KD -> an 8-float record per anchor
//   loss_reg_dense_kernel  thread = (anchor, SIDE): LD-KL + VLR-LD (+ DFL and the
//                          GIoU chain for positives) over one 17-bin side, forward
//                          and gradient.  This is the north-star fused LD-KL +
//                          Integral sweep of the train step: 34 input streams per
//                          thread, 17 gradient streams out (bench.py measures this
//                          very kernel at a saturating size)
//   loss_cls_dense_kernel  thread = (anchor, 16 class channels): QFL (+ KD)
//   loss_im_dense_kernel   thread = (anchor, 32 feature channels): masked MSE
// Round 1 ran one thread per anchor over all channels in 64-thread blocks: ~700
// wavefronts for 1024 SIMDs and 68-256 dependent iterations each (3-5 % of the
// HBM rate); the (anchor, group) split gives 2 800 / 3 500 / 5 600 wavefronts.
constexpr int kBlk = 256;
constexpr int kZMax = 8;     // channel groups per anchor (sides, class/feature chunks)
constexpr int kClsChunk = 16;
constexpr int kPosRec = 8;   // floats per anchor in the positives record

__device__ __forceinline__ Cell locate256(const ld_geom_t& g, const BlockMap& bm) {
  Cell c;
  c.n = blockIdx.y;
  c.l = block_level(bm, g.num_levels, blockIdx.x);
  const ld_level_t lv = g.lv[c.l];
  c.r = (blockIdx.x - bm.blk_start[c.l]) * kBlk + threadIdx.x;
  c.active = c.r < lv.H * lv.W;
  c.a = lv.offset + c.r;
  c.y = c.r / lv.W;
  c.x = c.r - c.y * lv.W;
  c.o = (size_t)c.n * g.num_anchors + c.a;
  return c;
}

// fixed-order sum over the 256 threads of a block; result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* lds4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();  // lds4 may still be read from the previous call
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// partial[(slot * kZMax + z) * nb + n * blocks_per_img + b]
__device__ __forceinline__ size_t part_idx(int slot, int z, const BlockMap& bm, int n) {
  const size_t nb = (size_t)gridDim.y * bm.blocks_per_img;
  return ((size_t)slot * kZMax + z) * nb + (size_t)n * bm.blocks_per_img + blockIdx.x;
}

// ------------------------------------------------ positives: GIoU + KD stats
__global__ __launch_bounds__(kBlk) void loss_pos_kernel(
    ld_geom_t geom, ld_loss_hp_t hp, BlockMap bm, ld_maps_t cls, ld_maps_t t_cls,
    ld_maps_t reg, const int64_t* __restrict__ labels,
    const float* __restrict__ bbox_targets, const float* __restrict__ weight_targets,
    const float* __restrict__ score, const float* __restrict__ norm,
    const float* __restrict__ upstream, float* __restrict__ posrec,
    float* __restrict__ partial) {
  // cls / t_cls here are the maps of the KD term (LDHead: the class logits;
  // LDv2Head: the raw cls_feat of student and teacher)
  __shared__ float lds4[4];
  const Cell c = locate256(geom, bm);
  float s_bbox = 0.0f;
  if (c.active) {
    const int64_t lab = labels[c.o];
    if (lab >= 0 && lab < hp.num_classes) {
      const int L = geom.num_levels;
      const float up_bbox = upstream ? upstream[1 * L + c.l] : 1.0f;
      const float inv_avg = 1.0f / avg_divisor(hp, norm);
      // GIoU weight: max class score (LDHead) / centerness target (LDATSSHead)
      // ... / 1 (LDRetinaHead: bbox_weights, ld_retina.py:104-108)
      const float wt = (hp.flags & LD_LOSS_RETINA) ? 1.0f
                       : (hp.flags & LD_LOSS_ATSS) ? score[c.o]
                                                   : weight_targets[c.o];
      const float c_bbox = up_bbox * hp.lw_bbox * wt * inv_avg;
      float e[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float sv[K17], p[K17];
        load_side(reg, c, s, sv);
        e[s] = ld::softmax_expect<K17>(sv, p);
      }
      const float stride = (float)geom.lv[c.l].stride;
      const bool fcos = (hp.flags & LD_LOSS_FCOS) != 0;
      const float cx = (float)c.x + (fcos ? 0.5f : 0.0f);
      const float cy = (float)c.y + (fcos ? 0.5f : 0.0f);
      const Box box{cx - e[0], cy - e[1], cx + e[2], cy + e[3]};
      const float4 t = reinterpret_cast<const float4*>(bbox_targets)[c.o];
      const Box tgt = target_box(fcos, cx, cy, t, stride);
      float iou, g[4];
      const float gl = ld::giou_loss_grad(box, tgt, hp.giou_eps, &iou, g);
      s_bbox = wt * gl;
      // softmax statistics of the class logits at temperature T_kd
      const int C = hp.cls_channels > 0 ? hp.cls_channels : hp.num_classes;
      const float invT = 1.0f / hp.T_kd;
      float ms = *chan_ptr(cls, c, 0), mt = *chan_ptr(t_cls, c, 0);
      for (int ch = 1; ch < C; ++ch) {
        ms = fmaxf(ms, *chan_ptr(cls, c, ch));
        mt = fmaxf(mt, *chan_ptr(t_cls, c, ch));
      }
      float zs = 0.0f, zt = 0.0f;
      for (int ch = 0; ch < C; ++ch) {
        zs += expf((*chan_ptr(cls, c, ch) - ms) * invT);
        zt += expf((*chan_ptr(t_cls, c, ch) - mt) * invT);
      }
      float4* rec = reinterpret_cast<float4*>(posrec + c.o * kPosRec);
      rec[0] = make_float4(-g[0] * c_bbox, -g[1] * c_bbox, g[2] * c_bbox, g[3] * c_bbox);
      rec[1] = make_float4(ms, zs, mt, zt);
    }
  }
  s_bbox = block_sum(s_bbox, lds4);
  if (threadIdx.x == 0) partial[part_idx(S_BBOX, 0, bm, c.n)] = s_bbox;
}

// --------------------------- reg side, dense: LD + VLR-LD (+ DFL, GIoU chain)
template <bool NT>
__global__ __launch_bounds__(kBlk) void loss_reg_dense_kernel(
    ld_geom_t geom, ld_loss_hp_t hp, BlockMap bm, ld_maps_t reg, ld_maps_t t_reg,
    const int64_t* __restrict__ labels, const float* __restrict__ bbox_targets,
    const float* __restrict__ vlr, const float* __restrict__ weight_targets,
    const float* __restrict__ norm, const float* __restrict__ upstream,
    const float* __restrict__ posrec, ld_maps_t grad_reg, float* __restrict__ partial) {
  __shared__ float lds4[4];
  const Cell c = locate256(geom, bm);
  const int side = blockIdx.z;
  float s_dfl = 0.0f, s_ld = 0.0f, s_vlr = 0.0f;
  if (c.active) {
    const int64_t lab = labels[c.o];
    const bool pos = lab >= 0 && lab < hp.num_classes;
    // VLR weight: the region value itself (LDHead / LDATSSHead); for LDFCOSHead
    // the "remain" flag times max_c sigmoid(cls) (ld_fcos_head.py:122-123)
    float v = vlr[c.o];
    if ((hp.flags & LD_LOSS_FCOS) && v > 0.0f) v *= weight_targets[c.o];
    const bool retina = (hp.flags & LD_LOSS_RETINA) != 0;
    if (retina && pos) v = 0.0f;  // vlr_weights[labels != bg] = 0, ld_retina.py:103
    const bool rem = v > 0.0f;
    float gr[K17];
#pragma unroll
    for (int k = 0; k < K17; ++k) gr[k] = 0.0f;
    if (pos || rem) {
      const int L = geom.num_levels;
      const float up_dfl = upstream ? upstream[2 * L + c.l] : 1.0f;
      const float up_ld = upstream ? upstream[3 * L + c.l] : 1.0f;
      const float up_vlr = upstream ? upstream[4 * L + c.l] : 1.0f;
      const float wt = pos ? weight_targets[c.o] : 0.0f;
      const float c_ld = up_ld * hp.lw_ld * wt * 0.25f;         // /4.0
      const float c_vlr = up_vlr * hp.lw_ld_vlr * v * 0.0625f;  // /16.0
      float sv[K17], tv[K17], d[K17];
#pragma unroll
      for (int k = 0; k < K17; ++k) {
        const float* ps = chan_ptr(reg, c, side * K17 + k);
        const float* pt = chan_ptr(t_reg, c, side * K17 + k);
        sv[k] = NT ? __builtin_nontemporal_load(ps) : *ps;
        tv[k] = NT ? __builtin_nontemporal_load(pt) : *pt;
      }
      if (retina) {
        // KL over the 68 logits of the anchor (all four sides): the statistics
        // span the other three side-threads' channels too -- re-read here (they
        // are L2 hits); this thread then owns the terms of its own 17 bins
        const float T = hp.T_ld, invT = 1.0f / hp.T_ld;
        float ms = sv[0], mt = tv[0];
        for (int ch = 0; ch < 4 * K17; ++ch) {
          ms = fmaxf(ms, *chan_ptr(reg, c, ch));
          mt = fmaxf(mt, *chan_ptr(t_reg, c, ch));
        }
        float zs = 0.0f, zt = 0.0f;
        for (int ch = 0; ch < 4 * K17; ++ch) {
          zs += expf((*chan_ptr(reg, c, ch) - ms) * invT);
          zt += expf((*chan_ptr(t_reg, c, ch) - mt) * invT);
        }
        const float lzs = logf(zs), lzt = logf(zt), rzs = 1.0f / zs, rzt = 1.0f / zt;
        float kl = 0.0f;
#pragma unroll
        for (int k = 0; k < K17; ++k) {
          const float as = (sv[k] - ms) * invT, at = (tv[k] - mt) * invT;
          const float ps = expf(as) * rzs, pt = expf(at) * rzt;
          kl += pt * ((at - lzt) - (as - lzs));
          d[k] = ps - pt;
        }
        kl *= T * T / (float)(4 * K17);
        s_ld = wt * kl;
        s_vlr = v * kl;
        const float cg = (c_ld + (rem ? c_vlr : 0.0f)) * (T / (float)(4 * K17));
#pragma unroll
        for (int k = 0; k < K17; ++k) gr[k] = cg * d[k];
      } else if (hp.T_ld == hp.T_ld_vlr) {
        const float T = hp.T_ld;
        const float kl = ld::kl_rows<K17>(sv, tv, 1.0f / T, T, d);
        s_ld = wt * kl;
        s_vlr = v * kl;
        const float cg = (c_ld + (rem ? c_vlr : 0.0f)) * (T / (float)K17);
#pragma unroll
        for (int k = 0; k < K17; ++k) gr[k] = cg * d[k];
      } else {
        if (pos) {
          const float T = hp.T_ld;
          const float kl = ld::kl_rows<K17>(sv, tv, 1.0f / T, T, d);
          s_ld = wt * kl;
          const float cg = c_ld * (T / (float)K17);
#pragma unroll
          for (int k = 0; k < K17; ++k) gr[k] += cg * d[k];
        }
        if (rem) {
          const float T = hp.T_ld_vlr;
          const float kl = ld::kl_rows<K17>(sv, tv, 1.0f / T, T, d);
          s_vlr = v * kl;
          const float cg = c_vlr * (T / (float)K17);
#pragma unroll
          for (int k = 0; k < K17; ++k) gr[k] += cg * d[k];
        }
      }
      if (pos) {
        const float inv_avg = 1.0f / avg_divisor(hp, norm);
        const float c_dfl = up_dfl * hp.lw_dfl * wt * 0.25f * inv_avg;
        const float stride = (float)geom.lv[c.l].stride;
        const bool fcos = (hp.flags & LD_LOSS_FCOS) != 0;
        const float cx = (float)c.x + (fcos ? 0.5f : 0.0f);
        const float cy = (float)c.y + (fcos ? 0.5f : 0.0f);
        const float4 t = reinterpret_cast<const float4*>(bbox_targets)[c.o];
        const Box tb = target_box(fcos, cx, cy, t, stride);
        const float rm = (float)hp.reg_max;
        const float dist = side == 0   ? cx - tb.x1
                           : side == 1 ? cy - tb.y1
                           : side == 2 ? tb.x2 - cx
                                       : tb.y2 - cy;
        const float ytgt = ld::clamp_dist(dist, rm);
        float p[K17];
        const float e = ld::softmax_expect<K17>(sv, p);
        float wl, wr;
        int yl;
        const float dl = ld::dfl_side<K17>(sv, p, ytgt, &wl, &wr, &yl);
        s_dfl = wt * dl;
        const float gd = posrec[c.o * kPosRec + side];  // d total / d E_side (GIoU)
#pragma unroll
        for (int k = 0; k < K17; ++k) {
          const float gk = p[k] - (k == yl ? wl : 0.0f) - (k == yl + 1 ? wr : 0.0f);
          gr[k] += c_dfl * gk + gd * p[k] * ((float)k - e);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < K17; ++k) {
      float* pg = chan_ptr_w(grad_reg, c, side * K17 + k);
      if (NT)
        __builtin_nontemporal_store(gr[k], pg);
      else
        *pg = gr[k];
    }
  }
  s_dfl = block_sum(s_dfl, lds4);
  s_ld = block_sum(s_ld, lds4);
  s_vlr = block_sum(s_vlr, lds4);
  if (threadIdx.x == 0) {
    partial[part_idx(S_DFL, side, bm, c.n)] = s_dfl;
    partial[part_idx(S_LD, side, bm, c.n)] = s_ld;
    partial[part_idx(S_VLR, side, bm, c.n)] = s_vlr;
  }
}

typedef __amdgpu_buffer_rsrc_t lrsrc_t;
__device__ __forceinline__ lrsrc_t lean_rsrc(const void* p, unsigned bytes) {
  const uintptr_t u = (uintptr_t)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  void* q = (void*)(((uintptr_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes),
                                           0x00020000);
}
// R floats per access (float / float2 / float4)
template <int R>
struct VecT;
template <>
struct VecT<1> { typedef float T; };
template <>
struct VecT<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <>
struct VecT<4> { typedef float T __attribute__((ext_vector_type(4))); };

// buffer access of R floats: lane byte offset in a VGPR, plane offset in an SGPR;
// aux 2 = the non-temporal ("nt") cache policy
template <int R, bool NT>
__device__ __forceinline__ typename VecT<R>::T lean_load(lrsrc_t r, unsigned voff,
                                                         unsigned soff) {
  typedef typename VecT<R>::T V;
  soff = __builtin_amdgcn_readfirstlane(soff);
  if constexpr (R == 1)
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, NT ? 2 : 0));
  else if constexpr (R == 2)
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, NT ? 2 : 0));
  else
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, NT ? 2 : 0));
}
template <int R, bool NT>
__device__ __forceinline__ void lean_store(typename VecT<R>::T v, lrsrc_t r, unsigned voff,
                                           unsigned soff) {
  soff = __builtin_amdgcn_readfirstlane(soff);
  if constexpr (R == 1)
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, NT ? 2 : 0);
  else if constexpr (R == 2) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, voff, soff, NT ? 2 : 0);
  } else {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, voff, soff, NT ? 2 : 0);
  }
}

// ------------------- reg side, dense, register-lean (round 3, the default) ----
// Same terms as loss_reg_dense_kernel for every head whose LD and VLR-LD terms
// share one temperature and one 17-bin softmax per side (LDHead, LDv2Head,
// LDATSSHead, LDFCOSHead); LDRetinaHead and T_ld != T_ld_vlr keep the kernel
// above.  What changed, and why (0.66-0.68 of 8 TB/s at 1.05 x
// the algorithmic traffic = an occupancy / issue problem, not a bytes problem):
//   * the kernel above holds 149 VGPRs (+ 8 B scratch): 3 waves per SIMD.  Here
//     the KL runs in place on the two logit rows (ld::kl_rows_inplace, 34 live
//     values per anchor) and the rare positive anchors RE-LOAD their 17 student
//     logits for the DFL + GIoU chain instead of keeping them live through the
//     KL -- the common path fits 64 VGPRs at VEC = 1: 8 waves per SIMD;
//   * VEC consecutive anchors of a plane per thread (8 / 16-byte accesses, a
//     wavefront request covers 512 B / 1 KiB of each channel plane);
//   * loads and stores take their non-temporal flavour separately (NTL / NTS);
//   * side_fast: the 4 sides of an anchor chunk in 4 ADJACENT workgroups (they
//     share the label / VLR lines) instead of 4 sweeps over the planes.
// The (VEC, NTL, NTS, FAST, side_fast) actually launched is one measured
// choice (g_reg_variant below; profiles/r03_ldkl_variants.json).
template <int VEC, bool NTL, bool NTS, bool FAST, int WPE>
__global__ __launch_bounds__(kBlk, WPE) void loss_reg_lean_kernel(
    ld_geom_t geom, ld_loss_hp_t hp, BlockMap bm, BlockMap bmv, int side_fast,
    ld_maps_t reg, ld_maps_t t_reg, const int64_t* __restrict__ labels,
    const float* __restrict__ bbox_targets, const float* __restrict__ vlr,
    const float* __restrict__ weight_targets, const float* __restrict__ norm,
    const float* __restrict__ upstream, const float* __restrict__ posrec,
    ld_maps_t grad_reg, float* __restrict__ partial) {
  typedef typename VecT<VEC>::T V;
  __shared__ float lds4[4];
  int side = (int)blockIdx.z, bxv = (int)blockIdx.x;
  if (side_fast == 1) {
    side = (int)(blockIdx.x & 3);
    bxv = (int)(blockIdx.x >> 2);
  } else if (side_fast == 2) {
    // the 4 sides of a chunk on ONE XCD (block b runs on XCD b % 8), back to
    // back: they share the chunk's label / VLR lines in that XCD's L2.  Blocks
    // beyond the last full group of 32 keep the plain side-fast order.
    const unsigned b = blockIdx.x, full = (gridDim.x >> 5) << 5;
    if (b < full) {
      const unsigned q = b >> 3, x = b & 7;
      side = (int)(q & 3);
      bxv = (int)((q >> 2) * 8 + x);
    } else {
      side = (int)(b & 3);
      bxv = (int)(b >> 2);
    }
  }
  const int n = blockIdx.y;
  const int l = block_level(bmv, geom.num_levels, bxv);
  const ld_level_t lv = geom.lv[l];
  const int HW = lv.H * lv.W;
  const int wg = bxv - bmv.blk_start[l];
  const int r0 = (wg * kBlk + (int)threadIdx.x) * VEC;
  const size_t o0 = (size_t)n * geom.num_anchors + lv.offset + r0;
  // one buffer descriptor per map (base = this image's level map, wave-uniform)
  // + a 32-bit lane offset + the channel-plane offset in an SGPR: no per-channel
  // 64-bit address registers (the flat form cost 2 x 34 VGPRs)
  const size_t cs = reg.stride_c[l], ct = t_reg.stride_c[l], cg_ = grad_reg.stride_c[l];
  const float* ps0 = reg.ptr[l] + (size_t)n * reg.stride_n[l];
  const float* pt0 = t_reg.ptr[l] + (size_t)n * t_reg.stride_n[l];
  float* pg0 = grad_reg.ptr[l] + (size_t)n * grad_reg.stride_n[l];
  const lrsrc_t rs = lean_rsrc(ps0, (unsigned)(cs * 4 * (4 * K17)));
  const lrsrc_t rt = lean_rsrc(pt0, (unsigned)(ct * 4 * (4 * K17)));
  const lrsrc_t rg = lean_rsrc(pg0, (unsigned)(cg_ * 4 * (4 * K17)));
  const unsigned ro = (unsigned)r0 * 4u;  // byte offset of this thread in a plane
  const unsigned cs4 = (unsigned)(cs * 4), ct4 = (unsigned)(ct * 4), cg4 = (unsigned)(cg_ * 4);
  const unsigned ss = (unsigned)(side * K17) * cs4;
  const unsigned st = (unsigned)(side * K17) * ct4;
  const unsigned sg = (unsigned)(side * K17) * cg4;
  // vector access needs every plane of this level aligned for this thread
  const bool full = r0 + VEC <= HW;
  bool vec_ok = full;
  if (VEC > 1) {
    const uintptr_t m = (uintptr_t)(VEC * 4 - 1);
    vec_ok = full && (((uintptr_t)ps0 | (uintptr_t)pt0 | (uintptr_t)pg0 |
                       (uintptr_t)cs4 | (uintptr_t)ct4 | (uintptr_t)cg4) & m) == 0;
  }
  float wt[VEC], vv[VEC];
  bool pos[VEC], act[VEC];
  bool need = false, anypos = false;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    act[j] = r0 + j < HW;
    pos[j] = false;
    wt[j] = 0.0f;
    vv[j] = 0.0f;
    if (act[j]) {
      const int64_t lab = labels[o0 + j];
      pos[j] = lab >= 0 && lab < hp.num_classes;
      float v = vlr[o0 + j];
      if ((hp.flags & LD_LOSS_FCOS) && v > 0.0f) v *= weight_targets[o0 + j];
      vv[j] = v > 0.0f ? v : 0.0f;
      if (pos[j]) wt[j] = weight_targets[o0 + j];
      need = need || pos[j] || v > 0.0f;
      anypos = anypos || pos[j];
    }
  }
  const int L = geom.num_levels;
  const float T = hp.T_ld;
  const float k_ld = (upstream ? upstream[3 * L + l] : 1.0f) * hp.lw_ld * 0.25f *
                     (T / (float)K17);  // /4.0
  const float k_vlr = (upstream ? upstream[4 * L + l] : 1.0f) * hp.lw_ld_vlr * 0.0625f *
                      (T / (float)K17);  // /16.0
  float s_dfl = 0.0f, s_ld = 0.0f, s_vlr = 0.0f;
  if (vec_ok && !anypos) {
    // ---- the dense path: VEC anchors, vector access, no positives ----------
    float sv[VEC][K17], tv[VEC][K17];
    if (need) {
#pragma unroll
      for (int k = 0; k < K17; ++k) {
        const V a = lean_load<VEC, NTL>(rs, ro, ss + k * cs4);
        const V b = lean_load<VEC, NTL>(rt, ro, st + k * ct4);
        const float* af = reinterpret_cast<const float*>(&a);
        const float* bf = reinterpret_cast<const float*>(&b);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          sv[j][k] = af[j];
          tv[j][k] = bf[j];
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float kl = ld::kl_rows_inplace<K17, FAST>(sv[j], tv[j], 1.0f / T, T);
        s_vlr += vv[j] * kl;  // wt = 0 here: no LD term away from the positives
        const float cg = k_vlr * vv[j];
#pragma unroll
        for (int k = 0; k < K17; ++k) sv[j][k] *= cg;
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j)
#pragma unroll
        for (int k = 0; k < K17; ++k) sv[j][k] = 0.0f;
    }
#pragma unroll
    for (int k = 0; k < K17; ++k) {
      V g;
      float* gf = reinterpret_cast<float*>(&g);
#pragma unroll
      for (int j = 0; j < VEC; ++j) gf[j] = sv[j][k];
      lean_store<VEC, NTS>(g, rg, ro, sg + k * cg4);
    }
  } else {
    // ---- threads holding a positive anchor, an unaligned level or the level
    // tail: one anchor at a time, scalar access.  Positives (~0.1 % of the
    // anchors) add DFL + the GIoU chain through Integral from the T = 1
    // softmax of the student row, read again after the KL is done with it.
    for (int j = 0; j < VEC; ++j) {
      if (r0 + j >= HW) break;
      const size_t o = o0 + j;
      const int64_t lab = labels[o];
      const bool p_ = lab >= 0 && lab < hp.num_classes;
      float v = vlr[o];
      if ((hp.flags & LD_LOSS_FCOS) && v > 0.0f) v *= weight_targets[o];
      v = v > 0.0f ? v : 0.0f;
      const float w = p_ ? weight_targets[o] : 0.0f;
      const unsigned rj = ro + 4u * (unsigned)j;
      float a[K17];
      if (p_ || v > 0.0f) {
        float b[K17];
#pragma unroll
        for (int k = 0; k < K17; ++k) {
          a[k] = lean_load<1, false>(rs, rj, ss + k * cs4);
          b[k] = lean_load<1, false>(rt, rj, st + k * ct4);
        }
        const float kl = ld::kl_rows_inplace<K17, FAST>(a, b, 1.0f / T, T);
        s_ld += w * kl;
        s_vlr += v * kl;
        const float cg = k_ld * w + k_vlr * v;
#pragma unroll
        for (int k = 0; k < K17; ++k) a[k] *= cg;
      } else {
#pragma unroll
        for (int k = 0; k < K17; ++k) a[k] = 0.0f;
      }
      if (p_) {
        // (scheduling fence: hoisting these loads above the KL would keep 51
        // row values live at once and set the whole kernel's VGPR count)
        __builtin_amdgcn_sched_barrier(0);
        float p[K17];
#pragma unroll
        for (int k = 0; k < K17; ++k) p[k] = lean_load<1, false>(rs, rj, ss + k * cs4);
        const float up_dfl = upstream ? upstream[2 * L + l] : 1.0f;
        const float inv_avg = 1.0f / avg_divisor(hp, norm);
        const float c_dfl = up_dfl * hp.lw_dfl * w * 0.25f * inv_avg;
        const float stride = (float)lv.stride;
        const bool fcos = (hp.flags & LD_LOSS_FCOS) != 0;
        const int r = r0 + j, y = r / lv.W, x = r - y * lv.W;
        const float cx = (float)x + (fcos ? 0.5f : 0.0f);
        const float cy = (float)y + (fcos ? 0.5f : 0.0f);
        const float4 t4 = reinterpret_cast<const float4*>(bbox_targets)[o];
        const Box tb = target_box(fcos, cx, cy, t4, stride);
        const float dist = side == 0   ? cx - tb.x1
                           : side == 1 ? cy - tb.y1
                           : side == 2 ? tb.x2 - cx
                                       : tb.y2 - cy;
        const float ytgt = ld::clamp_dist(dist, (float)hp.reg_max);
        float e, wl, wr;
        int yl;
        const float dl = ld::softmax_dfl_inplace<K17>(p, ytgt, &e, &wl, &wr, &yl);
        s_dfl += w * dl;
        const float gd = posrec[o * kPosRec + side];  // d total / d E_side (GIoU)
#pragma unroll
        for (int k = 0; k < K17; ++k) {
          const float gk = p[k] - (k == yl ? wl : 0.0f) - (k == yl + 1 ? wr : 0.0f);
          a[k] += c_dfl * gk + gd * p[k] * ((float)k - e);
        }
      }
#pragma unroll
      for (int k = 0; k < K17; ++k) lean_store<1, false>(a[k], rg, rj, sg + k * cg4);
    }
  }
  s_dfl = block_sum(s_dfl, lds4);
  s_ld = block_sum(s_ld, lds4);
  s_vlr = block_sum(s_vlr, lds4);
  if (threadIdx.x == 0) {
    // this workgroup covers VEC 256-anchor slots of the finalise kernel's map
    const size_t nb = (size_t)gridDim.y * bm.blocks_per_img;
    const int b0 = bm.blk_start[l] + wg * VEC, b1 = bm.blk_start[l + 1];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      if (b0 + j >= b1) break;
      const size_t at = (size_t)n * bm.blocks_per_img + b0 + j;
      partial[((size_t)S_DFL * kZMax + side) * nb + at] = j == 0 ? s_dfl : 0.0f;
      partial[((size_t)S_LD * kZMax + side) * nb + at] = j == 0 ? s_ld : 0.0f;
      partial[((size_t)S_VLR * kZMax + side) * nb + at] = j == 0 ? s_vlr : 0.0f;
    }
  }
}

// ------------------------------------------- cls side, dense: QFL (+ KD) -----
// SPLIT_KD = false (LDHead): KD runs on the class logits themselves and its
// gradient adds into grad_cls.  SPLIT_KD = true (LDv2Head): cls holds the
// probabilities cls_score (QFL via binary_cross_entropy), the KD term runs on
// kd_s / kd_t (raw cls_feat) and its gradient is written to grad_kd (zeros away
// from the positives).
template <bool SPLIT_KD>
__global__ __launch_bounds__(kBlk) void loss_cls_dense_kernel(
    ld_geom_t geom, ld_loss_hp_t hp, BlockMap bm, ld_maps_t cls, ld_maps_t kd_s,
    ld_maps_t kd_t, const int64_t* __restrict__ labels,
    const float* __restrict__ label_weights, const float* __restrict__ score,
    const int32_t* __restrict__ counts, const float* __restrict__ norm,
    const float* __restrict__ upstream, const float* __restrict__ posrec,
    ld_maps_t grad_cls, ld_maps_t grad_kd, float* __restrict__ partial) {
  __shared__ float lds4[4];
  const Cell c = locate256(geom, bm);
  const int NC = hp.num_classes;  // foreground classes: labels in [0, NC)
  const int C = hp.cls_channels > 0 ? hp.cls_channels : NC;
  const bool prob = (hp.flags & LD_LOSS_PROB_CLS) != 0;
  const bool focal = (hp.flags & (LD_LOSS_ATSS | LD_LOSS_RETINA)) != 0;
  const int ch0 = blockIdx.z * kClsChunk;
  const int ch1 = min(C, ch0 + kClsChunk);
  float s_cls = 0.0f, s_kd = 0.0f;
  if (c.active) {
    const float lw = label_weights[c.o];
    if (lw == 0.0f) {  // outside the image's valid region
      for (int ch = ch0; ch < ch1; ++ch) {
        *chan_ptr_w(grad_cls, c, ch) = 0.0f;
        if (SPLIT_KD) *chan_ptr_w(grad_kd, c, ch) = 0.0f;
      }
    } else {
      const int64_t lab = labels[c.o];
      const bool pos = lab >= 0 && lab < NC;
      const float up_cls = upstream ? upstream[0 * geom.num_levels + c.l] : 1.0f;
      const float up_kd = upstream ? upstream[5 * geom.num_levels + c.l] : 1.0f;
      const float nts = fmaxf(norm[0], 1.0f);  // ld_head.py:341
      const float c_cls = up_cls * hp.lw_cls * lw / nts;
      const float sc = pos ? score[c.o] : 0.0f;
      const float T = hp.T_kd, invT = 1.0f / hp.T_kd;
      float ms = 0, mt = 0, rzs = 0, rzt = 0, lzs = 0, lzt = 0, c_kd = 0;
      if (pos) {
        const float4 st = reinterpret_cast<const float4*>(posrec + c.o * kPosRec)[1];
        ms = st.x;
        mt = st.z;
        rzs = 1.0f / st.y;
        rzt = 1.0f / st.w;
        lzs = logf(st.y);
        lzt = logf(st.w);
        const int P_l = counts[geom.num_imgs + c.l];  // >= 1 here
        c_kd = up_kd * hp.lw_kd * lw / (float)P_l * (T / (float)C);
      }
      float rowsum = 0.0f, kl = 0.0f;
#pragma unroll 4
      for (int ch = ch0; ch < ch1; ++ch) {
        const float x = *chan_ptr(cls, c, ch);
        float dq, q;
        if (focal) {
          q = (pos && ch == (int)lab) ? ld::focal_pos(x, hp.focal_alpha, &dq)
                                      : ld::focal_neg(x, hp.focal_alpha, &dq);
        } else if (prob) {
          // gfocal_loss.py:41 takes every label < pred.size(1) as "positive",
          // background (label = num_classes, score 0) included; for it the
          // positive formula bce(p, 0) |0 - p|^2 IS the negative one
          if (pos && ch == (int)lab)
            q = ld::qfl_prob_pos(x, sc, &dq);
          else
            q = ld::qfl_prob_neg(x, &dq);
        } else if (pos && ch == (int)lab) {
          q = ld::qfl_pos(x, sc, &dq);
        } else {
          q = ld::qfl_neg(x, &dq);
        }
        rowsum += q;
        float g = c_cls * dq;
        float gk = 0.0f;
        if (pos) {
          const float xs = SPLIT_KD ? *chan_ptr(kd_s, c, ch) : x;
          const float t = *chan_ptr(kd_t, c, ch);
          const float ps = expf((xs - ms) * invT) * rzs;
          const float pt = expf((t - mt) * invT) * rzt;
          kl += pt * (((t - mt) * invT - lzt) - ((xs - ms) * invT - lzs));
          gk = c_kd * (ps - pt);
        }
        if (SPLIT_KD) {
          *chan_ptr_w(grad_kd, c, ch) = gk;
        } else {
          g += gk;
        }
        *chan_ptr_w(grad_cls, c, ch) = g;
      }
      s_cls = lw * rowsum;
      if (pos) s_kd = lw * kl * (T * T) / (float)C;
    }
  }
  s_cls = block_sum(s_cls, lds4);
  s_kd = block_sum(s_kd, lds4);
  if (threadIdx.x == 0) {
    partial[part_idx(S_CLS, blockIdx.z, bm, c.n)] = s_cls;
    partial[part_idx(S_KD, blockIdx.z, bm, c.n)] = s_kd;
  }
}

// ------------------------------------------------------ IM (MSE), dense ------
__global__ __launch_bounds__(kBlk) void loss_im_dense_kernel(
    ld_geom_t geom, ld_loss_hp_t hp, BlockMap bm, ld_maps_t x, ld_maps_t t_x,
    const float* __restrict__ im, const int32_t* __restrict__ counts,
    const float* __restrict__ upstream, ld_maps_t grad_x, int chunk,
    float* __restrict__ partial) {
  __shared__ float lds4[4];
  const Cell c = locate256(geom, bm);
  const int CH = hp.feat_channels;
  const int ch0 = blockIdx.z * chunk, ch1 = min(CH, ch0 + chunk);
  float s_im = 0.0f;
  if (c.active) {
    const int P_l = counts[geom.num_imgs + c.l];
    const int F_l = counts[geom.num_imgs + geom.num_levels + c.l];
    // ld_head.py:186-191 and :246-252 (quirk Q5: no positives -> loss_im = 0)
    const bool enabled = P_l > 0 && F_l > 0 && hp.lw_im != 0.0f;
    const bool sel = enabled && im[c.o] > 0.0f;
    if (!sel) {
#pragma unroll 8
      for (int ch = ch0; ch < ch1; ++ch) *chan_ptr_w(grad_x, c, ch) = 0.0f;
    } else {
      const float up = upstream ? upstream[7 * geom.num_levels + c.l] : 1.0f;
      const float cg = up * hp.lw_im * 2.0f / ((float)F_l * (float)CH);
#pragma unroll 8
      for (int ch = ch0; ch < ch1; ++ch) {
        const float d = *chan_ptr(x, c, ch) - *chan_ptr(t_x, c, ch);
        s_im += d * d;
        *chan_ptr_w(grad_x, c, ch) = cg * d;
      }
    }
  }
  s_im = block_sum(s_im, lds4);
  if (threadIdx.x == 0) partial[part_idx(S_IM, blockIdx.z, bm, c.n)] = s_im;
}

// ------------------------------------- 'gibox' imitation region (GI boxes) ---
// LDHead.get_gi_region (ld_head.py:613-637; LDv2Head: ld_gflv2.py:619-644
// without the sigmoids): per level, over ALL cells of all images,
//   z = teacher_score - student_score, giscore = max_c |z|, the GI box is the
//   teacher's decoded box where the teacher wins (z >= 0 at the arg-max class)
//   and the student's otherwise; keep the first `topn` boxes of a greedy NMS
//   (IoU > thr suppresses, descending giscore) -- torchvision.ops.nms(...)[:10].
// Two launches: a dense scoring sweep (thread = cell) and one 1024-thread
// workgroup per level that extracts the topn survivors by repeated
// arg-max + suppression (O(topn * cells), never the sort or the O(M^2) mask).
__global__ __launch_bounds__(kBlk) void gi_score_kernel(
    ld_geom_t geom, ld_loss_hp_t hp, BlockMap bm, ld_maps_t cls, ld_maps_t t_cls,
    ld_maps_t reg, ld_maps_t t_reg, float* __restrict__ giscore,
    float* __restrict__ gibox) {
  const Cell c = locate256(geom, bm);
  if (!c.active) return;
  const int C = hp.cls_channels > 0 ? hp.cls_channels : hp.num_classes;
  const bool prob = (hp.flags & LD_LOSS_PROB_CLS) != 0;
  float best = -1.0f, zbest = 0.0f;
  for (int ch = 0; ch < C; ++ch) {
    const float xs = *chan_ptr(cls, c, ch), xt = *chan_ptr(t_cls, c, ch);
    const float z = prob ? xt - xs : ld::sigmoidf_(xt) - ld::sigmoidf_(xs);
    const float az = fabsf(z);
    if (az > best) {  // first maximal class
      best = az;
      zbest = z;
    }
  }
  const ld_maps_t& src = zbest >= 0.0f ? t_reg : reg;  // who is bigger
  float e[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float v[K17], p[K17];
    load_side(src, c, s, v);
    e[s] = ld::softmax_expect<K17>(v, p);
  }
  const float cx = (float)c.x, cy = (float)c.y;  // anchor centre / stride
  giscore[c.o] = best;
  reinterpret_cast<float4*>(gibox)[c.o] =
      make_float4(cx - e[0], cy - e[1], cx + e[2], cy + e[3]);
}

__global__ __launch_bounds__(1024) void gi_select_kernel(
    ld_geom_t geom, int topn, float iou_thr, float* __restrict__ giscore,
    const float* __restrict__ gibox, float* __restrict__ im,
    int32_t* __restrict__ counts) {
  __shared__ float s_val[1024];
  __shared__ int s_idx[1024];
  __shared__ float4 s_keep;
  __shared__ int s_pick;
  const int l = blockIdx.x, t = threadIdx.x;
  const ld_level_t lv = geom.lv[l];
  const int Al = lv.H * lv.W, N = geom.num_imgs, A = geom.num_anchors;
  const int M = N * Al;
  auto flat = [&](int m) -> size_t {  // candidate m = n * A_l + r -> (n, anchor)
    const int n = m / Al, r = m - n * Al;
    return (size_t)n * A + lv.offset + r;
  };
  for (int m = t; m < M; m += 1024) im[flat(m)] = 0.0f;
  int kept = 0;
  for (int round = 0; round < topn; ++round) {
    float bv = -1.0f;
    int bi = 0x7fffffff;
    for (int m = t; m < M; m += 1024) {
      const float v = giscore[flat(m)];
      if (v > bv) {  // ascending m per thread: first maximum
        bv = v;
        bi = m;
      }
    }
    s_val[t] = bv;
    s_idx[t] = bi;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
      if (t < w) {
        const float ov = s_val[t + w];
        const int oi = s_idx[t + w];
        if (ov > s_val[t] || (ov == s_val[t] && oi < s_idx[t])) {
          s_val[t] = ov;
          s_idx[t] = oi;
        }
      }
      __syncthreads();
    }
    if (t == 0) {
      s_pick = s_val[0] >= 0.0f ? s_idx[0] : -1;
      if (s_pick >= 0) {
        const size_t o = flat(s_pick);
        s_keep = reinterpret_cast<const float4*>(gibox)[o];
        im[o] = 1.0f;
      }
    }
    __syncthreads();
    if (s_pick < 0) break;  // uniform
    ++kept;
    const float4 k = s_keep;
    const float ka = (k.z - k.x) * (k.w - k.y);
    for (int m = t; m < M; m += 1024) {
      const size_t o = flat(m);
      if (giscore[o] < 0.0f) continue;
      if (m == s_pick) {
        giscore[o] = -1.0f;
        continue;
      }
      const float4 b = reinterpret_cast<const float4*>(gibox)[o];
      const float iw = fmaxf(fminf(k.z, b.z) - fmaxf(k.x, b.x), 0.0f);
      const float ih = fmaxf(fminf(k.w, b.w) - fmaxf(k.y, b.y), 0.0f);
      const float inter = iw * ih;
      const float iou = inter / (ka + (b.z - b.x) * (b.w - b.y) - inter);
      if (iou > iou_thr) giscore[o] = -1.0f;
    }
    __syncthreads();
  }
  if (t == 0) counts[N + geom.num_levels + l] = kept;
}

// ------------------------------- centerness BCE on the positives (ATSS) -----
__global__ __launch_bounds__(kBlk) void loss_ctr_dense_kernel(
    ld_geom_t geom, ld_loss_hp_t hp, BlockMap bm, ld_maps_t ctr,
    const int64_t* __restrict__ labels, const float* __restrict__ score,
    const float* __restrict__ norm, const float* __restrict__ upstream,
    ld_maps_t grad_ctr, float* __restrict__ partial) {
  __shared__ float lds4[4];
  const Cell c = locate256(geom, bm);
  float s_ctr = 0.0f;
  if (c.active) {
    const int64_t lab = labels[c.o];
    float g = 0.0f;
    if (lab >= 0 && lab < hp.num_classes) {
      const float up = upstream ? upstream[6 * geom.num_levels + c.l] : 1.0f;
      const float nts = fmaxf(norm[0], 1.0f);
      float d;
      s_ctr = ld::bce_logits(*chan_ptr(ctr, c, 0), score[c.o], &d);
      g = up * hp.lw_ctr * d / nts;
    }
    *chan_ptr_w(grad_ctr, c, 0) = g;
  }
  s_ctr = block_sum(s_ctr, lds4);
  if (threadIdx.x == 0) partial[part_idx(S_WSUM, 0, bm, c.n)] = s_ctr;
}

// ------------------------------------------------------------ finalise ------
// Fixed-order sum of the per-block partials of level l: [slot][z][n][b].
__global__ __launch_bounds__(kWave) void loss_finalize_kernel(
    ld_geom_t geom, ld_loss_hp_t hp, BlockMap bm, int zcls, int zim,
    const int32_t* __restrict__ counts, const float* __restrict__ norm,
    const float* __restrict__ partial, float* __restrict__ losses) {
  const int L = geom.num_levels, N = geom.num_imgs;
  const int l = blockIdx.x;
  const size_t nb = (size_t)N * bm.blocks_per_img;
  const int b0 = bm.blk_start[l], b1 = bm.blk_start[l + 1];
  const int per = b1 - b0;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // S_WSUM of the MAIN partials = the centerness term (LD_LOSS_ATSS)
    const int Z = (k == S_BBOX || k == S_WSUM) ? 1
                  : (k == S_CLS || k == S_KD)  ? zcls
                  : k == S_IM                  ? zim
                                               : 4;
    if (k == S_WSUM && !(hp.flags & LD_LOSS_ATSS)) {
      acc[k] = 0.0f;
      continue;
    }
    float a = 0.0f;
    for (int i = threadIdx.x; i < per * N * Z; i += kWave) {
      const int z = i / (per * N), r = i - z * per * N;
      const int n = r / per, b = b0 + (r - n * per);
      a += partial[((size_t)k * kZMax + z) * nb + (size_t)n * bm.blocks_per_img + b];
    }
    acc[k] = wave_sum(a);
  }
  if (threadIdx.x == 0) {
    const float nts = fmaxf(norm[0], 1.0f);
    const float avg = avg_divisor(hp, norm);
    const int P_l = counts[N + l], F_l = counts[N + L + l];
    losses[0 * L + l] = hp.lw_cls * acc[S_CLS] / nts;
    losses[1 * L + l] = hp.lw_bbox * acc[S_BBOX] / avg;
    losses[2 * L + l] = hp.lw_dfl * acc[S_DFL] / 4.0f / avg;
    losses[3 * L + l] = hp.lw_ld * acc[S_LD] / 4.0f;
    losses[4 * L + l] = hp.lw_ld_vlr * acc[S_VLR] / 16.0f;
    losses[5 * L + l] = P_l > 0 ? hp.lw_kd * acc[S_KD] / (float)P_l : 0.0f;
    // LDHead: loss_kd_neg = 0 * KD (ld_head.py:267-271); LDATSSHead: this row
    // carries loss_centerness (ld_atss.py:136-140)
    losses[6 * L + l] = (hp.flags & LD_LOSS_ATSS) ? hp.lw_ctr * acc[S_WSUM] / nts : 0.0f;
    losses[7 * L + l] = (P_l > 0 && F_l > 0)
                            ? hp.lw_im * acc[S_IM] / ((float)F_l * (float)hp.feat_channels)
                            : 0.0f;
  }
}

// ------------------------------------ the north-star kernel, stand-alone ----
// LD KL + Integral (+ gradient) over a dense channel-major (68, rows) map.
// One thread owns R consecutive anchors of ONE side (grid.y = side): 34 input
// streams per thread, ~60 VGPRs, 8 waves/SIMD -- the kernel is pure HBM
// streaming, so occupancy (bytes in flight) is what sets the rate.  R floats
// are moved per access (float / float2 / float4); NT selects non-temporal
// (streaming) loads and stores.

template <int R, bool NT>
__device__ __forceinline__ void vload(const float* p, float* out) {
  typedef typename VecT<R>::T V;
  V v = NT ? __builtin_nontemporal_load(reinterpret_cast<const V*>(p))
           : *reinterpret_cast<const V*>(p);
  const float* f = reinterpret_cast<const float*>(&v);
#pragma unroll
  for (int j = 0; j < R; ++j) out[j] = f[j];
}
template <int R, bool NT>
__device__ __forceinline__ void vstore(float* p, const float* in) {
  typedef typename VecT<R>::T V;
  V v;
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int j = 0; j < R; ++j) f[j] = in[j];
  if (NT)
    __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
  else
    *reinterpret_cast<V*>(p) = v;
}

template <int R, bool GRAD, bool NT>
__global__ __launch_bounds__(256) void kl_integral_dense_kernel(
    const float* __restrict__ s_reg, const float* __restrict__ t_reg,
    const float* __restrict__ weight, int64_t rows, float T, float scale,
    float* __restrict__ integral, float* __restrict__ loss_rows,
    float* __restrict__ grad) {
  const int side = blockIdx.y;
  const int64_t r0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * R;
  if (r0 >= rows) return;
  const float invT = 1.0f / T;
  float w[R];
  vload<R, false>(weight + r0, w);
  float sv[R][K17], tv[R][K17];
#pragma unroll
  for (int k = 0; k < K17; ++k) {
    float a[R], b[R];
    vload<R, NT>(s_reg + (int64_t)(side * K17 + k) * rows + r0, a);
    vload<R, NT>(t_reg + (int64_t)(side * K17 + k) * rows + r0, b);
#pragma unroll
    for (int j = 0; j < R; ++j) {
      sv[j][k] = a[j];
      tv[j][k] = b[j];
    }
  }
  float e[R], l[R], gr[R][K17];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    float d[K17], p[K17];
    const float kl = ld::kl_rows<K17>(sv[j], tv[j], invT, T, d);
    e[j] = ld::softmax_expect<K17>(sv[j], p);
    l[j] = w[j] * kl;
    if (GRAD) {
      const float cg = scale * w[j] * (T / (float)K17);
#pragma unroll
      for (int k = 0; k < K17; ++k) gr[j][k] = cg * d[k];
    }
  }
  vstore<R, NT>(integral + (int64_t)side * rows + r0, e);
  vstore<R, NT>(loss_rows + (int64_t)side * rows + r0, l);
  if (GRAD) {
#pragma unroll
    for (int k = 0; k < K17; ++k) {
      float g[R];
#pragma unroll
      for (int j = 0; j < R; ++j) g[j] = gr[j][k];
      vstore<R, NT>(grad + (int64_t)(side * K17 + k) * rows + r0, g);
    }
  }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int check_geom(const ld_geom_t* g) {
  if (!g || g->num_levels < 1 || g->num_levels > LD_MAX_LEVELS ||
      g->num_imgs < 1 || g->num_anchors < 1)
    return LD_EINVAL;
  return 0;
}

int check_hp(const ld_loss_hp_t* hp) {
  if (!hp) return LD_EINVAL;
  if (hp->reg_max != 16 || hp->qfl_beta != 2.0f) return LD_EUNSUPPORTED;
  if (hp->num_classes < 1 || hp->feat_channels < 1) return LD_EINVAL;
  if (hp->T_ld < 1.0f || hp->T_ld_vlr < 1.0f || hp->T_kd < 1.0f)
    return LD_EINVAL;  // kd_loss.py:51 assert T >= 1
  return 0;
}

}  // namespace

namespace {
// workspace = [prepass partials | main partials | positives record]
struct LossWs {
  size_t pre_floats, main_floats, pos_floats;
  BlockMap bm64, bm256;
};
LossWs loss_ws(const ld_geom_t& g) {
  LossWs w;
  w.bm64 = make_block_map(g, kWave);
  w.bm256 = make_block_map(g, kBlk);
  w.pre_floats = align_up((size_t)S_COUNT * g.num_imgs * w.bm64.blocks_per_img, 64);
  w.main_floats =
      align_up((size_t)S_COUNT * kZMax * g.num_imgs * w.bm256.blocks_per_img, 64);
  w.pos_floats = (size_t)g.num_imgs * g.num_anchors * kPosRec;
  return w;
}
inline int im_chunk(int ch) { return max(32, (ch + kZMax - 1) / kZMax); }

// ---- which reg-side kernel ld_loss_main_parts launches ----------------------
// variant word: bits 0-1 log2(VEC) | bit 2 NT loads | bit 3 NT stores | bit 4
// FAST (v_exp_f32 / v_rcp_f32) | bit 5 side_fast | bit 6 "NT bits also apply
// below the Infinity-Cache size" | bit 7 VEC 1 held to 64 VGPRs (8 waves per
// SIMD) | bits 8-15 KiB of dynamic LDS per workgroup (an occupancy throttle for
// experiments) | bit 16 (with bit 5) the 4 sides of a chunk on one XCD; < 0 =
// the round-2 kernel.
// Default = the measured best of profiles/r03_ldkl_variants.json.  FAST replaces
// the correctly rounded expf / division of the KL rows by v_exp_f32 / v_rcp_f32
// (1 ulp each): the loss tables stay within 1e-6 of the reference's (bar 1e-4,
// tests/test_gpu_lossblock.py) and the gradient rows within 2.6e-7 of the scale;
// ld_loss_set_reg_variant(LD_REG_VARIANT_DEFAULT & ~16) restores the exact forms.
// vec 1, NT loads + stores, hardware exp, side-fast: 568-571 us at 2^24 rows =
// 0.76 of 8 TB/s (round-2 kernel: 665-743 us), 11.2 us at the C2 step size
#define LD_REG_VARIANT_DEFAULT (0 | 4 | 8 | 16 | 32)
int g_reg_variant = LD_REG_VARIANT_DEFAULT;

template <int VEC, bool NTL, bool NTS>
void launch_reg_lean_t(bool fast, bool w8, dim3 grid, size_t lds, hipStream_t stream,
                       const ld_geom_t& geom, const ld_loss_hp_t& hp, const BlockMap& bm,
                       const BlockMap& bmv, int side_fast, const ld_maps_t& reg,
                       const ld_maps_t& t_reg, const int64_t* labels,
                       const float* bbox_targets, const float* vlr,
                       const float* weight_targets, const float* norm,
                       const float* upstream, const float* posrec,
                       const ld_maps_t& grad_reg, float* partial) {
#define LD_LEAN_GO(F, W)                                                                  \
  LD_LAUNCH((loss_reg_lean_kernel<VEC, NTL, NTS, F, W>), grid, dim3(kBlk), lds,  \
                     stream, geom, hp, bm, bmv, side_fast, reg, t_reg, labels,            \
                     bbox_targets, vlr, weight_targets, norm, upstream, posrec, grad_reg, \
                     partial)
  // WPE = minimum waves per SIMD the register allocator is held to.  1 = its own
  // choice: VEC 1 -> 79 VGPRs (6 waves) with FAST, 86 (5) without; VEC 2 -> 109 /
  // 125 (4); VEC 4 -> 173 / 177 (2).  8 (VEC 1 only, on request) -> 64 VGPRs with
  // 12 dwords of spill, all but 2 of them in the rare positives path.
  if (VEC == 1 && w8) {
    if (fast) LD_LEAN_GO(true, (VEC == 1 ? 8 : 1));
    else LD_LEAN_GO(false, (VEC == 1 ? 8 : 1));
  } else {
    if (fast) LD_LEAN_GO(true, 1);
    else LD_LEAN_GO(false, 1);
  }
#undef LD_LEAN_GO
}

void launch_reg_lean(int variant, bool big, const ld_geom_t& geom, const ld_loss_hp_t& hp,
                     const BlockMap& bm, unsigned by, const ld_maps_t& reg,
                     const ld_maps_t& t_reg, const int64_t* labels,
                     const float* bbox_targets, const float* vlr,
                     const float* weight_targets, const float* norm, const float* upstream,
                     const float* posrec, const ld_maps_t& grad_reg, float* partial,
                     hipStream_t stream) {
  const int vec = 1 << (variant & 3);
  // at train-step sizes the gradient is re-read by the next kernel and the
  // logits were just written: keep them cached unless bit 6 says otherwise
  const bool nt_on = big || (variant & 64);
  const bool ntl = nt_on && (variant & 4), nts = nt_on && (variant & 8);
  const bool fast = (variant & 16) != 0, w8 = (variant & 128) != 0;
  const int side_fast = (variant & 32) ? ((variant & 0x10000) ? 2 : 1) : 0;
  const size_t lds = (size_t)((variant >> 8) & 0xff) << 10;
  const BlockMap bmv = make_block_map(geom, kBlk * vec);
  const dim3 grid = side_fast ? dim3(bmv.blocks_per_img * 4, by, 1)
                              : dim3(bmv.blocks_per_img, by, 4);
#define LD_LEAN(V, L, S)                                                               \
  launch_reg_lean_t<V, L, S>(fast, w8, grid, lds, stream, geom, hp, bm, bmv, side_fast, reg, \
                             t_reg, labels, bbox_targets, vlr, weight_targets, norm,    \
                             upstream, posrec, grad_reg, partial)
#define LD_LEAN_V(V)                          \
  do {                                        \
    if (ntl && nts) LD_LEAN(V, true, true);   \
    else if (ntl) LD_LEAN(V, true, false);    \
    else if (nts) LD_LEAN(V, false, true);    \
    else LD_LEAN(V, false, false);            \
  } while (0)
  if (vec == 4) LD_LEAN_V(4);
  else if (vec == 2) LD_LEAN_V(2);
  else LD_LEAN_V(1);
#undef LD_LEAN_V
#undef LD_LEAN
}
}  // namespace

extern "C" int ld_loss_set_reg_variant(int variant) {
  const int prev = g_reg_variant;
  if (variant >= 0 && (variant & 3) == 3) return LD_EINVAL;
  g_reg_variant = variant;
  return prev < 0 ? -1 : prev;
}

extern "C" size_t ld_loss_workspace_bytes(const ld_geom_t* geom) {
  if (check_geom(geom) != 0) return 0;
  const LossWs w = loss_ws(*geom);
  return align_up((w.pre_floats + w.main_floats + w.pos_floats) * sizeof(float), 256);
}

extern "C" int ld_loss_prepass(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                               const ld_maps_t* cls, const ld_maps_t* reg,
                               const int64_t* labels, const float* bbox_targets,
                               const int32_t* counts, float* weight_targets,
                               float* score, float* norm, void* workspace,
                               size_t workspace_bytes, ld_stream_t stream) {
  return ld_loss_prepass_ex(geom, hp, cls, reg, labels, bbox_targets, nullptr, counts,
                            weight_targets, score, norm, workspace, workspace_bytes,
                            stream);
}

extern "C" int ld_loss_prepass_ex(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                                  const ld_maps_t* cls, const ld_maps_t* reg,
                                  const int64_t* labels, const float* bbox_targets,
                                  const float* vlr, const int32_t* counts,
                                  float* weight_targets, float* score, float* norm,
                                  void* workspace, size_t workspace_bytes,
                                  ld_stream_t stream_) {
  if (int e = check_geom(geom)) return e;
  if (int e = check_hp(hp)) return e;
  if ((hp->flags & LD_LOSS_FCOS) && (!vlr || !(hp->flags & LD_LOSS_ATSS)))
    return LD_EINVAL;
  if (!cls || !reg || !labels || !bbox_targets || !counts || !weight_targets ||
      !score || !norm)
    return LD_EINVAL;
  if (!workspace || workspace_bytes < ld_loss_workspace_bytes(geom))
    return LD_ENOSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const BlockMap bm = make_block_map(*geom, kWave);
  float* partial = (float*)workspace;
  dim3 grid(bm.blocks_per_img, geom->num_imgs);
  LD_LAUNCH(loss_prepass_kernel, grid, dim3(kWave), 0, stream, *geom,
                     *hp, bm, *cls, *reg, labels, bbox_targets, vlr, weight_targets,
                     score, partial);
  const int nparts = geom->num_imgs * bm.blocks_per_img;
  LD_LAUNCH(loss_norm_kernel, dim3(1), dim3(256), 0, stream, nparts,
                     partial + (size_t)S_WSUM * nparts, counts,
                     geom->num_imgs + 2 * geom->num_levels, norm);
  return (int)hipGetLastError();
}

extern "C" int ld_loss_main_parts(
    const ld_geom_t* geom, const ld_loss_hp_t* hp, const ld_maps_t* cls,
    const ld_maps_t* reg, const ld_maps_t* t_cls, const ld_maps_t* t_reg,
    const ld_maps_t* x, const ld_maps_t* t_x, const int64_t* labels,
    const float* label_weights, const float* bbox_targets, const float* vlr,
    const float* im, const int32_t* counts, const float* weight_targets,
    const float* score, const float* norm, const float* upstream,
    const ld_maps_t* grad_cls, const ld_maps_t* grad_reg, const ld_maps_t* grad_x,
    const ld_maps_t* kd_s, const ld_maps_t* kd_t, const ld_maps_t* grad_kd,
    void* workspace, size_t workspace_bytes, int parts, ld_stream_t stream_) {
  if (int e = check_geom(geom)) return e;
  if (int e = check_hp(hp)) return e;
  if (!cls || !reg || !t_cls || !t_reg || !x || !t_x || !labels ||
      !label_weights || !bbox_targets || !vlr || !im || !counts ||
      !weight_targets || !score || !norm || !grad_cls || !grad_reg || !grad_x)
    return LD_EINVAL;
  if (!workspace || workspace_bytes < ld_loss_workspace_bytes(geom))
    return LD_ENOSPACE;
  const int CC = hp->cls_channels > 0 ? hp->cls_channels : hp->num_classes;
  if ((CC + kClsChunk - 1) / kClsChunk > kZMax) return LD_EUNSUPPORTED;
  const bool split = kd_s != nullptr;
  if (split != (kd_t != nullptr) || split != (grad_kd != nullptr)) return LD_EINVAL;
  if (((hp->flags & LD_LOSS_PROB_CLS) != 0) != split)
    return LD_EINVAL;  // probabilities cannot carry the KD logits, and vice versa
  if ((hp->flags & LD_LOSS_ATSS) && (hp->flags & LD_LOSS_PROB_CLS)) return LD_EINVAL;
  if ((hp->flags & LD_LOSS_RETINA) &&
      ((hp->flags & (LD_LOSS_ATSS | LD_LOSS_FCOS | LD_LOSS_PROB_CLS)) ||
       hp->T_ld != hp->T_ld_vlr))
    return LD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  const LossWs w = loss_ws(*geom);
  const BlockMap& bm = w.bm256;
  float* partial = (float*)workspace + w.pre_floats;
  float* posrec = partial + w.main_floats;
  const unsigned bx = bm.blocks_per_img, by = geom->num_imgs;
  if (parts & LD_LOSS_PART_POS)
    LD_LAUNCH(loss_pos_kernel, dim3(bx, by), dim3(kBlk), 0, stream, *geom, *hp,
                       bm, split ? *kd_s : *cls, split ? *kd_t : *t_cls, *reg, labels,
                       bbox_targets, weight_targets, score, norm, upstream, posrec,
                       partial);
  // the lean kernel addresses a level's 68 channel planes through ONE 32-bit
  // buffer descriptor extent (stride_c * 68 * 4 bytes): a level map that large
  // (>= 15.8 M positions) takes the round-2 kernel instead of wrapping
  bool extent_ok = true;
  for (int l = 0; l < geom->num_levels; ++l)
    extent_ok &= reg->stride_c[l] * 272 < ((int64_t)1 << 32) &&
                 t_reg->stride_c[l] * 272 < ((int64_t)1 << 32) &&
                 grad_reg->stride_c[l] * 272 < ((int64_t)1 << 32);
  const bool lean_ok = !(hp->flags & LD_LOSS_RETINA) && hp->T_ld == hp->T_ld_vlr &&
                       g_reg_variant >= 0 && extent_ok;
  if ((parts & LD_LOSS_PART_REG) && lean_ok) {
    const size_t bytes = (size_t)geom->num_imgs * geom->num_anchors * 68 * 4 * 3;
    const bool big = bytes > ((size_t)192 << 20);
    launch_reg_lean(g_reg_variant, big, *geom, *hp, bm, by, *reg, *t_reg, labels,
                    bbox_targets, vlr, weight_targets, norm, upstream, posrec, *grad_reg,
                    partial, stream);
  } else if (parts & LD_LOSS_PART_REG) {
    // streaming (non-temporal) access once the three 68-channel maps exceed what
    // the 256 MiB Infinity Cache can hold; at train-step sizes the gradient is
    // re-read by the next kernel and should stay cached
    const size_t bytes = (size_t)geom->num_imgs * geom->num_anchors * 68 * 4 * 3;
    if (bytes > ((size_t)192 << 20))
      LD_LAUNCH(loss_reg_dense_kernel<true>, dim3(bx, by, 4), dim3(kBlk), 0,
                         stream, *geom, *hp, bm, *reg, *t_reg, labels, bbox_targets, vlr,
                         weight_targets, norm, upstream, posrec, *grad_reg, partial);
    else
      LD_LAUNCH(loss_reg_dense_kernel<false>, dim3(bx, by, 4), dim3(kBlk), 0,
                         stream, *geom, *hp, bm, *reg, *t_reg, labels, bbox_targets, vlr,
                         weight_targets, norm, upstream, posrec, *grad_reg, partial);
  }
  if (parts & LD_LOSS_PART_CLS) {
    const unsigned zc = (CC + kClsChunk - 1) / kClsChunk;
    if (split)
      LD_LAUNCH(loss_cls_dense_kernel<true>, dim3(bx, by, zc), dim3(kBlk), 0,
                         stream, *geom, *hp, bm, *cls, *kd_s, *kd_t, labels,
                         label_weights, score, counts, norm, upstream, posrec,
                         *grad_cls, *grad_kd, partial);
    else
      LD_LAUNCH(loss_cls_dense_kernel<false>, dim3(bx, by, zc), dim3(kBlk), 0,
                         stream, *geom, *hp, bm, *cls, *cls, *t_cls, labels,
                         label_weights, score, counts, norm, upstream, posrec,
                         *grad_cls, *grad_cls, partial);
  }
  if (parts & LD_LOSS_PART_IM) {
    const int chunk = im_chunk(hp->feat_channels);
    const unsigned zi = (hp->feat_channels + chunk - 1) / chunk;
    LD_LAUNCH(loss_im_dense_kernel, dim3(bx, by, zi), dim3(kBlk), 0, stream,
                       *geom, *hp, bm, *x, *t_x, im, counts, upstream, *grad_x, chunk,
                       partial);
  }
  return (int)hipGetLastError();
}

extern "C" int ld_loss_centerness(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                                  const ld_maps_t* ctr, const int64_t* labels,
                                  const float* score, const float* norm,
                                  const float* upstream, const ld_maps_t* grad_ctr,
                                  void* workspace, size_t workspace_bytes,
                                  ld_stream_t stream_) {
  if (int e = check_geom(geom)) return e;
  if (int e = check_hp(hp)) return e;
  if (!(hp->flags & LD_LOSS_ATSS)) return LD_EINVAL;
  if (!ctr || !labels || !score || !norm || !grad_ctr) return LD_EINVAL;
  if (!workspace || workspace_bytes < ld_loss_workspace_bytes(geom))
    return LD_ENOSPACE;
  const LossWs w = loss_ws(*geom);
  float* partial = (float*)workspace + w.pre_floats;
  LD_LAUNCH(loss_ctr_dense_kernel, dim3(w.bm256.blocks_per_img, geom->num_imgs),
                     dim3(kBlk), 0, (hipStream_t)stream_, *geom, *hp, w.bm256, *ctr,
                     labels, score, norm, upstream, *grad_ctr, partial);
  return (int)hipGetLastError();
}

extern "C" int ld_loss_main(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                            const ld_maps_t* cls, const ld_maps_t* reg,
                            const ld_maps_t* t_cls, const ld_maps_t* t_reg,
                            const ld_maps_t* x, const ld_maps_t* t_x,
                            const int64_t* labels, const float* label_weights,
                            const float* bbox_targets, const float* vlr,
                            const float* im, const int32_t* counts,
                            const float* weight_targets, const float* score,
                            const float* norm, const float* upstream,
                            const ld_maps_t* grad_cls, const ld_maps_t* grad_reg,
                            const ld_maps_t* grad_x, void* workspace,
                            size_t workspace_bytes, ld_stream_t stream) {
  return ld_loss_main_parts(geom, hp, cls, reg, t_cls, t_reg, x, t_x, labels,
                            label_weights, bbox_targets, vlr, im, counts,
                            weight_targets, score, norm, upstream, grad_cls, grad_reg,
                            grad_x, nullptr, nullptr, nullptr, workspace,
                            workspace_bytes, LD_LOSS_PART_ALL, stream);
}

extern "C" int ld_loss_finalize(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                                const int32_t* counts, const float* norm,
                                const void* workspace, float* losses,
                                ld_stream_t stream_) {
  if (int e = check_geom(geom)) return e;
  if (int e = check_hp(hp)) return e;
  if (!counts || !norm || !workspace || !losses) return LD_EINVAL;
  const LossWs w = loss_ws(*geom);
  const int chunk = im_chunk(hp->feat_channels);
  LD_LAUNCH(loss_finalize_kernel, dim3(geom->num_levels), dim3(kWave), 0,
                     (hipStream_t)stream_, *geom, *hp, w.bm256,
                     ((hp->cls_channels > 0 ? hp->cls_channels : hp->num_classes) +
                      kClsChunk - 1) / kClsChunk,
                     (hp->feat_channels + chunk - 1) / chunk, counts, norm,
                     (const float*)workspace + w.pre_floats, losses);
  return (int)hipGetLastError();
}

extern "C" size_t ld_gi_region_workspace_bytes(const ld_geom_t* geom) {
  if (check_geom(geom) != 0) return 0;
  return (size_t)geom->num_imgs * geom->num_anchors * 5 * sizeof(float);
}

extern "C" int ld_gi_region(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                            const ld_maps_t* cls, const ld_maps_t* reg,
                            const ld_maps_t* t_cls, const ld_maps_t* t_reg, int topn,
                            float iou_thr, float* im, int32_t* counts,
                            void* workspace, size_t workspace_bytes,
                            ld_stream_t stream_) {
  if (int e = check_geom(geom)) return e;
  if (int e = check_hp(hp)) return e;
  if (!cls || !reg || !t_cls || !t_reg || !im || !counts || topn < 1 ||
      !(iou_thr >= 0.0f))
    return LD_EINVAL;
  if (!workspace || workspace_bytes < ld_gi_region_workspace_bytes(geom))
    return LD_ENOSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const BlockMap bm = make_block_map(*geom, kBlk);
  float* giscore = (float*)workspace;
  float* gibox = giscore + (size_t)geom->num_imgs * geom->num_anchors;
  LD_LAUNCH(gi_score_kernel, dim3(bm.blocks_per_img, geom->num_imgs),
                     dim3(kBlk), 0, stream, *geom, *hp, bm, *cls, *t_cls, *reg, *t_reg,
                     giscore, gibox);
  LD_LAUNCH(gi_select_kernel, dim3(geom->num_levels), dim3(1024), 0, stream,
                     *geom, topn, iou_thr, giscore, gibox, im, counts);
  return (int)hipGetLastError();
}

extern "C" int ld_kl_integral_dense(const float* s_reg, const float* t_reg,
                                    const float* weight, int64_t rows, float T,
                                    float scale, float* integral,
                                    float* loss_rows, float* grad,
                                    ld_stream_t stream_) {
  if (!s_reg || !t_reg || !weight || !integral || !loss_rows || rows < 1 ||
      T < 1.0f)
    return LD_EINVAL;
  hipStream_t stream = (hipStream_t)stream_;
  // widest vector the row count and the base alignment allow
  auto aligned = [](const void* p, size_t a) { return ((uintptr_t)p % a) == 0; };
  int R = 1;
  if (rows % 4 == 0 && aligned(s_reg, 16) && aligned(t_reg, 16) &&
      aligned(weight, 16) && aligned(integral, 16) && aligned(loss_rows, 16) &&
      (!grad || aligned(grad, 16)))
    R = 4;
  else if (rows % 2 == 0 && aligned(s_reg, 8) && aligned(t_reg, 8) &&
           aligned(weight, 8) && aligned(integral, 8) && aligned(loss_rows, 8) &&
           (!grad || aligned(grad, 8)))
    R = 2;
  // LD_KL_VEC / LD_KL_NT override the vector width (1/2/4) and the
  // non-temporal access mode for benchmarking
  bool nt = true;
  if (const char* env = getenv("LD_KL_VEC")) {
    int v = atoi(env);
    if ((v == 1 || v == 2 || v == 4) && v <= R) R = v;
  } else {
    R = 1;  // measured best on MI355X (profiles/r01_kernels_s6.json): one float
            // per lane per stream + non-temporal access, 8 waves/SIMD
  }
  if (const char* env = getenv("LD_KL_NT")) nt = atoi(env) != 0;
  const int64_t threads = (rows + R - 1) / R;
  const dim3 block(256), grid((unsigned)((threads + 255) / 256), 4);
#define LD_LAUNCH_KL(RR, GG, NN)                                                  \
  LD_LAUNCH((kl_integral_dense_kernel<RR, GG, NN>), grid, block, 0,      \
                     stream, s_reg, t_reg, weight, rows, T, scale, integral,      \
                     loss_rows, grad)
#define LD_LAUNCH_KL_R(RR)                                                        \
  do {                                                                            \
    if (grad) {                                                                   \
      if (nt) LD_LAUNCH_KL(RR, true, true);                                       \
      else LD_LAUNCH_KL(RR, true, false);                                         \
    } else {                                                                      \
      if (nt) LD_LAUNCH_KL(RR, false, true);                                      \
      else LD_LAUNCH_KL(RR, false, false);                                        \
    }                                                                             \
  } while (0)
  if (R == 4) LD_LAUNCH_KL_R(4);
  else if (R == 2) LD_LAUNCH_KL_R(2);
  else LD_LAUNCH_KL_R(1);
#undef LD_LAUNCH_KL_R
#undef LD_LAUNCH_KL
  return (int)hipGetLastError();
}
