// GFLv2's distribution-guided quality estimator for gfx950, forward and
// backward: the tail of GFocalHead.forward_single
// (mmdet/models/dense_heads/gfocal_head.py:201-217) with its reg_conf MLP
// (:140-144) -- config 5 / SURVEY.md row R-V2.
//
// Reference op sequence per level (~14 ATen launches forward, more backward):
//   softmax(dim=2) -> topk(4) -> mean -> cat -> reshape -> Conv2d(20,64,1) ->
//   ReLU -> Conv2d(64,1,1) -> Sigmoid ; cls_feat.sigmoid() * quality
// Here: ONE launch for all levels and images each way, on the level-concatenated
// (N, C, P) tensors.  Thread <-> anchor; the channel stride is P so a wavefront
// reads 64 consecutive floats of every channel (coalesced, HBM-bound: 68 + C in,
// C (+ 1) out per anchor); the 1 409 MLP parameters are wave-uniform scalar
// loads.  Parameter gradients are reduced per 256-anchor block through LDS
// (two small GEMMs: gh^T stat and gz2 h) and summed over blocks in a fixed
// order by a second kernel: deterministic, no float atomics.
#include <hip/hip_runtime.h>

#include "ld_launch.h"

#include "../../include/ld_hip.h"

namespace {

constexpr int K17 = 17;      // reg_max + 1
constexpr int TOPK = 4;      // reg_topk
constexpr int NSTAT = 20;    // 4 sides x (top-4 + mean)
constexpr int HID = 64;      // reg_channels
constexpr int kBlk = 256;
constexpr int NPAR = HID * NSTAT + HID + HID + 1;  // w1, b1, w2, b2 = 1409

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// softmax of one 17-bin side, its top-4 (values descending, lower index first
// on ties like a stable sort) and their mean
__device__ __forceinline__ void side_stat(const float* __restrict__ reg, size_t base,
                                          size_t cstride, int side, float* p /*[17]*/,
                                          int* idx /*[4]*/, float* stat /*[5]*/) {
  float m = -INFINITY;
#pragma unroll
  for (int k = 0; k < K17; ++k) {
    p[k] = reg[base + (size_t)(side * K17 + k) * cstride];
    m = fmaxf(m, p[k]);
  }
  float z = 0.0f;
#pragma unroll
  for (int k = 0; k < K17; ++k) {
    p[k] = expf(p[k] - m);
    z += p[k];
  }
  const float rz = 1.0f / z;
#pragma unroll
  for (int k = 0; k < K17; ++k) p[k] *= rz;
  unsigned taken = 0;
  float sum = 0.0f;
#pragma unroll
  for (int t = 0; t < TOPK; ++t) {
    float best = -1.0f;
    int bi = 0;
#pragma unroll
    for (int k = 0; k < K17; ++k) {
      const bool free_ = !((taken >> k) & 1u);
      if (free_ && p[k] > best) {
        best = p[k];
        bi = k;
      }
    }
    taken |= 1u << bi;
    idx[t] = bi;
    stat[t] = best;
    sum += best;
  }
  stat[TOPK] = sum / (float)TOPK;
}

__global__ __launch_bounds__(kBlk) void quality_fwd_kernel(
    const float* __restrict__ reg, const float* __restrict__ cls_feat, int C, int P,
    const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2,
    float* __restrict__ cls_score, float* __restrict__ quality) {
  const int n = blockIdx.y;
  const int p = blockIdx.x * kBlk + threadIdx.x;
  if (p >= P) return;
  const size_t rbase = (size_t)n * 4 * K17 * P + p;
  float stat[NSTAT];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    float pr[K17];
    int idx[TOPK];
    side_stat(reg, rbase, (size_t)P, s, pr, idx, stat + s * 5);
  }
  float z2 = b2[0];
  for (int j = 0; j < HID; ++j) {
    float z1 = b1[j];
#pragma unroll
    for (int i = 0; i < NSTAT; ++i) z1 += w1[j * NSTAT + i] * stat[i];
    z2 += w2[j] * fmaxf(z1, 0.0f);
  }
  const float q = sigmoidf_(z2);
  quality[(size_t)n * P + p] = q;
  const size_t cbase = (size_t)n * C * P + p;
  for (int c = 0; c < C; ++c)
    cls_score[cbase + (size_t)c * P] = sigmoidf_(cls_feat[cbase + (size_t)c * P]) * q;
}

// Backward.  Per block: partial parameter gradients -> part[block][NPAR].
__global__ __launch_bounds__(kBlk) void quality_bwd_kernel(
    const float* __restrict__ reg, const float* __restrict__ cls_feat,
    const float* __restrict__ quality, const float* __restrict__ g_cls_score, int C,
    int P, const float* __restrict__ w1, const float* __restrict__ b1,
    const float* __restrict__ w2, float* __restrict__ g_cls_feat,
    float* __restrict__ g_reg, float* __restrict__ part) {
  // [HID][a] hidden-side vector, [NSTAT][a] stat, [a] gz2 (anchor fastest, rows
  // padded by one float: lanes write consecutive addresses, readers of different
  // rows hit different banks)
  constexpr int SH = kBlk + 1;
  __shared__ float s_h[HID * SH];
  __shared__ float s_stat[NSTAT * SH];
  __shared__ float s_gz2[kBlk];
  const int n = blockIdx.y;
  const int t = threadIdx.x;
  const int p = blockIdx.x * kBlk + t;
  const bool active = p < P;
  float stat[NSTAT];
  float gz2 = 0.0f;
  float hval[HID];  // relu(z1)
#pragma unroll
  for (int i = 0; i < NSTAT; ++i) stat[i] = 0.0f;
#pragma unroll
  for (int j = 0; j < HID; ++j) hval[j] = 0.0f;
  float gstat[NSTAT];
#pragma unroll
  for (int i = 0; i < NSTAT; ++i) gstat[i] = 0.0f;
  if (active) {
    const size_t rbase = (size_t)n * 4 * K17 * P + p;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float pr[K17];
      int idx[TOPK];
      side_stat(reg, rbase, (size_t)P, s, pr, idx, stat + s * 5);
    }
    // d total / d quality = sum_c g_cls_score[c] * sigmoid(cls_feat[c]);
    // d total / d cls_feat[c] = g_cls_score[c] * q * s (1 - s)
    const float q = quality[(size_t)n * P + p];
    const size_t cbase = (size_t)n * C * P + p;
    float gq = 0.0f;
    for (int c = 0; c < C; ++c) {
      const float g = g_cls_score[cbase + (size_t)c * P];
      const float sg = sigmoidf_(cls_feat[cbase + (size_t)c * P]);
      gq += g * sg;
      g_cls_feat[cbase + (size_t)c * P] = g * q * sg * (1.0f - sg);
    }
    gz2 = gq * q * (1.0f - q);
    for (int j = 0; j < HID; ++j) {
      float z1 = b1[j];
#pragma unroll
      for (int i = 0; i < NSTAT; ++i) z1 += w1[j * NSTAT + i] * stat[i];
      hval[j] = fmaxf(z1, 0.0f);
      const float gh = z1 > 0.0f ? gz2 * w2[j] : 0.0f;
#pragma unroll
      for (int i = 0; i < NSTAT; ++i) gstat[i] += gh * w1[j * NSTAT + i];
    }
    // stat -> top-k probabilities -> softmax -> reg logits
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float pr[K17], st5[5];
      int idx[TOPK];
      side_stat(reg, rbase, (size_t)P, s, pr, idx, st5);
      float gp[K17];
#pragma unroll
      for (int k = 0; k < K17; ++k) gp[k] = 0.0f;
      const float gmean = gstat[s * 5 + TOPK] / (float)TOPK;
#pragma unroll
      for (int tt = 0; tt < TOPK; ++tt) {
        const float g = gstat[s * 5 + tt] + gmean;
#pragma unroll
        for (int k = 0; k < K17; ++k) gp[k] += (idx[tt] == k) ? g : 0.0f;
      }
      float dot = 0.0f;
#pragma unroll
      for (int k = 0; k < K17; ++k) dot += gp[k] * pr[k];
#pragma unroll
      for (int k = 0; k < K17; ++k)
        g_reg[rbase + (size_t)(s * K17 + k) * P] = pr[k] * (gp[k] - dot);
    }
  }
  // ---- parameter gradients of this block, two passes through LDS -----------
  float* out = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NPAR;
  // pass 1: gh[a][j] and stat[a][i] -> g_w1[j][i] = sum_a gh stat, g_b1[j] = sum_a gh
#pragma unroll
  for (int j = 0; j < HID; ++j)
    s_h[j * SH + t] = (active && hval[j] > 0.0f) ? gz2 * w2[j] : 0.0f;
#pragma unroll
  for (int i = 0; i < NSTAT; ++i) s_stat[i * SH + t] = stat[i];
  s_gz2[t] = gz2;
  __syncthreads();
  for (int o = t; o < HID * NSTAT + HID; o += kBlk) {
    float acc = 0.0f;
    if (o < HID * NSTAT) {
      const int j = o / NSTAT, i = o - j * NSTAT;
      for (int a = 0; a < kBlk; ++a) acc += s_h[j * SH + a] * s_stat[i * SH + a];
    } else {
      const int j = o - HID * NSTAT;
      for (int a = 0; a < kBlk; ++a) acc += s_h[j * SH + a];
    }
    out[o] = acc;
  }
  __syncthreads();
  // pass 2: h[a][j] -> g_w2[j] = sum_a gz2 h, g_b2 = sum_a gz2
#pragma unroll
  for (int j = 0; j < HID; ++j) s_h[j * SH + t] = hval[j];
  __syncthreads();
  if (t < HID) {
    float acc = 0.0f;
    for (int a = 0; a < kBlk; ++a) acc += s_gz2[a] * s_h[t * SH + a];
    out[HID * NSTAT + HID + t] = acc;
  } else if (t == HID) {
    float acc = 0.0f;
    for (int a = 0; a < kBlk; ++a) acc += s_gz2[a];
    out[NPAR - 1] = acc;
  }
}

__global__ __launch_bounds__(256) void quality_param_reduce_kernel(
    const float* __restrict__ part, int nblocks, float* __restrict__ g_w1,
    float* __restrict__ g_b1, float* __restrict__ g_w2, float* __restrict__ g_b2,
    int accumulate) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= NPAR) return;
  float acc = 0.0f;
  for (int b = 0; b < nblocks; ++b) acc += part[(size_t)b * NPAR + o];
  float* dst;
  if (o < HID * NSTAT)
    dst = g_w1 + o;
  else if (o < HID * NSTAT + HID)
    dst = g_b1 + (o - HID * NSTAT);
  else if (o < NPAR - 1)
    dst = g_w2 + (o - HID * NSTAT - HID);
  else
    dst = g_b2;
  *dst = accumulate ? *dst + acc : acc;
}

}  // namespace

extern "C" int ld_quality_forward(const float* reg, const float* cls_feat, int N, int C,
                                  int P, const float* w1, const float* b1,
                                  const float* w2, const float* b2, float* cls_score,
                                  float* quality, ld_stream_t stream) {
  if (!reg || !cls_feat || !w1 || !b1 || !w2 || !b2 || !cls_score || !quality ||
      N < 1 || C < 1 || P < 1)
    return LD_EINVAL;
  LD_LAUNCH(quality_fwd_kernel, dim3((P + kBlk - 1) / kBlk, N), dim3(kBlk), 0,
                     (hipStream_t)stream, reg, cls_feat, C, P, w1, b1, w2, b2,
                     cls_score, quality);
  return (int)hipGetLastError();
}

extern "C" size_t ld_quality_backward_workspace_bytes(int N, int P) {
  if (N < 1 || P < 1) return 0;
  return (size_t)N * ((P + kBlk - 1) / kBlk) * NPAR * sizeof(float);
}

extern "C" int ld_quality_backward(const float* reg, const float* cls_feat,
                                   const float* quality, const float* g_cls_score,
                                   int N, int C, int P, const float* w1,
                                   const float* b1, const float* w2, const float* b2,
                                   float* g_cls_feat, float* g_reg, float* g_w1,
                                   float* g_b1, float* g_w2, float* g_b2,
                                   int accumulate, void* workspace,
                                   size_t workspace_bytes, ld_stream_t stream_) {
  (void)b2;
  if (!reg || !cls_feat || !quality || !g_cls_score || !w1 || !b1 || !w2 ||
      !g_cls_feat || !g_reg || !g_w1 || !g_b1 || !g_w2 || !g_b2 || N < 1 || C < 1 ||
      P < 1)
    return LD_EINVAL;
  if (!workspace || workspace_bytes < ld_quality_backward_workspace_bytes(N, P))
    return LD_ENOSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int bx = (P + kBlk - 1) / kBlk;
  LD_LAUNCH(quality_bwd_kernel, dim3(bx, N), dim3(kBlk), 0, stream, reg,
                     cls_feat, quality, g_cls_score, C, P, w1, b1, w2, g_cls_feat, g_reg,
                     (float*)workspace);
  LD_LAUNCH(quality_param_reduce_kernel, dim3((NPAR + 255) / 256), dim3(256), 0,
                     stream, (const float*)workspace, bx * N, g_w1, g_b1, g_w2, g_b2,
                     accumulate);
  return (int)hipGetLastError();
}
