// Reference-layout loss operators: row-major (rows, K) tensors exactly as the
// mmdet loss modules receive them.  They back the registry modules
// (QualityFocalLoss, DistributionFocalLoss, GIoULoss,
// KnowledgeDistillationKLDivLoss, Integral, BboxOverlaps2D) when those are
// used on their own; the train step itself runs the fused NCHW-direct kernels
// of loss.hip.  One thread per row; rows are short (4..80 floats) so a
// wavefront touches 64 consecutive rows = one contiguous span.
#include <hip/hip_runtime.h>

#include "ld_launch.h"

#include "../../include/ld_hip.h"
#include "ld_math.h"

namespace {

using ld::Box;
constexpr int K17 = ld::kRegBins;

// knowledge_distillation_kl_div_loss, kd_loss.py:10-36 (generic K, two passes)
__global__ void kd_kl_rows_kernel(const float* __restrict__ pred,
                                  const float* __restrict__ soft,
                                  const float* __restrict__ weight, int64_t rows,
                                  int K, float T, float gscale,
                                  float* __restrict__ loss_rows,
                                  float* __restrict__ grad) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float* s = pred + r * K;
  const float* t = soft + r * K;
  const float invT = 1.0f / T;
  float ms = s[0], mt = t[0];
  for (int k = 1; k < K; ++k) {
    ms = fmaxf(ms, s[k]);
    mt = fmaxf(mt, t[k]);
  }
  float zs = 0.0f, zt = 0.0f;
  for (int k = 0; k < K; ++k) {
    zs += expf((s[k] - ms) * invT);
    zt += expf((t[k] - mt) * invT);
  }
  const float rzs = 1.0f / zs, rzt = 1.0f / zt, lzs = logf(zs), lzt = logf(zt);
  const float w = weight ? weight[r] : 1.0f;
  float acc = 0.0f;
  for (int k = 0; k < K; ++k) {
    const float ps = expf((s[k] - ms) * invT) * rzs;
    const float pt = expf((t[k] - mt) * invT) * rzt;
    acc += pt * (((t[k] - mt) * invT - lzt) - ((s[k] - ms) * invT - lzs));
    if (grad) grad[r * K + k] = gscale * w * (T / (float)K) * (ps - pt);
  }
  loss_rows[r] = w * acc * (T * T) / (float)K;
}

// quality_focal_loss, gfocal_loss.py:8-50
__global__ void qfl_rows_kernel(const float* __restrict__ pred,
                                const int64_t* __restrict__ label,
                                const float* __restrict__ score,
                                const float* __restrict__ weight, int64_t rows,
                                int C, float gscale, float* __restrict__ loss_rows,
                                float* __restrict__ grad) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int64_t lab = label[r];
  const bool pos = lab >= 0 && lab < C;
  const float sc = score[r];
  const float w = weight ? weight[r] : 1.0f;
  float sum = 0.0f;
  for (int c = 0; c < C; ++c) {
    float dq;
    const float x = pred[r * C + c];
    const float q = (pos && c == (int)lab) ? ld::qfl_pos(x, sc, &dq)
                                           : ld::qfl_neg(x, &dq);
    sum += q;
    if (grad) grad[r * C + c] = gscale * w * dq;
  }
  loss_rows[r] = w * sum;
}

// distribution_focal_loss, gfocal_loss.py:53-74 (K = 17)
__global__ void dfl_rows_kernel(const float* __restrict__ pred,
                                const float* __restrict__ target,
                                const float* __restrict__ weight, int64_t rows,
                                float gscale, float* __restrict__ loss_rows,
                                float* __restrict__ grad) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float s[K17], p[K17];
#pragma unroll
  for (int k = 0; k < K17; ++k) s[k] = pred[r * K17 + k];
  ld::softmax_expect<K17>(s, p);
  float wl, wr;
  int yl;
  const float l = ld::dfl_side<K17>(s, p, target[r], &wl, &wr, &yl);
  const float w = weight ? weight[r] : 1.0f;
  loss_rows[r] = w * l;
  if (grad) {
#pragma unroll
    for (int k = 0; k < K17; ++k)
      grad[r * K17 + k] =
          gscale * w * (p[k] - (k == yl ? wl : 0.0f) - (k == yl + 1 ? wr : 0.0f));
  }
}

// giou_loss, iou_loss.py:85-102
__global__ void giou_rows_kernel(const float* __restrict__ pred,
                                 const float* __restrict__ target,
                                 const float* __restrict__ weight, int64_t rows,
                                 float eps, float gscale,
                                 float* __restrict__ loss_rows,
                                 float* __restrict__ grad) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float4 p = reinterpret_cast<const float4*>(pred)[r];
  const float4 t = reinterpret_cast<const float4*>(target)[r];
  float iou, g[4];
  const float l = ld::giou_loss_grad(Box{p.x, p.y, p.z, p.w},
                                     Box{t.x, t.y, t.z, t.w}, eps, &iou, g);
  const float w = weight ? weight[r] : 1.0f;
  loss_rows[r] = w * l;
  if (grad)
    reinterpret_cast<float4*>(grad)[r] = make_float4(
        gscale * w * g[0], gscale * w * g[1], gscale * w * g[2], gscale * w * g[3]);
}

// Integral.forward, gfl_head.py:32-44
__global__ void integral_rows_kernel(const float* __restrict__ x, int64_t rows,
                                     float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * 4) return;  // one thread per (row, side)
  float s[K17], p[K17];
#pragma unroll
  for (int k = 0; k < K17; ++k) s[k] = x[i * K17 + k];
  out[i] = ld::softmax_expect<K17>(s, p);
}

__global__ void integral_rows_bwd_kernel(const float* __restrict__ x,
                                         const float* __restrict__ grad_out,
                                         int64_t rows, float* __restrict__ grad_x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * 4) return;
  float s[K17], p[K17];
#pragma unroll
  for (int k = 0; k < K17; ++k) s[k] = x[i * K17 + k];
  const float e = ld::softmax_expect<K17>(s, p);
  const float g = grad_out[i];
#pragma unroll
  for (int k = 0; k < K17; ++k) grad_x[i * K17 + k] = g * p[k] * ((float)k - e);
}

// bbox_overlaps, iou2d_calculator.py:43-188
__device__ __forceinline__ float overlaps_mode(const Box& a, const Box& b, int mode,
                                               float eps) {
  const float area1 = (a.x2 - a.x1) * (a.y2 - a.y1);
  const float area2 = (b.x2 - b.x1) * (b.y2 - b.y1);
  const float w = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.0f);
  const float h = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.0f);
  const float overlap = w * h;
  float uni = (mode == 0 || mode == 2) ? (area1 + area2 - overlap) : area1;
  uni = fmaxf(uni, eps);
  const float ious = overlap / uni;
  if (mode == 0 || mode == 1) return ious;
  const float ew = fmaxf(fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1), 0.0f);
  const float eh = fmaxf(fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1), 0.0f);
  if (mode == 2) {
    float earea = ew * eh;
    earea = fmaxf(earea, eps);
    return ious - (earea - uni) / earea;
  }
  const float l = (b.x1 + b.x2) - (a.x1 + a.x2);
  const float r = (b.y1 + b.y2) - (a.y1 + a.y2);
  const float rho2 = (l * l) / 4.0f + (r * r) / 4.0f;
  float ec = ew * ew + eh * eh;
  ec = fmaxf(ec, eps);
  return ious - rho2 / ec;
}

__global__ void bbox_overlaps_kernel(const float* __restrict__ b1,
                                     const float* __restrict__ b2, int64_t m,
                                     int64_t n, int mode, int aligned, float eps,
                                     float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = aligned ? m : m * n;
  if (i >= total) return;
  const int64_t ia = aligned ? i : i / n, ib = aligned ? i : i % n;
  const float4 p = reinterpret_cast<const float4*>(b1)[ia];
  const float4 q = reinterpret_cast<const float4*>(b2)[ib];
  out[i] = overlaps_mode(Box{p.x, p.y, p.z, p.w}, Box{q.x, q.y, q.z, q.w}, mode, eps);
}

// deterministic two-stage sum
__global__ __launch_bounds__(256) void sum_stage1(const float* __restrict__ x,
                                                  int64_t n, float* __restrict__ part) {
  __shared__ float s[256];
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * 256)
    acc += x[i];
  s[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w >= 1; w >>= 1) {
    if (threadIdx.x < w) s[threadIdx.x] += s[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = s[0];
}

constexpr int kSumBlocks = 256;

dim3 grid_for(int64_t n, int block) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

#define LD_STREAM ((hipStream_t)stream)

extern "C" int ld_kd_kl_rows(const float* pred, const float* soft,
                             const float* weight, int64_t rows, int K, float T,
                             float gscale, float* loss_rows, float* grad,
                             ld_stream_t stream) {
  if (!pred || !soft || !loss_rows || rows < 0 || K < 1 || T < 1.0f)
    return LD_EINVAL;
  if (rows == 0) return 0;
  LD_LAUNCH(kd_kl_rows_kernel, grid_for(rows, 128), dim3(128), 0,
                     LD_STREAM, pred, soft, weight, rows, K, T, gscale, loss_rows,
                     grad);
  return (int)hipGetLastError();
}

extern "C" int ld_qfl_rows(const float* pred, const int64_t* label,
                           const float* score, const float* weight, int64_t rows,
                           int C, float gscale, float* loss_rows, float* grad,
                           ld_stream_t stream) {
  if (!pred || !label || !score || !loss_rows || rows < 0 || C < 1)
    return LD_EINVAL;
  if (rows == 0) return 0;
  LD_LAUNCH(qfl_rows_kernel, grid_for(rows, 128), dim3(128), 0, LD_STREAM,
                     pred, label, score, weight, rows, C, gscale, loss_rows, grad);
  return (int)hipGetLastError();
}

extern "C" int ld_dfl_rows(const float* pred, const float* target,
                           const float* weight, int64_t rows, int K, float gscale,
                           float* loss_rows, float* grad, ld_stream_t stream) {
  if (!pred || !target || !loss_rows || rows < 0) return LD_EINVAL;
  if (K != K17) return LD_EUNSUPPORTED;
  if (rows == 0) return 0;
  LD_LAUNCH(dfl_rows_kernel, grid_for(rows, 128), dim3(128), 0, LD_STREAM,
                     pred, target, weight, rows, gscale, loss_rows, grad);
  return (int)hipGetLastError();
}

extern "C" int ld_giou_rows(const float* pred, const float* target,
                            const float* weight, int64_t rows, float eps,
                            float gscale, float* loss_rows, float* grad,
                            ld_stream_t stream) {
  if (!pred || !target || !loss_rows || rows < 0) return LD_EINVAL;
  if (rows == 0) return 0;
  LD_LAUNCH(giou_rows_kernel, grid_for(rows, 128), dim3(128), 0, LD_STREAM,
                     pred, target, weight, rows, eps, gscale, loss_rows, grad);
  return (int)hipGetLastError();
}

extern "C" int ld_integral_rows(const float* x, int64_t rows, float* out,
                                ld_stream_t stream) {
  if (!x || !out || rows < 0) return LD_EINVAL;
  if (rows == 0) return 0;
  LD_LAUNCH(integral_rows_kernel, grid_for(rows * 4, 128), dim3(128), 0,
                     LD_STREAM, x, rows, out);
  return (int)hipGetLastError();
}

extern "C" int ld_integral_rows_bwd(const float* x, const float* grad_out,
                                    int64_t rows, float* grad_x,
                                    ld_stream_t stream) {
  if (!x || !grad_out || !grad_x || rows < 0) return LD_EINVAL;
  if (rows == 0) return 0;
  LD_LAUNCH(integral_rows_bwd_kernel, grid_for(rows * 4, 128), dim3(128),
                     0, LD_STREAM, x, grad_out, rows, grad_x);
  return (int)hipGetLastError();
}

extern "C" int ld_bbox_overlaps(const float* b1, const float* b2, int64_t m,
                                int64_t n, int mode, int aligned, float eps,
                                float* out, ld_stream_t stream) {
  if (!b1 || !b2 || !out || m < 0 || n < 0 || mode < 0 || mode > 3)
    return LD_EINVAL;
  if (aligned && m != n) return LD_EINVAL;
  const int64_t total = aligned ? m : m * n;
  if (total == 0) return 0;
  LD_LAUNCH(bbox_overlaps_kernel, grid_for(total, 256), dim3(256), 0,
                     LD_STREAM, b1, b2, m, n, mode, aligned, eps, out);
  return (int)hipGetLastError();
}

extern "C" int ld_sum(const float* x, int64_t n, float* out, void* workspace,
                      size_t workspace_bytes, ld_stream_t stream) {
  if (!x || !out || n < 0) return LD_EINVAL;
  if (!workspace || workspace_bytes < kSumBlocks * sizeof(float))
    return LD_ENOSPACE;
  float* part = (float*)workspace;
  LD_LAUNCH(sum_stage1, dim3(kSumBlocks), dim3(256), 0, LD_STREAM, x, n,
                     part);
  LD_LAUNCH(sum_stage1, dim3(1), dim3(256), 0, LD_STREAM, part,
                     (int64_t)kSumBlocks, out);
  return (int)hipGetLastError();
}

extern "C" int ld_abi_version(void) { return 10; }
extern "C" const char* ld_target_arch(void) { return "gfx950"; }
