// Deformable convolution v1, forward only, for gfx950: the conv2 of the R101-DCN
// teacher's Bottlenecks in stages c3-c5 (BASELINE.json config 4;
// mmdet/models/backbones/resnet.py:171-194 with dcn=dict(type='DCN',
// deform_groups=1) from configs/gfl/gfl_r101_fpn_dconv_c3-c5_mstrain_2x_coco.py).
//
// The arithmetic lives in mmcv-full (mmcv.ops.DeformConv2dPack, pinned
// 1.2.4-1.3; absent from the reference checkout), so it is restated from the
// published algorithm (Dai et al. 2017, "Deformable Convolutional Networks";
// mmcv/ops/csrc/deform_conv_cuda_kernel.cuh deformable_im2col):
//   offset = conv3x3(x) with 2*KH*KW channels, (dy, dx) interleaved per tap
//   y[co][p] = sum_{ci,k} W[co][ci][k] * bilinear(x[ci], p*stride - pad + k*dil + offset_k(p))
//   bilinear: zero outside (-1, H) x (-1, W); neighbours outside the map are 0.
// Parity status: UNPINNED against mmcv itself (no golden vectors exist for it in
// the reference); checked against an independent torch-CPU restatement
// (oracle/dcn_oracle.py).
//
// MI355X mapping: the sampled patches are written once as a column tensor
// col (N, Cin*KH*KW, Pout) -- channel = ci*KH*KW + k, exactly the order of
// weight.view(Cout, Cin*KH*KW) -- and the product with the weights is the
// existing MFMA implicit GEMM as a 1x1 convolution over Cin*KH*KW channels
// (BN/ReLU folded into its epilogue).  Thread = (n, tap, position): the
// sampling location and the four bilinear weights are computed once and reused
// over the channel loop; consecutive lanes are consecutive positions, so both the
// gathers (neighbouring cells) and the column stores are coalesced.  HBM-bound:
// Cin*KH*KW*4 B written per output position.
#include <hip/hip_runtime.h>

#include "ld_launch.h"

#include "../../include/ld_hip.h"

namespace {

__global__ __launch_bounds__(256) void deform_im2col_kernel(
    const float* __restrict__ x, const float* __restrict__ offset, int Cin, int Hin,
    int Win, int Hout, int Wout, int KH, int KW, int stride, int pad, int dil,
    float* __restrict__ col) {
  const int Pout = Hout * Wout, Pin = Hin * Win, ntaps = KH * KW;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int k = blockIdx.y, n = blockIdx.z;
  if (p >= Pout) return;
  const int ho = p / Wout, wo = p - ho * Wout;
  const int kh = k / KW, kw = k - kh * KW;
  const float* off = offset + (size_t)n * 2 * ntaps * Pout;
  const float h = (float)(ho * stride - pad + kh * dil) + off[(size_t)(2 * k) * Pout + p];
  const float w = (float)(wo * stride - pad + kw * dil) + off[(size_t)(2 * k + 1) * Pout + p];
  float w1 = 0.0f, w2 = 0.0f, w3 = 0.0f, w4 = 0.0f;
  int o1 = 0, o2 = 0, o3 = 0, o4 = 0;
  if (h > -1.0f && w > -1.0f && h < (float)Hin && w < (float)Win) {
    const int hl = (int)floorf(h), wl = (int)floorf(w);
    const int hh_ = hl + 1, wh = wl + 1;
    const float lh = h - (float)hl, lw = w - (float)wl;
    const float uh = 1.0f - lh, uw = 1.0f - lw;
    const bool t = hl >= 0, b = hh_ <= Hin - 1, l = wl >= 0, r = wh <= Win - 1;
    if (t && l) { w1 = uh * uw; o1 = hl * Win + wl; }
    if (t && r) { w2 = uh * lw; o2 = hl * Win + wh; }
    if (b && l) { w3 = lh * uw; o3 = hh_ * Win + wl; }
    if (b && r) { w4 = lh * lw; o4 = hh_ * Win + wh; }
  }
  const float* xn = x + (size_t)n * Cin * Pin;
  float* cn = col + ((size_t)n * Cin * ntaps + k) * Pout + p;
  const size_t cstep = (size_t)ntaps * Pout;
#pragma unroll 4
  for (int ci = 0; ci < Cin; ++ci) {
    const float* xc = xn + (size_t)ci * Pin;
    // same order of the four terms as deformable_im2col_bilinear
    const float v = w1 * xc[o1] + w2 * xc[o2] + w3 * xc[o3] + w4 * xc[o4];
    cn[(size_t)ci * cstep] = v;
  }
}

}  // namespace

extern "C" int ld_deform_im2col(const float* x, const float* offset, int N, int Cin,
                                int Hin, int Win, int KH, int KW, int stride, int pad,
                                int dilation, float* col, ld_stream_t stream) {
  if (!x || !offset || !col || N < 1 || Cin < 1 || Hin < 1 || Win < 1 || KH < 1 ||
      KW < 1 || stride < 1 || dilation < 1 || pad < 0)
    return LD_EINVAL;
  const int Hout = (Hin + 2 * pad - dilation * (KH - 1) - 1) / stride + 1;
  const int Wout = (Win + 2 * pad - dilation * (KW - 1) - 1) / stride + 1;
  if (Hout < 1 || Wout < 1) return LD_EINVAL;
  const int Pout = Hout * Wout;
  LD_LAUNCH(deform_im2col_kernel, dim3((Pout + 255) / 256, KH * KW, N),
                     dim3(256), 0, (hipStream_t)stream, x, offset, Cin, Hin, Win, Hout,
                     Wout, KH, KW, stride, pad, dilation, col);
  return (int)hipGetLastError();
}
