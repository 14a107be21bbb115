// Grouped convolution, forward only, for gfx950 -- the 3x3 conv2 of ResNeXt's
// Bottleneck (mmdet/models/backbones/resnext.py:49-61: groups = 32, width =
// floor(planes * base_width / 64) * groups, i.e. 4 / 8 / 16 / 32 channels per
// group in the four stages of ResNeXt-101 32x4d), the frozen X-101 teacher of
// BASELINE config 5.  With K = 1 it is also the grouped GEMM behind a grouped
// deformable conv (columns of ld_deform_im2col: Cin*9 rows, groups of cg*9).
//
// Roofline: a grouped conv does cg * 9 * 2 flop per 8 bytes it must move (one
// fp32 in, one out per channel and position) = 9 ... 72 flop/B for cg = 4 ... 32,
// every layer of the net is 1.24 GFLOP on 17-138 MB of activations: HBM- /
// L1-bound work, NOT an MFMA GEMM (a 32-wide MFMA tile would be 8x ... 1x
// block-diagonal zeros).  So: VALU FMAs with the weights as SCALAR operands.
//   thread = one output position of one group: acc[CG] in registers;
//   the 64 lanes of a wavefront read 64 consecutive positions of each input
//     channel and tap (coalesced; the 9 taps of a channel re-hit the same
//     lines in L1);
//   the group's weights sit in the image [g][ci][tap][co]: for a (ci, tap) the
//     CG out-channel weights are CG consecutive dwords at a wave-uniform
//     address -> s_load_dwordx4/x8/x16 + v_fma with an SGPR operand, no LDS.
// Epilogue as the dense convs': y = relu(scale[c] * acc + shift[c]).
#include <hip/hip_runtime.h>

#include "ld_launch.h"

#include "../../include/ld_hip.h"

namespace {

// w (Cout, cin_g, KK) -> image (G, cin_g, KK, CG), CG = Cout / G
__global__ __launch_bounds__(256) void gconv_weight_image_kernel(
    const float* __restrict__ w, float* __restrict__ img, int G, int CG, int cin_g,
    int KK) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int total = G * CG * cin_g * KK;
  if (i >= total) return;
  // i indexes the image: ((g * cin_g + ci) * KK + t) * CG + co
  const int co = i % CG;
  int r = i / CG;
  const int t = r % KK;
  r /= KK;
  const int ci = r % cin_g, g = r / cin_g;
  img[i] = w[((size_t)(g * CG + co) * cin_g + ci) * KK + t];
}

template <int CG, int K>
__global__ __launch_bounds__(256) void gconv_forward_kernel(
    const float* __restrict__ x, const float* __restrict__ wimg, float* __restrict__ y,
    int Cin, int Cout, int cin_g, int stride, int pad, int Hin, int Win, int Hout,
    int Wout, const float* __restrict__ scale, const float* __restrict__ shift,
    int relu) {
  constexpr int KK = K * K;
  const int g = blockIdx.y, n = blockIdx.z;
  const int p = blockIdx.x * 256 + threadIdx.x;
  const int Pout = Hout * Wout, Pin = Hin * Win;
  const bool live = p < Pout;
  const int ho = live ? p / Wout : 0, wo = live ? p - (p / Wout) * Wout : 0;
  int off[KK];
  bool ok[KK];
#pragma unroll
  for (int kh = 0; kh < K; ++kh)
#pragma unroll
    for (int kw = 0; kw < K; ++kw) {
      const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
      ok[kh * K + kw] = live && hi >= 0 && hi < Hin && wi >= 0 && wi < Win;
      off[kh * K + kw] = hi * Win + wi;
    }
  float acc[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) acc[c] = 0.0f;
  const float* xg = x + ((size_t)n * Cin + (size_t)g * cin_g) * Pin;
  const float* wg = wimg + (size_t)g * cin_g * KK * CG;  // wave-uniform
  for (int ci = 0; ci < cin_g; ++ci) {
    const float* xc = xg + (size_t)ci * Pin;
    float xv[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) xv[t] = ok[t] ? xc[off[t]] : 0.0f;
#pragma unroll
    for (int t = 0; t < KK; ++t) {
      const float* wr = wg + ((size_t)ci * KK + t) * CG;
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[c] = fmaf(wr[c], xv[t], acc[c]);
    }
  }
  if (!live) return;
  float* yg = y + ((size_t)n * Cout + (size_t)g * CG) * Pout + p;
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    float v = acc[c];
    if (scale) v = v * scale[g * CG + c] + shift[g * CG + c];
    if (relu) v = fmaxf(v, 0.0f);
    yg[(size_t)c * Pout] = v;
  }
}

template <int K>
int launch_gconv(int CG, dim3 grid, hipStream_t st, const float* x, const float* wimg,
                 float* y, int Cin, int Cout, int cin_g, int stride, int pad, int Hin,
                 int Win, int Hout, int Wout, const float* scale, const float* shift,
                 int relu) {
#define LD_GC(C)                                                                      \
  LD_LAUNCH((gconv_forward_kernel<C, K>), grid, dim3(256), 0, st, x, wimg, y, \
                     Cin, Cout, cin_g, stride, pad, Hin, Win, Hout, Wout, scale, shift, \
                     relu)
  switch (CG) {
    case 4: LD_GC(4); break;
    case 8: LD_GC(8); break;
    case 16: LD_GC(16); break;
    case 32: LD_GC(32); break;
    default: return LD_EUNSUPPORTED;
  }
#undef LD_GC
  return (int)hipGetLastError();
}

}  // namespace

extern "C" size_t ld_gconv_weight_image_floats(int Cout, int Cin, int groups, int K) {
  if (groups < 1 || Cout % groups || Cin % groups) return 0;
  return (size_t)Cout * (Cin / groups) * K * K;
}

extern "C" int ld_gconv_weight_transform(const float* w, int Cout, int Cin, int groups,
                                         int K, float* image, ld_stream_t stream) {
  if (!w || !image || groups < 1 || Cout % groups || Cin % groups || K < 1)
    return LD_EINVAL;
  const int total = Cout * (Cin / groups) * K * K;
  LD_LAUNCH(gconv_weight_image_kernel, dim3((total + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, w, image, groups, Cout / groups, Cin / groups,
                     K * K);
  return (int)hipGetLastError();
}

extern "C" int ld_gconv_forward(const float* x, const float* wimage, float* y, int N,
                                int Cin, int Cout, int groups, int K, int stride, int pad,
                                int Hin, int Win, const float* scale, const float* shift,
                                int relu, ld_stream_t stream) {
  if (!x || !wimage || !y || N < 1 || groups < 1 || Cin % groups || Cout % groups ||
      stride < 1 || pad < 0 || Hin < 1 || Win < 1 || (scale == nullptr) != (shift == nullptr))
    return LD_EINVAL;
  if (K != 1 && K != 3) return LD_EUNSUPPORTED;
  const int Hout = (Hin + 2 * pad - K) / stride + 1, Wout = (Win + 2 * pad - K) / stride + 1;
  if (Hout < 1 || Wout < 1) return LD_EINVAL;
  const int CG = Cout / groups, cin_g = Cin / groups;
  const dim3 grid((Hout * Wout + 255) / 256, groups, N);
  hipStream_t st = (hipStream_t)stream;
  if (K == 3)
    return launch_gconv<3>(CG, grid, st, x, wimage, y, Cin, Cout, cin_g, stride, pad, Hin,
                           Win, Hout, Wout, scale, shift, relu);
  return launch_gconv<1>(CG, grid, st, x, wimage, y, Cin, Cout, cin_g, stride, pad, Hin,
                         Win, Hout, Wout, scale, shift, relu);
}
