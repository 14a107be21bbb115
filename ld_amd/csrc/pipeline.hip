// Device input pipeline for gfx950: what the
// reference's train_pipeline does per image on CPU workers
//   Resize(img_scale=(1333, 800), keep_ratio=True) -> RandomFlip ->
//   Normalize(mean, std, to_rgb=True) -> Pad(size_divisor=32) -> collate
//   (configs/ld/ld_r18_gflv1_r101_fpn_coco_1x.py:66-77,
//    mmdet/datasets/pipelines/transforms.py:203-233,416-450,524-580)
// as ONE launch for a whole batch: decoded uint8 HWC images (BGR, as
// cv2.imread / LoadImageFromFile deliver them) already on the device ->
// the padded fp32 NCHW batch tensor the detector consumes.  At 2 workers per
// GPU (data.workers_per_gpu=2) the CPU pipeline cannot feed an MI355X at the
// measured step rate; this kernel is HBM-bound (3 bytes in, 12 bytes out per
// pixel: 2 x 800 x 1344 x 15 B = 32 MB -> a few microseconds).
//
// Arithmetic restated from the published behaviour of the libraries the
// reference calls (absent from the checkout: mmcv.imrescale -> cv2.resize
// INTER_LINEAR on 8-bit data, mmcv.imflip, mmcv.imnormalize):
//   * bilinear with pixel-centre alignment, source index clamped at the
//     borders, 11-bit fixed-point coefficients and a rounded 8-bit result
//     (cv2's INTER_RESIZE_COEF_BITS = 11 path; the result is uint8 BEFORE the
//     normalisation, as in the reference);
//   * horizontal flip of the resized image; BGR -> RGB; (v - mean) * (1 / std)
//     in fp32; zeros in the padding.
// Parity status: UNPINNED against cv2 itself (checked bit-exactly against the
// numpy restatement in oracle/pipeline_oracle.py).
#include <hip/hip_runtime.h>

#include "ld_launch.h"

#include "../../include/ld_hip.h"

namespace {

constexpr int kCoefBits = 11, kCoefOne = 1 << kCoefBits;

__device__ __forceinline__ void lin_coef(int d, double scale, int ssize, int& s0,
                                         int& s1, int& a0, int& a1) {
  // cv2 resize (linear): fx = (d + 0.5) * scale - 0.5
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) {
    f = 0.0f;
    s = 0;
  }
  if (s >= ssize - 1) {
    f = 0.0f;
    s = ssize - 1;
  }
  s0 = s;
  s1 = min(s + 1, ssize - 1);
  // saturate_cast<short>(x * 2048): round to nearest even like cvRound
  a0 = (int)rintf((1.0f - f) * (float)kCoefOne);
  a1 = (int)rintf(f * (float)kCoefOne);
}

__global__ __launch_bounds__(256) void preprocess_kernel(
    const ld_image_t* __restrict__ imgs, int Hpad, int Wpad, float mean0, float mean1,
    float mean2, float inv0, float inv1, float inv2, int to_rgb,
    float* __restrict__ out) {
  const int n = blockIdx.z;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= Wpad || y >= Hpad) return;
  const ld_image_t im = imgs[n];
  const size_t plane = (size_t)Hpad * Wpad;
  float* o = out + (size_t)n * 3 * plane + (size_t)y * Wpad + x;
  if (y >= im.new_h || x >= im.new_w) {  // Pad(size_divisor) / collate: zeros
    o[0] = 0.0f;
    o[plane] = 0.0f;
    o[2 * plane] = 0.0f;
    return;
  }
  const int xr = im.flip ? im.new_w - 1 - x : x;  // flip AFTER the resize
  int sx0, sx1, ax0, ax1, sy0, sy1, ay0, ay1;
  lin_coef(xr, (double)im.src_w / (double)im.new_w, im.src_w, sx0, sx1, ax0, ax1);
  lin_coef(y, (double)im.src_h / (double)im.new_h, im.src_h, sy0, sy1, ay0, ay1);
  const unsigned char* r0 = im.data + ((size_t)sy0 * im.src_w) * 3;
  const unsigned char* r1 = im.data + ((size_t)sy1 * im.src_w) * 3;
  const float mean[3] = {mean0, mean1, mean2}, inv[3] = {inv0, inv1, inv2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {  // c = output channel
    const int cs = to_rgb ? 2 - c : c;  // source (BGR) channel
    const int h0 = r0[sx0 * 3 + cs] * ax0 + r0[sx1 * 3 + cs] * ax1;
    const int h1 = r1[sx0 * 3 + cs] * ax0 + r1[sx1 * 3 + cs] * ax1;
    // FixedPtCast<int, uchar, 2 * 11>: (v + (1 << 21)) >> 22
    int v = (h0 * ay0 + h1 * ay1 + (1 << (2 * kCoefBits - 1))) >> (2 * kCoefBits);
    v = min(max(v, 0), 255);
    o[(size_t)c * plane] = ((float)v - mean[c]) * inv[c];
  }
}

}  // namespace

extern "C" int ld_preprocess_batch(const ld_image_t* imgs, int N, int Hpad, int Wpad,
                                   const float* mean, const float* std_inv, int to_rgb,
                                   float* out, ld_stream_t stream) {
  if (!imgs || !mean || !std_inv || !out || N < 1 || Hpad < 1 || Wpad < 1)
    return LD_EINVAL;
  LD_LAUNCH(preprocess_kernel, dim3((Wpad + 63) / 64, (Hpad + 3) / 4, N),
                     dim3(256), 0, (hipStream_t)stream, imgs, Hpad, Wpad, mean[0],
                     mean[1], mean[2], std_inv[0], std_inv[1], std_inv[2], to_rgb, out);
  return (int)hipGetLastError();
}
