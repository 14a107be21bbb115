// HBM streaming ceilings, measured by bench.py in the same process as the LD-KL
// kernel (and by tools/ldkl_variants.py): what the memory system gives for
//   (a) a plain copy (1 read : 1 write), 4 / 16 bytes per lane, NT or not;
//   (b) the LD-KL kernel's own pattern with the arithmetic removed: 34 read
//       planes + 17 write planes per side, channel planes `rows` floats apart,
//       2 reads : 1 write -- the ceiling the fused kernel can approach.
// Measurement support, not on the train step.  gfx950 only.
#include <hip/hip_runtime.h>

#include "ld_launch.h"
#include <stdint.h>

#include "../../include/ld_hip.h"

namespace {
typedef float f4 __attribute__((ext_vector_type(4)));

template <int R, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const float* __restrict__ src,
                                                   float* __restrict__ dst, int64_t n) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * R;
  if (i + R > n) return;
  if (R == 4) {
    f4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(src + i))
              : *reinterpret_cast<const f4*>(src + i);
    if (NT)
      __builtin_nontemporal_store(v, reinterpret_cast<f4*>(dst + i));
    else
      *reinterpret_cast<f4*>(dst + i) = v;
  } else {
    float v = NT ? __builtin_nontemporal_load(src + i) : src[i];
    if (NT)
      __builtin_nontemporal_store(v, dst + i);
    else
      dst[i] = v;
  }
}

// thread = (row, side); 17 + 17 planes in, 17 planes out; out = s - t
// SIDE_FAST: the 4 sides of a 256-row chunk in 4 adjacent workgroups (one pass
// over the 68 planes) instead of one sweep per side
template <bool NT, bool SIDE_FAST>
__global__ __launch_bounds__(256) void planes_kernel(const float* __restrict__ s,
                                                     const float* __restrict__ t,
                                                     float* __restrict__ g, int64_t rows) {
  const int side = SIDE_FAST ? (int)(blockIdx.x & 3) : (int)blockIdx.y;
  const int64_t r = (int64_t)(SIDE_FAST ? blockIdx.x >> 2 : blockIdx.x) * 256 + threadIdx.x;
  if (r >= rows) return;
  float a[17], b[17];
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    const float* ps = s + (int64_t)(side * 17 + k) * rows + r;
    const float* pt = t + (int64_t)(side * 17 + k) * rows + r;
    a[k] = NT ? __builtin_nontemporal_load(ps) : *ps;
    b[k] = NT ? __builtin_nontemporal_load(pt) : *pt;
  }
#pragma unroll
  for (int k = 0; k < 17; ++k) {
    float* pg = g + (int64_t)(side * 17 + k) * rows + r;
    if (NT)
      __builtin_nontemporal_store(a[k] - b[k], pg);
    else
      *pg = a[k] - b[k];
  }
}
}  // namespace

extern "C" int ld_probe_copy(const float* src, float* dst, int64_t n, int width, int nt,
                             ld_stream_t stream) {
  if (!src || !dst || n < 1 || (width != 1 && width != 4) || n % width) return LD_EINVAL;
  const int64_t threads = n / width;
  const dim3 grid((unsigned)((threads + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  if (width == 4) {
    if (nt) LD_LAUNCH((copy_kernel<4, true>), grid, dim3(256), 0, st, src, dst, n);
    else LD_LAUNCH((copy_kernel<4, false>), grid, dim3(256), 0, st, src, dst, n);
  } else {
    if (nt) LD_LAUNCH((copy_kernel<1, true>), grid, dim3(256), 0, st, src, dst, n);
    else LD_LAUNCH((copy_kernel<1, false>), grid, dim3(256), 0, st, src, dst, n);
  }
  return (int)hipGetLastError();
}

extern "C" int ld_probe_planes(const float* s, const float* t, float* g, int64_t rows,
                               int nt, int side_fast, ld_stream_t stream) {
  if (!s || !t || !g || rows < 1) return LD_EINVAL;
  const unsigned nb = (unsigned)((rows + 255) / 256);
  const dim3 grid = side_fast ? dim3(nb * 4, 1) : dim3(nb, 4);
  hipStream_t st = (hipStream_t)stream;
  if (side_fast) {
    if (nt) LD_LAUNCH((planes_kernel<true, true>), grid, dim3(256), 0, st, s, t, g, rows);
    else LD_LAUNCH((planes_kernel<false, true>), grid, dim3(256), 0, st, s, t, g, rows);
  } else {
    if (nt) LD_LAUNCH((planes_kernel<true, false>), grid, dim3(256), 0, st, s, t, g, rows);
    else LD_LAUNCH((planes_kernel<false, false>), grid, dim3(256), 0, st, s, t, g, rows);
  }
  return (int)hipGetLastError();
}
