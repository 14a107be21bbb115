// Micro-probe (tools only, not part of libldhip.so): how many bytes per clock a
// CU can pull through the vector-memory path from an L2-resident buffer with
// 4-byte vs 16-byte per-lane loads -- the rate that bounds the bf16 convolution's
// fp32 activation gathers.  out[0..] gets checksums so nothing is optimised away.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int W>  // W = dwords per lane per load (1 or 4)
__global__ __launch_bounds__(256) void load_rate_kernel(const float* buf, unsigned bytes,
                                                         int iters, float* out) {
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, bytes, 0x00020000);
  const unsigned lane_off = (threadIdx.x & 63) * 4u * W;
  const unsigned wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
  // each wave walks the buffer with a stride that keeps it inside `bytes`
  const unsigned mask = bytes - 1u;  // bytes is a power of two
  unsigned base = (wave * 64u * 4u * W * 16u) & mask;
  float acc = 0.0f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const unsigned off = ((base + u * 64u * 4u * W) & mask) + lane_off;
      if (W == 1) {
        acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
      } else {
        const uintx4 v = __builtin_bit_cast(
            uintx4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
        acc += __builtin_bit_cast(float, v[0]) + __builtin_bit_cast(float, v[3]);
      }
    }
    base = (base + 16u * 64u * 4u * W * 1031u) & mask;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

extern "C" int probe_load_rate(const float* buf, unsigned bytes, int width, int blocks,
                               int iters, float* out, void* stream) {
  if (width == 1)
    hipLaunchKernelGGL(load_rate_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       buf, bytes, iters, out);
  else
    hipLaunchKernelGGL(load_rate_kernel<4>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       buf, bytes, iters, out);
  return (int)hipGetLastError();
}
