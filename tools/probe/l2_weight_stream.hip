// Probe (tools only, not part of libldhip.so): the rate at which EVERY workgroup
// of a launch can stream the SAME L2-resident buffer -- the access pattern of a
// fused frozen-teacher bottleneck kernel (1x1 -> 3x3 -> 1x1 with the
// intermediates in LDS), whose every workgroup must walk the block's whole bf16
// weight set (R101 layer3: 256x1024 + 256x2304 + 1024x256 = 2.2 MB) for its
// spatial tile.  VERDICT r4 next #1(a): replace the paper price of DESIGN 3.4
// with this number.  Each wave issues 16-byte per-lane buffer loads, UNROLL of
// them in flight, all workgroups walk the buffer in the same order (workgroup w
// starts `skew * w` rows in, skew = 0 for lock-step).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int UNROLL>
__global__ __launch_bounds__(256) void weight_stream_kernel(const void* buf, unsigned bytes,
                                                            int reps, unsigned skew,
                                                            float* out) {
  const rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, bytes, 0x00020000);
  const unsigned tid = threadIdx.x;
  const unsigned rows = bytes / 4096u;  // one row = 256 threads x 16 B
  unsigned row = (blockIdx.x * skew) % rows;
  unsigned acc = 0;
  for (int rep = 0; rep < reps; ++rep) {
    for (unsigned i = 0; i < rows; i += UNROLL) {
      uintx4 v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        unsigned rr = row + (unsigned)u;
        rr = rr >= rows ? rr - rows : rr;
        v[u] = __builtin_bit_cast(
            uintx4, __builtin_amdgcn_raw_buffer_load_b128(r, rr * 4096u + tid * 16u, 0, 0));
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) acc += v[u][0] ^ v[u][3];
      row += UNROLL;
      row = row >= rows ? row - rows : row;
    }
  }
  out[blockIdx.x * 256 + tid] = __builtin_bit_cast(float, acc);
}

extern "C" int probe_weight_stream(const void* buf, unsigned bytes, int blocks, int reps,
                                   int unroll, unsigned skew, float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (unroll == 4)
    hipLaunchKernelGGL(weight_stream_kernel<4>, dim3(blocks), dim3(256), 0, st, buf, bytes, reps,
                       skew, out);
  else if (unroll == 8)
    hipLaunchKernelGGL(weight_stream_kernel<8>, dim3(blocks), dim3(256), 0, st, buf, bytes, reps,
                       skew, out);
  else
    hipLaunchKernelGGL(weight_stream_kernel<16>, dim3(blocks), dim3(256), 0, st, buf, bytes,
                       reps, skew, out);
  return (int)hipGetLastError();
}
