// What does ds_read_b64_tr_b16 return?  LDS holds element index i at 16-bit slot
// i; every lane passes its own byte address; the four 16-bit results per lane
// are written out.  tools/probe/run_tr_probe.py tries address patterns.
#include <hip/hip_runtime.h>

__global__ void tr_probe_kernel(const int* __restrict__ addr, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)(&lds[0]) + (unsigned)addr[threadIdx.x];
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j)
    out[threadIdx.x * 4 + j] = (unsigned short)(v >> (16 * j));
}

extern "C" int tr_probe(const int* addr, unsigned short* out, void* stream) {
  hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, addr, out);
  return (int)hipGetLastError();
}
