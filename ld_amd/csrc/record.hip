// Launch lists: the recording side of ld_launch.h and its C ABI.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>

#include "../../include/ld_hip.h"
#include "ld_launch.h"

namespace ldrec {
Recorder*& active() {
  static thread_local Recorder* r = nullptr;
  return r;
}
}  // namespace ldrec

namespace {
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

// 16 bytes per lane, grid-stride; the byte tail by the last lanes
__global__ __launch_bounds__(256) void copy_d2d_kernel(uintx4* __restrict__ dst,
                                                       const uintx4* __restrict__ src,
                                                       size_t n16, size_t bytes) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
    dst[i] = src[i];
  const size_t tail = bytes - n16 * 16;
  if (blockIdx.x == 0 && threadIdx.x < tail)
    ((char*)dst)[n16 * 16 + threadIdx.x] = ((const char*)src)[n16 * 16 + threadIdx.x];
}

__global__ __launch_bounds__(256) void copy_d2d_bytes_kernel(char* __restrict__ dst,
                                                             const char* __restrict__ src,
                                                             size_t bytes) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < bytes; i += stride)
    dst[i] = src[i];
}
}  // namespace

namespace ldrec {
hipError_t copy_d2d_launch(void* dst, const void* src, size_t bytes, hipStream_t stream) {
  if (!bytes) return hipSuccess;
  if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
    const size_t n16 = bytes / 16;
    const size_t blocks = std::min<size_t>((n16 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(copy_d2d_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       (uintx4*)dst, (const uintx4*)src, n16, bytes);
  } else {
    const size_t blocks = std::min<size_t>((bytes + 255) / 256, 2048);
    hipLaunchKernelGGL(copy_d2d_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
                       (char*)dst, (const char*)src, bytes);
  }
  return hipGetLastError();
}
}  // namespace ldrec

// Device-to-device copy of `bytes` bytes as one kernel launch (capturable into a
// hipGraph as a kernel node; see ld_launch.h).
extern "C" int ld_copy_d2d(void* dst, const void* src, size_t bytes, ld_stream_t stream) {
  if ((!dst || !src) && bytes) return LD_EINVAL;
  return (int)ldrec::memcpy_d2d_async(dst, src, bytes, (hipStream_t)stream);
}

namespace {
std::mutex g_mu;
std::unordered_map<int64_t, ldrec::Recorder*> g_lists;
int64_t g_next = 1;
}  // namespace

extern "C" int ld_record_begin(void) {
  if (ldrec::active()) return LD_EINVAL;  // no nesting
  ldrec::active() = new ldrec::Recorder();
  return 0;
}

// Ends the recording of this thread; returns its handle (> 0), or a negative
// error.  A list may be empty.
extern "C" int64_t ld_record_end(void) {
  ldrec::Recorder* r = ldrec::active();
  if (!r) return (int64_t)LD_EINVAL;
  ldrec::active() = nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  const int64_t h = g_next++;
  g_lists[h] = r;
  return h;
}

// Drops the recording in progress (error paths).
extern "C" int ld_record_abort(void) {
  delete ldrec::active();
  ldrec::active() = nullptr;
  return 0;
}

extern "C" int ld_record_count(int64_t handle) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_lists.find(handle);
  return it == g_lists.end() ? LD_EINVAL : (int)it->second->ops.size();
}

// Re-issues every launch of the list on `stream`, in order.  Only enqueues.
extern "C" int ld_record_replay(int64_t handle, ld_stream_t stream) {
  ldrec::Recorder* r;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_lists.find(handle);
    if (it == g_lists.end()) return LD_EINVAL;
    r = it->second;
  }
  if (ldrec::active() == r) return LD_EINVAL;
  for (auto& op : r->ops)
    if (hipError_t e = op((hipStream_t)stream)) return (int)e;
  return 0;
}

extern "C" int ld_record_free(int64_t handle) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_lists.find(handle);
  if (it == g_lists.end()) return LD_EINVAL;
  delete it->second;
  g_lists.erase(it);
  return 0;
}

// ---- stream ordering without the host language in the path ---------------------
// `to` continues after everything enqueued on `from` so far: one event record +
// one stream wait on a reusable per-device event (a wait binds to the record
// that precedes it; re-recording the event later does not affect it).  What
// torch's side.wait_stream(main) does, as ONE call from the host language: the
// weight gradients of the backward pass fork onto their side stream ~65 times
// per step.  Not for use on a capturing stream.
extern "C" int ld_stream_fork(ld_stream_t from, ld_stream_t to) {
  static std::mutex mu;
  static hipEvent_t events[64] = {};
  int dev = 0;
  if (hipError_t e = hipGetDevice(&dev)) return (int)e;
  if (dev < 0 || dev >= 64) return LD_EINVAL;
  std::lock_guard<std::mutex> lock(mu);
  if (!events[dev])
    if (hipError_t e = hipEventCreateWithFlags(&events[dev], hipEventDisableTiming))
      return (int)e;
  if (hipError_t e = hipEventRecord(events[dev], (hipStream_t)from)) return (int)e;
  if (hipError_t e = hipStreamWaitEvent((hipStream_t)to, events[dev], 0)) return (int)e;
  return 0;
}
