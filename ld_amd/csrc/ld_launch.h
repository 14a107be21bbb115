// Launch lists (round 5, VERDICT r4 next #4: "get Python out of the per-launch
// path").  Every kernel launch of libldhip.so goes through LD_LAUNCH.  While the
// calling thread is RECORDING (ld_record_begin ... ld_record_end, record.hip) each
// launch is also appended -- kernel, grid, a by-value copy of its argument structs
// -- to a list that ld_record_replay later re-issues on any stream with one C
// loop: the shape-table lookups, descriptor building and the ~25 us of Python per
// launch happened once, at record time.  The frozen teacher's whole forward (~150
// launches per step, 1/4 of the step's) is replayed this way
// (ld_amd/detectors.py TeacherPlan); the buffers a recording points at are kept
// alive by the host side.  Not recording costs one thread-local load per launch.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <vector>

namespace ldrec {
struct Recorder {
  std::vector<std::function<hipError_t(hipStream_t)>> ops;
};
// the recorder of THIS thread (null when not recording); defined in record.hip
Recorder*& active();

template <typename K, typename... A>
inline void launch(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t stream,
                   A... args) {
  if (Recorder* r = active())
    r->ops.emplace_back([=](hipStream_t s) -> hipError_t {
      hipLaunchKernelGGL(kernel, grid, block, shmem, s, args...);
      return hipGetLastError();
    });
  hipLaunchKernelGGL(kernel, grid, block, shmem, stream, args...);
}

inline hipError_t memset_async(void* p, int v, size_t bytes, hipStream_t stream) {
  if (Recorder* r = active())
    r->ops.emplace_back(
        [=](hipStream_t s) -> hipError_t { return hipMemsetAsync(p, v, bytes, s); });
  return hipMemsetAsync(p, v, bytes, stream);
}

// Device-to-device copy as a KERNEL launch (record.hip): a hipMemcpyAsync captured
// into a hipGraph becomes a 1-D memcpy node whose parameters this runtime does not
// hand back (hipGraphMemcpyNodeGetParams returns an unfilled 3-D descriptor), so a
// step list (graphlist.hip) could not re-issue it.
hipError_t copy_d2d_launch(void* dst, const void* src, size_t bytes, hipStream_t stream);

inline hipError_t memcpy_d2d_async(void* dst, const void* src, size_t bytes,
                                   hipStream_t stream) {
  if (Recorder* r = active())
    r->ops.emplace_back([=](hipStream_t s) -> hipError_t {
      return copy_d2d_launch(dst, src, bytes, s);
    });
  return copy_d2d_launch(dst, src, bytes, stream);
}
}  // namespace ldrec

#define LD_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  ldrec::launch(kernel, grid, block, shmem, stream, ##__VA_ARGS__)
