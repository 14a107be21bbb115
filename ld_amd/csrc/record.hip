// Launch lists: the recording side of ld_launch.h and its C ABI.
#include <hip/hip_runtime.h>

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
