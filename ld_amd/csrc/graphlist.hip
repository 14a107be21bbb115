// Step lists (round 6): a captured hipGraph re-issued as plain stream launches.
//
// The whole LD train step -- student forward, loss block, backward, optimizer,
// the teacher of the next batch on its side stream: ~500 launches -- captures into
// a hipGraph (ld_amd/train.py GraphedStep), but hipGraphLaunch on this runtime
// costs 10-12 ms of host time per replay (~22 us per node, measured in round 5:
// profiles/r05_graph_launch_knobs.jsonl) and its executor serialises the branches
// through cross-queue barriers; eager issue from Python costs ~25 us per launch.
// A plain hipLaunchKernel from C costs ~3-4 us.  So the captured graph is only
// used as the RECORD: this file walks its nodes once (kernel / memcpy / memset
// parameters, dependency edges), assigns every node to one of a few stream
// "lanes" so that the capture's concurrency survives (the weight gradients and
// the teacher ran on side streams), and a replay is one C loop of
// hipLaunchKernel / hipMemcpyAsync / hipMemsetAsync calls plus an event record /
// wait pair per cross-lane edge.  The hipGraph must stay alive (its nodes own the
// argument copies the launches point at) and so must every buffer the captured
// step touched -- both are what the owner of the capture (torch.cuda.CUDAGraph with
// keep_graph=True and its private pool) guarantees.  Nothing in the reference:
// mmcv's runner issues every kernel from Python (mmdet/apis/train.py:74-127).
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/ld_hip.h"

namespace {

struct Node {
  hipGraphNodeType type;
  hipKernelNodeParams kp;
  hipMemsetParams ms;
  hipMemcpy3DParms cp;
  std::vector<int> preds;
  int lane = 0;
  bool record = false;         // some successor sits on another lane
  std::vector<int> waits;      // predecessors on other lanes
  hipEvent_t event = nullptr;  // recorded after this node (if `record`)
  bool module_fn = false;      // func is a hipFunction_t (hipModuleLaunchKernel)
};

struct StepList {
  std::vector<Node> nodes;
  int nlanes = 1;
  int n_kernel = 0, n_memcpy = 0, n_memset = 0, n_noop = 0, n_cross = 0;
  std::vector<hipStream_t> side;  // lanes 1 .. nlanes - 1 (lane 0 = the caller's)
  hipEvent_t fork = nullptr;
  std::vector<hipEvent_t> join;
  std::vector<int> lane_tail;
};

std::mutex g_mu;
int g_fail[8] = {0};  // last failing node: index, type, hip error, lane, details
std::unordered_map<int64_t, StepList*> g_lists;
int64_t g_next = 1;

void destroy(StepList* s) {
  for (Node& n : s->nodes)
    if (n.event) (void)hipEventDestroy(n.event);
  for (hipStream_t st : s->side) (void)hipStreamDestroy(st);
  if (s->fork) (void)hipEventDestroy(s->fork);
  for (hipEvent_t e : s->join) (void)hipEventDestroy(e);
  delete s;
}

hipError_t issue(Node& n, hipStream_t st) {
  switch (n.type) {
    case hipGraphNodeTypeKernel: {
      if (!n.module_fn) {
        hipError_t e = hipLaunchKernel(n.kp.func, n.kp.gridDim, n.kp.blockDim,
                                       n.kp.kernelParams, n.kp.sharedMemBytes, st);
        if (e != hipErrorInvalidDeviceFunction) return e;
        (void)hipGetLastError();
        n.module_fn = true;  // launched through the module API when captured
      }
      return hipModuleLaunchKernel((hipFunction_t)n.kp.func, n.kp.gridDim.x,
                                   n.kp.gridDim.y, n.kp.gridDim.z, n.kp.blockDim.x,
                                   n.kp.blockDim.y, n.kp.blockDim.z, n.kp.sharedMemBytes,
                                   st, n.kp.kernelParams, n.kp.extra);
    }
    case hipGraphNodeTypeMemcpy:
      if (n.cp.extent.height <= 1 && n.cp.extent.depth <= 1 && !n.cp.srcArray &&
          !n.cp.dstArray)
        return hipMemcpyAsync((char*)n.cp.dstPtr.ptr + n.cp.dstPos.x,
                              (const char*)n.cp.srcPtr.ptr + n.cp.srcPos.x,
                              n.cp.extent.width, n.cp.kind, st);
      return hipMemcpy3DAsync(&n.cp, st);
    case hipGraphNodeTypeMemset:
      if (n.ms.height <= 1) {
        const size_t count = n.ms.width;
        if (n.ms.elementSize == 1) return hipMemsetAsync(n.ms.dst, (int)n.ms.value, count, st);
        if (n.ms.elementSize == 2)
          return hipMemsetD16Async((hipDeviceptr_t)n.ms.dst, (unsigned short)n.ms.value,
                                   count, st);
        return hipMemsetD32Async((hipDeviceptr_t)n.ms.dst, (int)n.ms.value, count, st);
      }
      return hipMemset2DAsync(n.ms.dst, n.ms.pitch, (int)n.ms.value,
                              n.ms.width * n.ms.elementSize, n.ms.height, st);
    default:
      return hipSuccess;  // empty / event nodes: ordering only
  }
}

}  // namespace

// Builds the list of a captured graph.  max_lanes >= 1: streams a replay may use
// (1 = everything on the caller's stream, in capture order).  Returns a handle
// > 0, or a negative error: LD_EUNSUPPORTED for a graph with host / child-graph /
// allocation nodes, LD_EINVAL for a graph this runtime cannot describe.
extern "C" int64_t ld_step_list_build(void* hip_graph, int max_lanes) {
  hipGraph_t g = (hipGraph_t)hip_graph;
  if (!g || max_lanes < 1) return LD_EINVAL;
  size_t nn = 0;
  if (hipGraphGetNodes(g, nullptr, &nn) != hipSuccess || nn == 0) return LD_EINVAL;
  std::vector<hipGraphNode_t> hn(nn);
  if (hipGraphGetNodes(g, hn.data(), &nn) != hipSuccess) return LD_EINVAL;
  size_t ne = 0;
  if (hipGraphGetEdges(g, nullptr, nullptr, &ne) != hipSuccess) return LD_EINVAL;
  std::vector<hipGraphNode_t> ef(ne), et(ne);
  if (ne && hipGraphGetEdges(g, ef.data(), et.data(), &ne) != hipSuccess) return LD_EINVAL;
  std::unordered_map<hipGraphNode_t, int> index;
  for (size_t i = 0; i < nn; ++i) index[hn[i]] = (int)i;
  std::vector<std::vector<int>> preds(nn), succs(nn);
  for (size_t e = 0; e < ne; ++e) {
    auto a = index.find(ef[e]), b = index.find(et[e]);
    if (a == index.end() || b == index.end()) return LD_EINVAL;
    preds[b->second].push_back(a->second);
    succs[a->second].push_back(b->second);
  }
  // topological order that follows the capture order wherever it can (Kahn with
  // the smallest ready index first): the order the host issued the launches in
  std::vector<int> indeg(nn), order;
  std::vector<int> ready;
  for (size_t i = 0; i < nn; ++i) {
    indeg[i] = (int)preds[i].size();
    if (!indeg[i]) ready.push_back((int)i);
  }
  std::make_heap(ready.begin(), ready.end(), std::greater<int>());
  while (!ready.empty()) {
    std::pop_heap(ready.begin(), ready.end(), std::greater<int>());
    const int i = ready.back();
    ready.pop_back();
    order.push_back(i);
    for (int s : succs[i])
      if (--indeg[s] == 0) {
        ready.push_back(s);
        std::push_heap(ready.begin(), ready.end(), std::greater<int>());
      }
  }
  if (order.size() != nn) return LD_EINVAL;  // a cycle: not a graph
  std::vector<int> pos(nn);
  for (size_t k = 0; k < nn; ++k) pos[order[k]] = (int)k;

  StepList* s = new StepList();
  s->nodes.resize(nn);
  for (size_t k = 0; k < nn; ++k) {
    Node& n = s->nodes[k];
    const int i = order[k];
    if (hipGraphNodeGetType(hn[i], &n.type) != hipSuccess) { destroy(s); return LD_EINVAL; }
    hipError_t e = hipSuccess;
    switch (n.type) {
      case hipGraphNodeTypeKernel:
        e = hipGraphKernelNodeGetParams(hn[i], &n.kp);
        // arguments packed in `extra` (module-API launch): only hipModuleLaunchKernel
        // takes that form
        if (e == hipSuccess && !n.kp.kernelParams && n.kp.extra) n.module_fn = true;
        ++s->n_kernel;
        break;
      case hipGraphNodeTypeMemcpy:
        // a captured hipMemcpyAsync is a 1-D node: this runtime leaves the 3-D
        // descriptor unfilled for it (seen: kind 24661, zero extents).  Refuse
        // instead of re-issuing garbage; the step uses ld_copy_d2d (a kernel).
        n.cp = hipMemcpy3DParms{};
        e = hipGraphMemcpyNodeGetParams(hn[i], &n.cp);
        if (e == hipSuccess &&
            ((int)n.cp.kind < 0 || (int)n.cp.kind > 4 || n.cp.extent.width == 0 ||
             !n.cp.srcPtr.ptr || !n.cp.dstPtr.ptr)) {
          if (getenv("LD_STEP_LIST_DEBUG")) {
            // name the kernels around it: who issued this copy?
            auto name_of = [&](int j) -> const char* {
              hipGraphNodeType t;
              hipKernelNodeParams kp;
              if (hipGraphNodeGetType(hn[j], &t) != hipSuccess || t != hipGraphNodeTypeKernel ||
                  hipGraphKernelNodeGetParams(hn[j], &kp) != hipSuccess)
                return "(not a kernel)";
              const char* nm = hipKernelNameRefByPtr(kp.func, nullptr);
              return nm ? nm : "(unnamed)";
            };
            fprintf(stderr, "[ld_step_list] memcpy node %zu of %zu (capture index %d) has no usable "
                    "parameters\n", k, nn, i);
            for (int p : preds[i]) fprintf(stderr, "    after  %s\n", name_of(p));
            for (int q : succs[i]) fprintf(stderr, "    before %s\n", name_of(q));
          }
          destroy(s);
          return LD_EUNSUPPORTED;
        }
        ++s->n_memcpy;
        break;
      case hipGraphNodeTypeMemset:
        e = hipGraphMemsetNodeGetParams(hn[i], &n.ms);
        ++s->n_memset;
        break;
      case hipGraphNodeTypeEmpty:
      case hipGraphNodeTypeWaitEvent:
      case hipGraphNodeTypeEventRecord:
        ++s->n_noop;
        break;
      default:
        if (getenv("LD_STEP_LIST_DEBUG"))
          fprintf(stderr, "[ld_step_list] node %zu: unsupported type %d\n", k, (int)n.type);
        destroy(s);
        return LD_EUNSUPPORTED;
    }
    if (e != hipSuccess) { destroy(s); return LD_EINVAL; }
    for (int p : preds[i]) n.preds.push_back(pos[p]);
    std::sort(n.preds.begin(), n.preds.end());
  }
  // lanes: a node continues the lane of its OLDEST predecessor that is still its
  // lane's last node -- a weight gradient depends on the previous one (old) and on
  // the main-chain launch that produced its operand (just issued): it must stay
  // with the former, or the main chain's next launch would have to queue behind
  // it.  (At a join the main chain may thereby move to the side chain's lane; the
  // lanes are only names.)  With no such predecessor the node opens a lane, or --
  // all in use -- takes the one whose tail is oldest (an extra ordering edge).
  std::vector<int> tail;  // last node of each lane
  for (size_t k = 0; k < nn; ++k) {
    Node& n = s->nodes[k];
    int lane = -1;
    for (int p : n.preds) {  // ascending
      const int lp = s->nodes[p].lane;
      if (tail[lp] == p) {
        lane = lp;
        break;
      }
    }
    if (lane < 0) {
      if ((int)tail.size() < max_lanes) {
        lane = (int)tail.size();
        tail.push_back(-1);
      } else {
        lane = 0;
        for (int l = 1; l < (int)tail.size(); ++l)
          if (tail[l] < tail[lane]) lane = l;
      }
    }
    n.lane = lane;
    tail[lane] = (int)k;
  }
  s->nlanes = (int)tail.size();
  s->lane_tail = tail;
  for (size_t k = 0; k < nn; ++k) {
    Node& n = s->nodes[k];
    for (int p : n.preds)
      if (s->nodes[p].lane != n.lane) {
        // the newest predecessor per foreign lane is enough (stream order covers
        // the older ones)
        bool covered = false;
        for (int& w : n.waits)
          if (s->nodes[w].lane == s->nodes[p].lane) {
            w = std::max(w, p);
            covered = true;
          }
        if (!covered) n.waits.push_back(p);
      }
    for (int w : n.waits) s->nodes[w].record = true;
    s->n_cross += (int)n.waits.size();
  }
  for (Node& n : s->nodes)
    if (n.record &&
        hipEventCreateWithFlags(&n.event, hipEventDisableTiming) != hipSuccess) {
      destroy(s);
      return LD_EINVAL;
    }
  // LD_STEP_LIST_SIDE_PRIORITY=low: the lanes the list owns get the lowest stream
  // priority, so the caller's lane (the critical path) wins CUs when both have
  // work (A/B knob; measured in profiles/r06_step_list_lanes.txt)
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  const char* pe = getenv("LD_STEP_LIST_SIDE_PRIORITY");
  const bool low = pe && pe[0] == 'l';
  for (int l = 1; l < s->nlanes; ++l) {
    hipStream_t st;
    hipEvent_t ev;
    if ((low ? hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio_lo)
             : hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess ||
        hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      destroy(s);
      return LD_EINVAL;
    }
    s->side.push_back(st);
    s->join.push_back(ev);
  }
  if (hipEventCreateWithFlags(&s->fork, hipEventDisableTiming) != hipSuccess) {
    destroy(s);
    return LD_EINVAL;
  }
  std::lock_guard<std::mutex> lock(g_mu);
  const int64_t h = g_next++;
  g_lists[h] = s;
  return h;
}

// counts[8] = {kernel, memcpy, memset, ordering-only nodes, lanes, cross-lane
// waits, total nodes, 0}
extern "C" int ld_step_list_info(int64_t handle, int* counts) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_lists.find(handle);
  if (it == g_lists.end() || !counts) return LD_EINVAL;
  const StepList* s = it->second;
  const int v[8] = {s->n_kernel, s->n_memcpy, s->n_memset, s->n_noop,
                    s->nlanes,   s->n_cross,  (int)s->nodes.size(), 0};
  for (int i = 0; i < 8; ++i) counts[i] = v[i];
  return 0;
}

// Re-issues the step on `stream` (lane 0) and the list's own side streams: they
// start after everything enqueued on `stream` so far, and `stream` ends up behind
// all of them, so the replay is ordered like ONE operation of `stream`.  Only
// enqueues.  One replay at a time per list (its events are re-recorded).
extern "C" int ld_step_list_replay(int64_t handle, ld_stream_t stream) {
  StepList* s;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_lists.find(handle);
    if (it == g_lists.end()) return LD_EINVAL;
    s = it->second;
  }
  hipStream_t main = (hipStream_t)stream;
  if (s->nlanes > 1) {
    if (hipError_t e = hipEventRecord(s->fork, main)) return (int)e;
    for (hipStream_t st : s->side)
      if (hipError_t e = hipStreamWaitEvent(st, s->fork, 0)) return (int)e;
  }
  for (Node& n : s->nodes) {
    hipStream_t st = n.lane == 0 ? main : s->side[n.lane - 1];
    for (int w : n.waits)
      if (hipError_t e = hipStreamWaitEvent(st, s->nodes[w].event, 0)) return (int)e;
    if (hipError_t e = issue(n, st)) {
      g_fail[0] = (int)(&n - s->nodes.data());
      g_fail[1] = (int)n.type;
      g_fail[2] = (int)e;
      g_fail[3] = n.lane;
      if (n.type == hipGraphNodeTypeMemcpy) {
        g_fail[4] = (int)n.cp.kind;
        g_fail[5] = (int)n.cp.extent.width;
        g_fail[6] = (int)n.cp.extent.height;
        g_fail[7] = (int)n.cp.extent.depth;
      } else if (n.type == hipGraphNodeTypeMemset) {
        g_fail[4] = (int)n.ms.elementSize;
        g_fail[5] = (int)n.ms.width;
        g_fail[6] = (int)n.ms.height;
        g_fail[7] = (int)n.ms.pitch;
      } else if (n.type == hipGraphNodeTypeKernel) {
        g_fail[4] = (int)n.kp.gridDim.x;
        g_fail[5] = (int)n.kp.blockDim.x;
        g_fail[6] = (int)n.kp.sharedMemBytes;
        g_fail[7] = n.kp.kernelParams ? 1 : 0;
      }
      return (int)e;
    }
    if (n.record)
      if (hipError_t e = hipEventRecord(n.event, st)) return (int)e;
  }
  for (int l = 1; l < s->nlanes; ++l) {
    if (hipError_t e = hipEventRecord(s->join[l - 1], s->side[l - 1])) return (int)e;
    if (hipError_t e = hipStreamWaitEvent(main, s->join[l - 1], 0)) return (int)e;
  }
  return 0;
}

// Debugging aid: what the last failing ld_step_list_replay stopped at (node index,
// node type, hipError_t, lane, four type-specific details).
extern "C" int ld_step_list_last_failure(int* out8) {
  if (!out8) return LD_EINVAL;
  for (int i = 0; i < 8; ++i) out8[i] = g_fail[i];
  return 0;
}

extern "C" int ld_step_list_free(int64_t handle) {
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_lists.find(handle);
  if (it == g_lists.end()) return LD_EINVAL;
  destroy(it->second);
  g_lists.erase(it);
  return 0;
}
