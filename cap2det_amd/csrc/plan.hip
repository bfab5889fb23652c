// Step plan: the launch list of one training step, recorded once and replayed by ONE C-ABI call.
//
// The reference hands its whole step to the TensorFlow runtime in one session.run
// (train/trainer.py:141-146, slim.learning.create_train_op); here a step is ~140 C-ABI calls on four
// streams with ~40 event records / waits between them, which a Python loop queued one ctypes call
// at a time (2.2 ms of host time per 2.9-ms bf16 step).  A plan holds that sequence —
// (entry point, argument words, stream) and (event record | stream wait) nodes in issue order — and
// c2d_plan_replay walks it natively: the same entry points run their own host-side dispatch, so a
// replayed step launches bit for bit what the recorded step launched.  What may change between
// replays is declared when a node is added: pointer arguments bound to a slot (the step's input
// tensors: base pointer of the slot + the recorded offset) and scalar arguments bound to a slot
// (dropout key, learning rate).
//
// Entry points are reached through typed thunks generated from the header (plan_thunks.inc).
#include "c2d_common.h"
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct PlanThunk { const char* name; int (*fn)(const long long*); int nargs; };
static inline float plan_f32(long long w) { const unsigned u = (unsigned)w; float f; memcpy(&f, &u, 4); return f; }
#include "plan_thunks.inc"

constexpr int PLAN_MAX_ARGS = 40;
enum { ARG_CONST = 0, ARG_PTR_SLOT = 1, ARG_WORD_SLOT = 2 };
enum { NODE_CALL = 0, NODE_RECORD = 1, NODE_WAIT = 2 };

struct Node {
  int kind;
  int (*fn)(const long long*);
  int nargs, nbound;
  long long vals[PLAN_MAX_ARGS];        // constants; for ARG_PTR_SLOT the byte offset from the slot's base
  unsigned char kinds[PLAN_MAX_ARGS];
  short slots[PLAN_MAX_ARGS];
  hipStream_t stream;
  int event;
};

struct Plan {
  std::vector<Node> nodes;
  std::vector<hipEvent_t> events;
  int num_slots = 0;
  bool finished = false;
};

const std::unordered_map<std::string, const PlanThunk*>& thunk_map() {
  static const std::unordered_map<std::string, const PlanThunk*> m = [] {
    std::unordered_map<std::string, const PlanThunk*> t;
    for (const PlanThunk& k : kPlanThunks) t[k.name] = &k;
    return t;
  }();
  return m;
}

int ensure_event(Plan* p, int idx) {
  if (idx < 0 || idx > 4096) return C2D_ERR_INVALID_ARG;
  while ((int)p->events.size() <= idx) p->events.push_back(nullptr);
  // (device-side ordering between the plan's own streams only — nobody inspects these events from the
  //  host: no system-scope fence, i.e. no cache writeback / invalidate at each of a step's ~40 records)
  if (!p->events[idx] &&
      hipEventCreateWithFlags(&p->events[idx], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess)
    return C2D_ERR_LAUNCH;
  return C2D_OK;
}

}  // namespace

// Device-to-device copy on a stream (the look-ahead hand-over of a planned step: FrcnnEngine.forward).
// (a kernel, not hipMemcpyAsync: a device-to-device hipMemcpyAsync between the kernels of a stream
//  costs tens of microseconds of pipeline bubble, the 2.4 MB copy itself two)
__global__ __launch_bounds__(256) void copy16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                     long long n16) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}
extern "C" int c2d_copy_bytes(const void* src, void* dst, long long bytes, void* stream) {
  C2D_CHECK_ARG(src && dst && bytes > 0 && bytes % 16 == 0);
  C2D_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0);
  const long long n16 = bytes / 16;
  hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)src, (uint4*)dst, n16);
  return c2d_launch_status();
}

// Rehearsal of CUs lost to a collective (tools/cu_withhold.py): `workgroups` blocks of 1024 threads
// with the whole LDS of a CU each, asleep until `microseconds` have passed on the 100-MHz counter.
__global__ __launch_bounds__(1024) void hold_cus_kernel(long long ticks, int* sink) {
  extern __shared__ int lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) lds[0] = 1;
  while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(64);
  if (sink && threadIdx.x == 0 && lds[0] == 2) *sink = 1;
}
extern "C" int c2d_debug_hold_cus(int workgroups, long long microseconds, void* stream) {
  C2D_CHECK_ARG(workgroups >= 0 && workgroups <= 256 && microseconds >= 0 && microseconds <= 5000000);
  if (workgroups == 0) return C2D_OK;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)hold_cus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return C2D_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(hold_cus_kernel, dim3(workgroups), dim3(1024), 160 * 1024, (hipStream_t)stream,
                     microseconds * 100, (int*)nullptr);
  return c2d_launch_status();
}

extern "C" long long c2d_plan_create(void) { return (long long)(uintptr_t) new Plan(); }

extern "C" int c2d_plan_destroy(void* plan) {
  Plan* p = (Plan*)plan;
  if (!p) return C2D_OK;
  for (hipEvent_t e : p->events)
    if (e) (void)hipEventDestroy(e);
  delete p;
  return C2D_OK;
}

extern "C" int c2d_plan_add_call(void* plan, const char* name, int nargs, const long long* vals,
                                 const uint8_t* kinds, const int* slots) {
  Plan* p = (Plan*)plan;
  C2D_CHECK_ARG(p && name && !p->finished && nargs >= 0 && nargs <= PLAN_MAX_ARGS);
  C2D_CHECK_ARG(nargs == 0 || (vals && kinds && slots));
  const auto it = thunk_map().find(name);
  if (it == thunk_map().end()) return C2D_ERR_UNSUPPORTED;
  C2D_CHECK_ARG(it->second->nargs == nargs);
  Node n = {};
  n.kind = NODE_CALL; n.fn = it->second->fn; n.nargs = nargs;
  for (int i = 0; i < nargs; ++i) {
    C2D_CHECK_ARG(kinds[i] <= ARG_WORD_SLOT);
    n.vals[i] = vals[i]; n.kinds[i] = kinds[i]; n.slots[i] = (short)slots[i];
    if (kinds[i] != ARG_CONST) {
      C2D_CHECK_ARG(slots[i] >= 0 && slots[i] < 1024);
      ++n.nbound;
      if (slots[i] + 1 > p->num_slots) p->num_slots = slots[i] + 1;
    }
  }
  p->nodes.push_back(n);
  return C2D_OK;
}

extern "C" int c2d_plan_add_event_record(void* plan, int event, void* stream) {
  Plan* p = (Plan*)plan;
  C2D_CHECK_ARG(p && !p->finished);
  const int rc = ensure_event(p, event);
  if (rc) return rc;
  Node n = {};
  n.kind = NODE_RECORD; n.event = event; n.stream = (hipStream_t)stream;
  p->nodes.push_back(n);
  return C2D_OK;
}

extern "C" int c2d_plan_add_stream_wait(void* plan, void* stream, int event) {
  Plan* p = (Plan*)plan;
  C2D_CHECK_ARG(p && !p->finished && event >= 0 && event < (int)p->events.size() && p->events[event]);
  Node n = {};
  n.kind = NODE_WAIT; n.event = event; n.stream = (hipStream_t)stream;
  p->nodes.push_back(n);
  return C2D_OK;
}

// Closes the plan.  The eager schedule's cross-step events (look-ahead hand-over, scratch rotation)
// are not recorded; instead
//   * every other stream of the plan records a plan event when it has been issued all its work of a
//     replay, and the main stream waits for those events at the START of the next replay (where the
//     eager step waits for the look-ahead) — never at the end of a step, where a barrier on the main
//     stream is a bubble;
//   * a stream whose first node is not already a wait for an event the main stream recorded in
//     this plan (the eager forks are) starts behind the main stream's position at the replay call.
// A replay therefore depends on nothing but the previous replay of the same plan (or joined
// streams: the caller joins them after an eager step).
extern "C" int c2d_plan_finish(void* plan, void* main_stream) {
  Plan* p = (Plan*)plan;
  C2D_CHECK_ARG(p && !p->finished);
  const hipStream_t mainst = (hipStream_t)main_stream;
  std::vector<hipStream_t> others;
  for (const Node& n : p->nodes)
    if (n.kind != NODE_CALL && n.stream != mainst && std::find(others.begin(), others.end(), n.stream) == others.end())
      others.push_back(n.stream);
  if (!others.empty()) {
    std::vector<Node> head, tail;
    int e0 = -1;
    for (hipStream_t s : others) {
      // does the stream start with a wait for an event of the main stream?
      bool forked = false;
      for (size_t i = 0; i < p->nodes.size(); ++i) {
        const Node& n = p->nodes[i];
        if (n.kind == NODE_CALL || n.stream != s) continue;
        if (n.kind == NODE_WAIT)
          for (size_t j = 0; j < i; ++j)
            if (p->nodes[j].kind == NODE_RECORD && p->nodes[j].event == n.event && p->nodes[j].stream == mainst)
              forked = true;
        break;
      }
      if (!forked) {
        if (e0 < 0) {
          e0 = (int)p->events.size();
          const int rc = ensure_event(p, e0);
          if (rc) return rc;
          Node r = {}; r.kind = NODE_RECORD; r.event = e0; r.stream = mainst;
          head.insert(head.begin(), r);
        }
        Node w = {}; w.kind = NODE_WAIT; w.event = e0; w.stream = s;
        head.push_back(w);
      }
      // a stream whose last node is a record that the main stream waits for later in the plan is
      // joined inside the step (the filter-gradient and branch streams are): nothing to carry over
      bool joined = false;
      for (size_t i = p->nodes.size(); i-- > 0;) {
        const Node& n = p->nodes[i];
        if (n.kind == NODE_CALL || n.stream != s) continue;
        if (n.kind == NODE_RECORD)
          for (size_t j = i + 1; j < p->nodes.size(); ++j)
            if (p->nodes[j].kind == NODE_WAIT && p->nodes[j].event == n.event && p->nodes[j].stream == mainst)
              joined = true;
        break;
      }
      if (joined) continue;
      const int ei = (int)p->events.size();
      const int rc = ensure_event(p, ei);
      if (rc) return rc;
      Node ww = {}; ww.kind = NODE_WAIT; ww.event = ei; ww.stream = mainst;     // (the previous replay's)
      head.push_back(ww);
      Node rr = {}; rr.kind = NODE_RECORD; rr.event = ei; rr.stream = s;
      tail.push_back(rr);
    }
    p->nodes.insert(p->nodes.begin(), head.begin(), head.end());
    p->nodes.insert(p->nodes.end(), tail.begin(), tail.end());
  }
  p->finished = true;
  return C2D_OK;
}

extern "C" int c2d_plan_size(void* plan) {
  Plan* p = (Plan*)plan;
  return p ? (int)p->nodes.size() : C2D_ERR_INVALID_ARG;
}

extern "C" int c2d_plan_replay(void* plan, const long long* bindings, int num_bindings,
                               int* failed_node) {
  Plan* p = (Plan*)plan;
  C2D_CHECK_ARG(p && p->finished && num_bindings >= p->num_slots && (num_bindings == 0 || bindings));
  long long v[PLAN_MAX_ARGS];
  const int count = (int)p->nodes.size();
  for (int i = 0; i < count; ++i) {
    const Node& n = p->nodes[i];
    int rc = C2D_OK;
    if (n.kind == NODE_CALL) {
      const long long* args = n.vals;
      if (n.nbound) {
        for (int a = 0; a < n.nargs; ++a)
          v[a] = n.kinds[a] == ARG_CONST ? n.vals[a]
                 : n.kinds[a] == ARG_PTR_SLOT ? bindings[n.slots[a]] + n.vals[a] : bindings[n.slots[a]];
        args = v;
      }
      rc = n.fn(args);
    } else if (n.kind == NODE_RECORD) {
      rc = hipEventRecord(p->events[n.event], n.stream) == hipSuccess ? C2D_OK : C2D_ERR_LAUNCH;
    } else {
      rc = hipStreamWaitEvent(n.stream, p->events[n.event], 0) == hipSuccess ? C2D_OK : C2D_ERR_LAUNCH;
    }
    if (rc) {
      if (failed_node) *failed_node = i;
      return rc;
    }
  }
  return C2D_OK;
}
