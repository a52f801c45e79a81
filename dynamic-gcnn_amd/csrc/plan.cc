// plan.cc -- launch plans: record the launches of one step once, re-issue them from a single C loop afterwards.
//
// The reference builds a static TF graph and replays it with sess.run (dgcnn/trainval.py:103-129).  The eager path here issues
// ~140 kernels per training step from Python (12-30 us of host time each): enough for the host to fall behind the GPU whenever
// something else keeps it busy (BENCH_r04: a 4.75 ms step measured as 6.69 ms with 5.4 ms of host enqueue).  A HIP graph removes the
// host cost but turns the second stream of the step into graph branches, which this runtime schedules 2-5 % slower than the
// same two streams issued eagerly, and it cannot contain an RCCL call.  A plan keeps plain stream semantics instead:
//   * dgcnn_plan_begin(): from here on every dg::launch / dg::memset_async / dg::stream_wait / all-reduce of the library is
//     executed as usual AND appended to the plan under construction (kernel address, geometry, a private copy of the argument
//     bytes, stream handles);
//   * dgcnn_plan_end(&plan); dgcnn_plan_replay(plan): one loop of hipLaunchKernel / hipMemsetAsync / hipEventRecord +
//     hipStreamWaitEvent / ncclAllReduce calls on the recorded streams -- the GPU sees exactly the eager schedule.
// The caller guarantees what a captured graph needs as well: the same device addresses on every replay (dgcnn/trainval.py records
// inside a private torch memory pool that outlives the plan) and no host-side per-step values among the arguments (the dropout
// seed lives in device memory; Adam stays outside).  One recorder per process (the Python host is single threaded).
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <vector>

namespace dg {
int comm_allreduce(float* buf, int64_t count, void* comm, hipStream_t st);    // comm.cc
}

namespace {

enum Kind { K_KERNEL = 0, K_MEMSET = 1, K_WAIT = 2, K_ALLREDUCE = 3 };

struct Node {
  int kind;
  const void* fn;          // kernel
  dim3 g, b;
  unsigned sh;
  hipStream_t st;          // kernel / memset / collective stream; the WAITER of a wait
  hipStream_t st2;         // the signaller of a wait
  size_t arg0, nargs;      // slice of Plan::argv
  void* ptr;               // memset target / all-reduce buffer
  int value;
  size_t bytes;            // memset bytes / all-reduce element count
  void* comm;
  hipEvent_t ev;           // wait: owned by the node
};

struct Plan {
  std::vector<Node> nodes;
  std::vector<void*> argv;                           // argument pointers of all kernel nodes, into `store`
  std::vector<std::unique_ptr<char[]>> store;        // 16-byte aligned copies of the argument values
  int counts[4] = {0, 0, 0, 0};
  ~Plan() {
    for (Node& n : nodes)
      if (n.kind == K_WAIT && n.ev) (void)hipEventDestroy(n.ev);
  }
};

Plan* g_rec = nullptr;

// events of eager (unrecorded) cross-stream waits: hipStreamWaitEvent latches the record that precedes it, so an event can be
// recorded again as soon as the wait call has returned; a small ring only spreads the calls over several objects
constexpr int EVRING = 16;
hipEvent_t g_ring[EVRING];
int g_ring_n = 0, g_ring_i = 0;

// Events of cross-stream waits order kernels of ONE device and are never looked at by the host (or by another device: RCCL's kernels
// fence their own transfers), so the system-scope fence of a record is skipped: 4.45 -> 4.43 ms per step at configs[1] with 15 waits
// (hipEventReleaseToDevice: no difference).  DGCNN_EVENT_FLAGS overrides (A/B switch).
unsigned event_flags() {
  static long f = -1;
  if (f < 0) {
    const char* e = getenv("DGCNN_EVENT_FLAGS");
    f = e ? (long)strtoul(e, nullptr, 0) : (long)(hipEventDisableTiming | hipEventDisableSystemFence);
  }
  return (unsigned)f;
}

hipEvent_t ring_event() {
  if (g_ring_n < EVRING) {
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, event_flags()) != hipSuccess) return nullptr;
    g_ring[g_ring_n++] = e;
    return e;
  }
  g_ring_i = (g_ring_i + 1) % EVRING;
  return g_ring[g_ring_i];
}

}  // namespace

namespace dg {

bool plan_recording() { return g_rec != nullptr; }

void plan_add_kernel(const void* fn, dim3 g, dim3 b, size_t sh, hipStream_t st, void** argv, const size_t* sizes, int n) {
  Plan& p = *g_rec;
  size_t total = 0;
  for (int i = 0; i < n; ++i) total += (sizes[i] + 15) & ~(size_t)15;
  std::unique_ptr<char[]> buf(new char[total + 16]);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(buf.get()) + 15) & ~(uintptr_t)15);
  Node nd;
  memset(&nd, 0, sizeof(nd));
  nd.kind = K_KERNEL;
  nd.fn = fn;
  nd.g = g;
  nd.b = b;
  nd.sh = (unsigned)sh;
  nd.st = st;
  nd.arg0 = p.argv.size();
  nd.nargs = (size_t)n;
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    memcpy(base + off, argv[i], sizes[i]);
    p.argv.push_back(base + off);
    off += (sizes[i] + 15) & ~(size_t)15;
  }
  p.store.push_back(std::move(buf));
  p.nodes.push_back(nd);
  p.counts[K_KERNEL]++;
}

int memset_async(void* ptr, int value, size_t bytes, hipStream_t st) {
  const hipError_t e = hipMemsetAsync(ptr, value, bytes, st);
  if (g_rec) {
    Node nd;
    memset(&nd, 0, sizeof(nd));
    nd.kind = K_MEMSET;
    nd.st = st;
    nd.ptr = ptr;
    nd.value = value;
    nd.bytes = bytes;
    g_rec->nodes.push_back(nd);
    g_rec->counts[K_MEMSET]++;
  }
  return e == hipSuccess ? DGCNN_OK : DGCNN_ELAUNCH;
}

int stream_wait(hipStream_t waiter, hipStream_t signaller) {
  hipEvent_t ev = nullptr;
  if (g_rec) {
    if (hipEventCreateWithFlags(&ev, event_flags()) != hipSuccess) return DGCNN_ELAUNCH;
    Node nd;
    memset(&nd, 0, sizeof(nd));
    nd.kind = K_WAIT;
    nd.st = waiter;
    nd.st2 = signaller;
    nd.ev = ev;
    g_rec->nodes.push_back(nd);
    g_rec->counts[K_WAIT]++;
  } else {
    ev = ring_event();
    if (!ev) return DGCNN_ELAUNCH;
  }
  if (hipEventRecord(ev, signaller) != hipSuccess) return DGCNN_ELAUNCH;
  if (hipStreamWaitEvent(waiter, ev, 0) != hipSuccess) return DGCNN_ELAUNCH;
  return DGCNN_OK;
}

void plan_add_allreduce(float* buf, int64_t count, void* comm, hipStream_t st) {
  if (!g_rec) return;
  Node nd;
  memset(&nd, 0, sizeof(nd));
  nd.kind = K_ALLREDUCE;
  nd.st = st;
  nd.ptr = buf;
  nd.bytes = (size_t)count;
  nd.comm = comm;
  g_rec->nodes.push_back(nd);
  g_rec->counts[K_ALLREDUCE]++;
}

}  // namespace dg

extern "C" int dgcnn_plan_begin(void) {
  DG_REQUIRE(!g_rec, DGCNN_EINVAL, "dgcnn_plan_begin: a plan is already being recorded");
  (void)hipGetLastError();
  g_rec = new Plan();
  return DGCNN_OK;
}

extern "C" int dgcnn_plan_end(void** plan_out) {
  DG_REQUIRE(g_rec && plan_out, DGCNN_EINVAL, "dgcnn_plan_end: no plan is being recorded");
  *plan_out = g_rec;
  g_rec = nullptr;
  return dg::check_launch("dgcnn_plan_end");
}

extern "C" int dgcnn_plan_abort(void) {
  delete g_rec;
  g_rec = nullptr;
  return DGCNN_OK;
}

extern "C" int dgcnn_plan_replay(void* plan) {
  DG_REQUIRE(plan && !g_rec, DGCNN_EINVAL, "dgcnn_plan_replay: null plan, or called while recording");
  Plan& p = *reinterpret_cast<Plan*>(plan);
  for (const Node& n : p.nodes) {
    hipError_t e = hipSuccess;
    switch (n.kind) {
      case K_KERNEL:
        e = hipLaunchKernel(n.fn, n.g, n.b, p.argv.data() + n.arg0, n.sh, n.st);
        break;
      case K_MEMSET:
        e = hipMemsetAsync(n.ptr, n.value, n.bytes, n.st);
        break;
      case K_WAIT:
        e = hipEventRecord(n.ev, n.st2);
        if (e == hipSuccess) e = hipStreamWaitEvent(n.st, n.ev, 0);
        break;
      case K_ALLREDUCE: {
        const int rc = dg::comm_allreduce(reinterpret_cast<float*>(n.ptr), (int64_t)n.bytes, n.comm, n.st);
        if (rc) return rc;
        break;
      }
    }
    if (e != hipSuccess) {
      dg::set_error("dgcnn_plan_replay: %s", hipGetErrorString(e));
      return DGCNN_ELAUNCH;
    }
  }
  return DGCNN_OK;
}

extern "C" int dgcnn_plan_info(void* plan, int* kernels, int* memsets, int* waits, int* collectives) {
  DG_REQUIRE(plan, DGCNN_EINVAL, "dgcnn_plan_info: null plan");
  const Plan& p = *reinterpret_cast<Plan*>(plan);
  if (kernels) *kernels = p.counts[K_KERNEL];
  if (memsets) *memsets = p.counts[K_MEMSET];
  if (waits) *waits = p.counts[K_WAIT];
  if (collectives) *collectives = p.counts[K_ALLREDUCE];
  return DGCNN_OK;
}

extern "C" int dgcnn_plan_destroy(void* plan) {
  delete reinterpret_cast<Plan*>(plan);
  return DGCNN_OK;
}

extern "C" int dgcnn_stream_wait(void* waiter, void* signaller) {
  const int rc = dg::stream_wait((hipStream_t)waiter, (hipStream_t)signaller);
  if (rc) dg::set_error("dgcnn_stream_wait: %s", hipGetErrorString(hipGetLastError()));
  return rc;
}

extern "C" int dgcnn_memset_async(void* ptr, int value, size_t bytes, void* stream) {
  DG_REQUIRE(ptr && bytes > 0, DGCNN_EINVAL, "dgcnn_memset_async: bad args");
  const int rc = dg::memset_async(ptr, value, bytes, (hipStream_t)stream);
  if (rc) dg::set_error("dgcnn_memset_async: %s", hipGetErrorString(hipGetLastError()));
  return rc;
}

// ---- the side stream (weight-gradient GEMMs, transposed adjacency, parameter-only preparation): nothing on the critical path waits
// for it before the optimizer, so it is created at the LOWEST priority the device offers and, optionally, on a CU mask that leaves
// `reserve_cus` compute units (spread evenly over the XCDs: the mask's bits are interleaved across them) to the main stream alone --
// a 256-workgroup GEMM with all of a CU's LDS otherwise makes a 4-workgroup BatchNorm finalize of the main stream wait for one of its
// workgroups to retire (profiles/r06/timeline_*.txt: 6 us alone, 66-113 us beside a weight-gradient GEMM).
extern "C" int dgcnn_stream_create(int low_priority, int reserve_cus, void** stream_out) {
  DG_REQUIRE(stream_out, DGCNN_EINVAL, "dgcnn_stream_create: null output");
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);           // numerically: least >= greatest (lower number = higher priority)
  const int prio = low_priority ? least : 0;
  hipStream_t st = nullptr;
  hipError_t e;
  if (reserve_cus > 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev);
    e = hipGetDeviceProperties(&prop, dev);
    DG_REQUIRE(e == hipSuccess, DGCNN_ELAUNCH, "dgcnn_stream_create: %s", hipGetErrorString(e));
    const int ncu = prop.multiProcessorCount;
    DG_REQUIRE(reserve_cus < ncu, DGCNN_EINVAL, "dgcnn_stream_create: cannot reserve %d of %d CUs", reserve_cus, ncu);
    const int words = (ncu + 31) / 32;
    uint32_t mask[32] = {0};
    DG_REQUIRE(words <= 32, DGCNN_EUNSUP, "dgcnn_stream_create: %d CUs", ncu);
    for (int i = 0; i < ncu - reserve_cus; ++i) mask[i / 32] |= 1u << (i % 32);   // the LAST reserve_cus bits stay clear
    e = hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask);
    DG_REQUIRE(e == hipSuccess, DGCNN_ELAUNCH, "dgcnn_stream_create: hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    (void)prio;                                                        // (the CU-mask entry point takes no priority)
  } else {
    e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio);
    DG_REQUIRE(e == hipSuccess, DGCNN_ELAUNCH, "dgcnn_stream_create: hipStreamCreateWithPriority(%d): %s", prio, hipGetErrorString(e));
  }
  *stream_out = (void*)st;
  return DGCNN_OK;
}

extern "C" int dgcnn_stream_destroy(void* stream) {
  if (stream) (void)hipStreamDestroy((hipStream_t)stream);
  return DGCNN_OK;
}

extern "C" int dgcnn_stream_priority_range(int* least, int* greatest) {
  DG_REQUIRE(least && greatest, DGCNN_EINVAL, "dgcnn_stream_priority_range: null output");
  hipError_t e = hipDeviceGetStreamPriorityRange(least, greatest);
  DG_REQUIRE(e == hipSuccess, DGCNN_ELAUNCH, "dgcnn_stream_priority_range: %s", hipGetErrorString(e));
  return DGCNN_OK;
}
