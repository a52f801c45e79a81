// comm.cc -- the one collective of the path (dgcnn/trainval.py:64-73: gradients averaged over the towers), as a thin layer over
// RCCL: libdgcnn_hip.so does not link librccl -- it is dlopen'ed on first use, so single-GPU use needs no RCCL at all.
//   dgcnn_comm_unique_id  rank 0 draws the 128-byte id; the HOST side ships it to the other ranks (dgcnn/rccl.py: a socket on
//                         MASTER_ADDR:MASTER_PORT + 1 -- no torch.distributed anywhere)
//   dgcnn_comm_init       ncclCommInitRank (one process per GPU, the device already selected with hipSetDevice)
//   dgcnn_comm_info       ncclCommCount / ncclCommUserRank / ncclCommCuDevice: the communicator as RCCL sees it
//   dgcnn_allreduce_f32   in-place SUM all-reduce of a device buffer on the caller's stream (gradients: followed by a 1/world
//                         scale in the caller's Adam launch); dgcnn_broadcast_f32: rank `root`'s buffer to everyone
// One communicator serves several streams sequentially; the host orders the calls (a bucket per call).
#include "common.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

namespace {

typedef struct { char internal[128]; } UniqueId;           // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE
typedef void* Comm;                                        // ncclComm_t
enum { kFloat32 = 7, kSum = 0 };                           // ncclFloat32, ncclSum (rccl.h)

struct Api {
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(Comm, int*) = nullptr;
  int (*CommUserRank)(Comm, int*) = nullptr;
  int (*CommCuDevice)(Comm, int*) = nullptr;
} g;

int load_api() {
  if (g.lib) return DGCNN_OK;
  const char* names[] = {getenv("DGCNN_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* n : names) {
    if (n && *n && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  }
  if (!h) {
    dg::set_error("RCCL not found (librccl.so; set DGCNN_RCCL_LIB): %s", dlerror());
    return DGCNN_EUNSUP;
  }
#define DG_SYM(field, name)                                             \
  *(void**)(&g.field) = dlsym(h, name);                                 \
  if (!g.field) { dg::set_error("librccl: missing symbol %s", name); dlclose(h); return DGCNN_EUNSUP; }
  DG_SYM(GetUniqueId, "ncclGetUniqueId")
  DG_SYM(CommInitRank, "ncclCommInitRank")
  DG_SYM(CommDestroy, "ncclCommDestroy")
  DG_SYM(AllReduce, "ncclAllReduce")
  DG_SYM(Broadcast, "ncclBroadcast")
  DG_SYM(GetErrorString, "ncclGetErrorString")
  DG_SYM(CommCount, "ncclCommCount")
  DG_SYM(CommUserRank, "ncclCommUserRank")
  DG_SYM(CommCuDevice, "ncclCommCuDevice")
#undef DG_SYM
  g.lib = h;
  return DGCNN_OK;
}

// This RCCL build prints a banner ("RCCL version : ...", HIP / ROCm versions, host name, library path) to STDOUT when a
// communicator is initialised.  A library must not write to its host's stdout (bench.py's contract is ONE JSON line there):
// while RCCL initialises, file descriptor 1 points at stderr; C stdio is flushed on both sides of the switch.
struct StdoutToStderr {
  int saved;
  StdoutToStderr() {
    fflush(stdout);
    saved = dup(1);
    if (saved >= 0) dup2(2, 1);
  }
  ~StdoutToStderr() {
    fflush(stdout);
    if (saved >= 0) { dup2(saved, 1); close(saved); }
  }
};

// Lazy initialisation inside RCCL may print as well: the FIRST collective of every communicator is quiet too.  (fd 1 is process
// wide: a thread of the host that writes to stdout during those few calls lands on stderr -- the host of this library is the
// single-threaded Python driver; NCCL_DEBUG / RCCL_LOG do not silence the banner of this build.)
int64_t g_ar_calls = 0, g_ar_elems = 0;      // every all-reduce issued through this library (direct calls and plan replays)
void* g_warm[16];
int g_warm_n = 0;
bool quiet_first(void* comm) {
  for (int i = 0; i < g_warm_n; ++i)
    if (g_warm[i] == comm) return false;
  if (g_warm_n < 16) g_warm[g_warm_n++] = comm;
  return true;
}
void forget_comm(void* comm) {
  for (int i = 0; i < g_warm_n; ++i)
    if (g_warm[i] == comm) { g_warm[i] = g_warm[--g_warm_n]; return; }
}

int check(int rc, const char* what) {
  if (rc == 0) return DGCNN_OK;
  dg::set_error("%s: RCCL error %d (%s)", what, rc, g.GetErrorString ? g.GetErrorString(rc) : "?");
  return DGCNN_ELAUNCH;
}

}  // namespace

// dlopen librccl and resolve its entry points (every other dgcnn_comm_* call does this on first use; a separate entry point lets
// the host name the stage that failed)
extern "C" int dgcnn_comm_available(void) { return load_api(); }

extern "C" int dgcnn_comm_unique_id(void* id128) {
  DG_REQUIRE(id128, DGCNN_EINVAL, "dgcnn_comm_unique_id: null pointer");
  int rc = load_api();
  if (rc) return rc;
  UniqueId id;
  memset(&id, 0, sizeof(id));
  rc = check(g.GetUniqueId(&id), "ncclGetUniqueId");
  if (rc) return rc;
  memcpy(id128, &id, sizeof(id));
  return DGCNN_OK;
}

extern "C" int dgcnn_comm_init(int world, int rank, const void* id128, void** comm_out) {
  DG_REQUIRE(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, DGCNN_EINVAL, "dgcnn_comm_init: bad args");
  int rc = load_api();
  if (rc) return rc;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  Comm c = nullptr;
  {
    StdoutToStderr quiet;
    rc = check(g.CommInitRank(&c, world, id, rank), "ncclCommInitRank");
  }
  if (rc) return rc;
  *comm_out = c;
  return DGCNN_OK;
}

extern "C" int dgcnn_comm_destroy(void* comm) {
  if (!comm || !g.lib) return DGCNN_OK;
  forget_comm(comm);
  StdoutToStderr quiet;
  return check(g.CommDestroy((Comm)comm), "ncclCommDestroy");
}

namespace dg {
void plan_add_allreduce(float* buf, int64_t count, void* comm, hipStream_t st);      // plan.cc

// the collective itself (also what a replayed launch plan calls)
int comm_allreduce(float* buf, int64_t count, void* comm, hipStream_t st) {
  g_ar_calls += 1;
  g_ar_elems += count;
  if (quiet_first(comm)) {
    StdoutToStderr quiet;
    return check(g.AllReduce(buf, buf, (size_t)count, kFloat32, kSum, (Comm)comm, st), "ncclAllReduce");
  }
  return check(g.AllReduce(buf, buf, (size_t)count, kFloat32, kSum, (Comm)comm, st), "ncclAllReduce");
}
}  // namespace dg

extern "C" int dgcnn_allreduce_f32(float* buf, int64_t count, void* comm, void* stream) {
  DG_REQUIRE(buf && comm && count > 0 && g.lib, DGCNN_EINVAL, "dgcnn_allreduce_f32: bad args (communicator from dgcnn_comm_init)");
  const int rc = dg::comm_allreduce(buf, count, comm, (hipStream_t)stream);
  if (rc == DGCNN_OK) dg::plan_add_allreduce(buf, count, comm, (hipStream_t)stream);
  return rc;
}

extern "C" int dgcnn_broadcast_f32(float* buf, int64_t count, int root, void* comm, void* stream) {
  DG_REQUIRE(buf && comm && count > 0 && root >= 0 && g.lib, DGCNN_EINVAL, "dgcnn_broadcast_f32: bad args");
  if (quiet_first(comm)) {
    StdoutToStderr quiet;
    return check(g.Broadcast(buf, buf, (size_t)count, kFloat32, root, (Comm)comm, (hipStream_t)stream), "ncclBroadcast");
  }
  return check(g.Broadcast(buf, buf, (size_t)count, kFloat32, root, (Comm)comm, (hipStream_t)stream), "ncclBroadcast");
}

// What RCCL ITSELF says about the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) -- not what the launcher
// claimed: bench.py reports these as config.rccl_ranks / rccl_rank, the evidence that a line at --gpus N ran on N ranks.
extern "C" int dgcnn_comm_info(void* comm, int* nranks, int* rank, int* device) {
  DG_REQUIRE(comm && nranks && rank && device && g.lib, DGCNN_EINVAL, "dgcnn_comm_info: bad args (communicator from dgcnn_comm_init)");
  int rc = check(g.CommCount((Comm)comm, nranks), "ncclCommCount");
  if (rc) return rc;
  rc = check(g.CommUserRank((Comm)comm, rank), "ncclCommUserRank");
  if (rc) return rc;
  return check(g.CommCuDevice((Comm)comm, device), "ncclCommCuDevice");
}

// All-reduce calls / elements issued through this library so far, replayed launch plans included (a replay issues its recorded
// collectives from C: the host-side wrappers never see them) -- tests and bench.py count the collectives of a step with it.
extern "C" int dgcnn_comm_counters(int64_t* allreduce_calls, int64_t* allreduce_elems) {
  if (allreduce_calls) *allreduce_calls = g_ar_calls;
  if (allreduce_elems) *allreduce_elems = g_ar_elems;
  return DGCNN_OK;
}
