// hero_comm_*: the gradient exchange of the data-parallel step over RCCL, behind the C ABI.
//
//   replaces  all_reduce_and_rescale_tensors / broadcast_tensors        utils/distributed.py:19-46, 103-151 (Horovod)
//             the all-gather of the cross-GPU negatives                 model/pretrain.py:427-451
//
// Every call ENQUEUES on the caller's HIP stream and returns: no host synchronisation, no stream of its own, no thread.
// That is the difference to going through torch.distributed's ProcessGroupNCCL (hero_amd/utils/distributed.py, the
// default): there the collectives run on the process group's internal stream and its watchdog thread polls events, which
// is why a hipGraph capture of the data-parallel step needs capture_error_mode="thread_local" and a watchdog around it;
// a collective enqueued here is just another node of the capturing stream.
//
// librccl is opened lazily (dlopen) by the first hero_comm_* call: libhero_hip.so itself does not link it, so the
// kernels load and run on a box without RCCL, and `hero_comm_available()` says whether the exchange can.
// One communicator per process (one process per GPU); the 128-byte unique id comes from rank 0
// (hero_comm_unique_id) and reaches the other ranks by whatever the host side already has (hero_amd.utils.comm
// broadcasts it over the existing torch.distributed group; a launcher could use a file or an environment variable).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "common.h"

namespace hero {
namespace {

typedef struct { char internal[128]; } UniqueId;          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;                                       // ncclComm_t
enum { kSum = 0, kInt8 = 0, kFloat32 = 7, kBfloat16 = 9 };  // ncclSum, ncclInt8, ncclFloat32, ncclBfloat16 (rccl.h)

struct Api {
  void* so = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Api& api() {
  static Api a = [] {
    Api r;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.so) break;
    }
    if (!r.so) return r;
#define HERO_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.so, sym))
    HERO_SYM(GetUniqueId, "ncclGetUniqueId");
    HERO_SYM(CommInitRank, "ncclCommInitRank");
    HERO_SYM(CommDestroy, "ncclCommDestroy");
    HERO_SYM(AllReduce, "ncclAllReduce");
    HERO_SYM(Broadcast, "ncclBroadcast");
    HERO_SYM(AllGather, "ncclAllGather");
    HERO_SYM(GroupStart, "ncclGroupStart");
    HERO_SYM(GroupEnd, "ncclGroupEnd");
    HERO_SYM(GetErrorString, "ncclGetErrorString");
#undef HERO_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.Broadcast && r.AllGather && r.GroupStart && r.GroupEnd;
    return r;
  }();
  return a;
}

struct State { Comm comm; int rank, world; };

int fail(const char* what, int rc) {
  const Api& a = api();
  set_error("%s: RCCL error %d (%s)", what, rc, a.GetErrorString ? a.GetErrorString(rc) : "?");
  return HERO_ERR_LAUNCH;
}

}  // namespace
}  // namespace hero

using namespace hero;

#define HERO_COMM_READY(h)                                                                                  \
  HERO_REQUIRE(api().ok, "hero_comm: librccl.so could not be opened (RCCL is not installed on this box)"); \
  HERO_REQUIRE((h) != nullptr, "hero_comm: null communicator")

extern "C" int hero_comm_available(void) { return api().ok ? 1 : 0; }

extern "C" int hero_comm_unique_id(void* id128) {
  HERO_REQUIRE(api().ok, "hero_comm_unique_id: librccl.so could not be opened");
  HERO_REQUIRE(id128, "hero_comm_unique_id: null pointer");
  UniqueId id;
  const int rc = api().GetUniqueId(&id);
  if (rc) return fail("hero_comm_unique_id", rc);
  memcpy(id128, id.internal, sizeof(id.internal));
  return HERO_OK;
}

// Collective over all `world` ranks (blocks until every rank has called it); uses the current HIP device.
extern "C" int hero_comm_init(const void* id128, int rank, int world, void** comm_out) {
  HERO_REQUIRE(api().ok, "hero_comm_init: librccl.so could not be opened");
  HERO_REQUIRE(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, "hero_comm_init: bad arguments (rank %d of %d)", rank, world);
  UniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  Comm c = nullptr;
  const int rc = api().CommInitRank(&c, world, id, rank);
  if (rc) return fail("hero_comm_init", rc);
  *comm_out = new State{c, rank, world};
  return HERO_OK;
}

extern "C" int hero_comm_destroy(void* comm) {
  if (!comm) return HERO_OK;
  State* s = static_cast<State*>(comm);
  const int rc = api().ok ? api().CommDestroy(s->comm) : 0;
  delete s;
  return rc ? fail("hero_comm_destroy", rc) : HERO_OK;
}

extern "C" int hero_comm_rank(void* comm) { return comm ? static_cast<State*>(comm)->rank : -1; }
extern "C" int hero_comm_world(void* comm) { return comm ? static_cast<State*>(comm)->world : 0; }

// In-place SUM all-reduce of up to 64 gradient buckets as ONE group (one launch on the wire side); the 1 / world of the
// reference's average (Horovod's default, utils/distributed.py:38-39) is folded into the optimiser kernel, not done here.
extern "C" int hero_comm_allreduce_buckets(void* comm, const HeroCommBucket* b, int n, hero_stream_t stream) {
  HERO_COMM_READY(comm);
  HERO_REQUIRE(b && n >= 1 && n <= 64, "hero_comm_allreduce_buckets: 1..64 buckets");
  State* s = static_cast<State*>(comm);
  for (int i = 0; i < n; ++i)
    HERO_REQUIRE(b[i].buf && b[i].count > 0 && (b[i].dtype == HERO_F32 || b[i].dtype == HERO_BF16), "hero_comm_allreduce_buckets: bad bucket %d", i);
  int rc = api().GroupStart();
  if (rc) return fail("hero_comm_allreduce_buckets(group)", rc);
  for (int i = 0; i < n && !rc; ++i)
    rc = api().AllReduce(b[i].buf, b[i].buf, b[i].count, b[i].dtype == HERO_BF16 ? kBfloat16 : kFloat32, kSum, s->comm, static_cast<hipStream_t>(stream));
  const int rc2 = api().GroupEnd();
  if (rc || rc2) return fail("hero_comm_allreduce_buckets", rc ? rc : rc2);
  return HERO_OK;
}

extern "C" int hero_comm_broadcast(void* comm, void* buf, size_t bytes, int root, hero_stream_t stream) {
  HERO_COMM_READY(comm);
  HERO_REQUIRE(buf && bytes > 0, "hero_comm_broadcast: empty buffer");
  State* s = static_cast<State*>(comm);
  HERO_REQUIRE(root >= 0 && root < s->world, "hero_comm_broadcast: root %d of %d", root, s->world);
  const int rc = api().Broadcast(buf, buf, bytes, kInt8, root, s->comm, static_cast<hipStream_t>(stream));
  return rc ? fail("hero_comm_broadcast", rc) : HERO_OK;
}

// recv[r * bytes_per_rank ...] = rank r's send buffer (equal sizes: the callers pad, model/pretrain.py:383-401)
extern "C" int hero_comm_allgather(void* comm, const void* send, void* recv, size_t bytes_per_rank, hero_stream_t stream) {
  HERO_COMM_READY(comm);
  HERO_REQUIRE(send && recv && bytes_per_rank > 0, "hero_comm_allgather: empty buffer");
  State* s = static_cast<State*>(comm);
  const int rc = api().AllGather(send, recv, bytes_per_rank, kInt8, s->comm, static_cast<hipStream_t>(stream));
  return rc ? fail("hero_comm_allgather", rc) : HERO_OK;
}

// recv = rank 0's bytes, rank 1's bytes, ... back to back (bytes_per_rank[world], host array, the same on every rank): the
// variable-length gather of the cross-GPU negatives without padding to the longest rank (model/pretrain.py:383-401 pads,
// gathers and slices); one broadcast per rank inside ONE RCCL group.
extern "C" int hero_comm_allgather_var(void* comm, const void* send, void* recv, const size_t* bytes_per_rank, hero_stream_t stream) {
  HERO_COMM_READY(comm);
  HERO_REQUIRE(recv && bytes_per_rank, "hero_comm_allgather_var: null pointer");
  State* s = static_cast<State*>(comm);
  HERO_REQUIRE(bytes_per_rank[s->rank] == 0 || send, "hero_comm_allgather_var: null send buffer");
  int rc = api().GroupStart();
  if (rc) return fail("hero_comm_allgather_var(group)", rc);
  size_t off = 0;
  for (int r = 0; r < s->world && !rc; ++r) {
    const size_t n = bytes_per_rank[r];
    if (n) rc = api().Broadcast(r == s->rank ? send : static_cast<const char*>(recv) + off, static_cast<char*>(recv) + off, n, kInt8, r, s->comm,
                                static_cast<hipStream_t>(stream));
    off += n;
  }
  const int rc2 = api().GroupEnd();
  if (rc || rc2) return fail("hero_comm_allgather_var", rc ? rc : rc2);
  return HERO_OK;
}
