// Data-parallel gradient exchange over RCCL, owned by the library (C ABI: vct_comm_*).
//
// replaces: torch.nn.parallel.DistributedDataParallel's bucketed NCCL all-reduce (reference train.py:217-219,
// utils.py:137-146) -- and, in its sharded form, also 7/8 of the optimizer pass at 8 GPUs.
//
// One communicator per process (one process per GPU), created from a 128-byte unique id that rank 0 generates and
// the caller distributes (the Python side uses the torch.distributed store it already has).  Every collective is
// enqueued on a stream the communicator OWNS, ordered behind the caller's compute stream by a recorded event edge, so
// the xGMI transfers of a gradient bucket run while backward keeps producing the next one; vct_comm_wait makes a
// compute stream wait for everything issued so far.  The calls are plain stream work: they can be recorded into launch
// lists (vct_cmdlist_*) like kernel launches.
//
// Collectives are in place over slices of the flat fp32 gradient / parameter buffers:
//   all-reduce(AVG)            g[a:b) <- mean over ranks                                   (replicated optimizer)
//   reduce-scatter(AVG)        rank r receives mean(g)[a + r n : a + (r+1) n),  n = (b-a)/W (sharded optimizer: Adam
//   all-gather                 p[a:b) <- every rank's updated shard                         on the owned 1/W, then gather)
// xGMI on MI355X is a full mesh of point-to-point links (7 x ~153 GB/s per GPU): RCCL picks direct reduce-scatter /
// all-gather algorithms there, whose per-link volume is (b-a)/W per phase instead of a ring's 2 (W-1)/W (b-a) over one link.
//
// RCCL is bound at run time (dlopen of the librccl the process already has -- torch's -- or the system one): the
// library itself links against the HIP runtime only, so it loads on a host without RCCL and single-GPU use never touches it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "../../include/vct_hip.h"
#include "vct_runtime.h"

namespace vct {

// the slice of rccl.h this file needs (ABI-stable NCCL 2.x definitions)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclSum = 0, ncclAvg = 4 };
enum { ncclFloat32 = 7, ncclBfloat16 = 9 };

struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
};
static Rccl g_rccl;
static std::once_flag g_rccl_once;

static void rccl_bind() {
  const char* names[] = {"librccl.so", "librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (h) break; }   // the copy already in the process
  if (!h) for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  if (!h) return;
  Rccl& r = g_rccl;
  r.h = h;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.ReduceScatter = (decltype(r.ReduceScatter))dlsym(h, "ncclReduceScatter");
  r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
  r.Broadcast = (decltype(r.Broadcast))dlsym(h, "ncclBroadcast");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.ReduceScatter && r.AllGather && r.Broadcast;
}
static bool rccl_ready() { std::call_once(g_rccl_once, rccl_bind); return g_rccl.ok; }

struct Comm {
  ncclComm_t nccl = nullptr;
  hipStream_t stream = nullptr;
  int rank = 0, world = 1;
};

static int nccl_dtype(int dtype) { return dtype == VCT_F32 ? ncclFloat32 : (dtype == VCT_BF16 ? ncclBfloat16 : -1); }
static size_t esize(int dtype) { return dtype == VCT_F32 ? 4 : 2; }
constexpr int VCT_E_RCCL = 10000;   // positive codes >= 10000: ncclResult_t + 10000

}  // namespace vct
using namespace vct;

extern "C" int vct_stream_wait(void* waiter_stream, void* signal_stream);

extern "C" int vct_comm_available(void) { return rccl_ready() ? 1 : 0; }

extern "C" int vct_comm_unique_id(uint8_t* out128) {
  if (out128 == nullptr) return VCT_E_ARG;
  if (!rccl_ready()) return VCT_E_RCCL;
  ncclUniqueId id;
  const int r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return VCT_E_RCCL + r;
  memcpy(out128, id.internal, 128);
  return VCT_OK;
}

extern "C" int vct_comm_init(const uint8_t* id128, int rank, int world, void** out_comm) {
  if (id128 == nullptr || out_comm == nullptr || world < 1 || rank < 0 || rank >= world) return VCT_E_ARG;
  if (!rccl_ready()) return VCT_E_RCCL;
  Comm* c = new (std::nothrow) Comm();
  if (c == nullptr) return (int)hipErrorOutOfMemory;
  c->rank = rank; c->world = world;
  // an ordinary (non-blocking) stream.  A high-priority stream was measured on MI355X and rejected: its mere existence
  // slowed the compute streams' kernels by 10 %, and with collectives in flight on it every kernel of the step ran 2-3.5x
  // slower (step 2.65 -> 7.7 ms at world size 1); VCT_COMM_PRIO=1 re-enables it for experiments.
  static const char* prio_env = getenv("VCT_COMM_PRIO");
  // VCT_COMM_CU_MASK=N (bench.py --comm-cu-mask N): confine the collectives' kernels to the first N CUs (hipExtStreamCreateWithCUMask;
  // only "first N" masks take effect on this runtime, tools/cu_mask_probe2.py) so that RCCL cannot spread over the compute
  // streams' CUs.  Unmeasured at N > 1 ranks (1-GPU boxes): an experiment switch for the driver's scaling run.  NOTE: the masked
  // stream is a BLOCKING stream (hipExtStreamCreateWithCUMask takes no flags): unlike the other two branches it synchronizes
  // implicitly with the NULL stream -- nothing of this library runs on the NULL stream, but the A/B changes that as well as placement.
  const char* mask_env = getenv("VCT_COMM_CU_MASK");
  const int mask_n = mask_env != nullptr ? atoi(mask_env) : 0;
  if (mask_env != nullptr && (mask_n < 1 || mask_n > 255)) { delete c; return VCT_E_ARG; }   // a mask that cannot be applied is an error, not a silent no-op
  hipError_t e;
  if (mask_n > 0) {
    uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int cu = 0; cu < mask_n; cu++) words[cu >> 5] |= 1u << (cu & 31);
    e = hipExtStreamCreateWithCUMask(&c->stream, 8, words);
  } else if (prio_env != nullptr && prio_env[0] == '1') {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi);
  } else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  }
  if (e != hipSuccess) { delete c; return (int)e; }
  ncclUniqueId id;
  memcpy(id.internal, id128, 128);
  const int r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
  if (r != ncclSuccess) { (void)hipStreamDestroy(c->stream); delete c; return VCT_E_RCCL + r; }
  *out_comm = c;
  return VCT_OK;
}

extern "C" int vct_comm_destroy(void* comm) {
  if (comm == nullptr) return VCT_E_ARG;
  Comm* c = reinterpret_cast<Comm*>(comm);
  (void)hipStreamSynchronize(c->stream);
  if (c->nccl != nullptr) (void)g_rccl.CommDestroy(c->nccl);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return VCT_OK;
}

extern "C" int vct_comm_rank(void* comm) { return comm ? reinterpret_cast<Comm*>(comm)->rank : VCT_E_ARG; }
extern "C" int vct_comm_world(void* comm) { return comm ? reinterpret_cast<Comm*>(comm)->world : VCT_E_ARG; }
extern "C" int vct_comm_stream(void* comm, void** out_stream) {
  if (comm == nullptr || out_stream == nullptr) return VCT_E_ARG;
  *out_stream = (void*)reinterpret_cast<Comm*>(comm)->stream;
  return VCT_OK;
}

// order the comm stream behind `after_stream` (NULL: no new edge -- the work follows what is already on the comm stream)
static int comm_edge(Comm* c, void* after_stream, bool have_after) {
  if (!have_after) return VCT_OK;
  return vct_stream_wait((void*)c->stream, after_stream);
}

template <typename F> static int comm_issue(Comm* c, F&& f) {
  if (g_rec != nullptr) {
    rec_push(c->stream, [f](hipStream_t s) { const int r = f(s); if (r != ncclSuccess) replay_note_error(VCT_E_RCCL + r); });
    return VCT_OK;
  }
  const int r = f(c->stream);
  return r == ncclSuccess ? VCT_OK : VCT_E_RCCL + r;
}

extern "C" int vct_comm_allreduce_avg(void* comm, void* buf, int64_t count, int dtype, void* after_stream, int order_after) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const int dt = nccl_dtype(dtype);
  if (c == nullptr || buf == nullptr || count < 0 || dt < 0) return VCT_E_ARG;
  if (count == 0) return VCT_OK;
  int rc = comm_edge(c, after_stream, order_after != 0);
  if (rc != VCT_OK) return rc;
  ncclComm_t nc = c->nccl;
  return comm_issue(c, [=](hipStream_t s) { return g_rccl.AllReduce(buf, buf, (size_t)count, dt, ncclAvg, nc, s); });
}

extern "C" int vct_comm_reduce_scatter_avg(void* comm, void* buf, int64_t count_per_rank, int dtype, void* after_stream,
                                           int order_after) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const int dt = nccl_dtype(dtype);
  if (c == nullptr || buf == nullptr || count_per_rank < 0 || dt < 0) return VCT_E_ARG;
  if (count_per_rank == 0) return VCT_OK;
  int rc = comm_edge(c, after_stream, order_after != 0);
  if (rc != VCT_OK) return rc;
  ncclComm_t nc = c->nccl;
  char* mine = reinterpret_cast<char*>(buf) + (size_t)c->rank * (size_t)count_per_rank * esize(dtype);   // in place
  return comm_issue(c, [=](hipStream_t s) { return g_rccl.ReduceScatter(buf, mine, (size_t)count_per_rank, dt, ncclAvg, nc, s); });
}

extern "C" int vct_comm_all_gather(void* comm, void* buf, int64_t count_per_rank, int dtype, void* after_stream, int order_after) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const int dt = nccl_dtype(dtype);
  if (c == nullptr || buf == nullptr || count_per_rank < 0 || dt < 0) return VCT_E_ARG;
  if (count_per_rank == 0) return VCT_OK;
  int rc = comm_edge(c, after_stream, order_after != 0);
  if (rc != VCT_OK) return rc;
  ncclComm_t nc = c->nccl;
  const char* mine = reinterpret_cast<const char*>(buf) + (size_t)c->rank * (size_t)count_per_rank * esize(dtype);
  return comm_issue(c, [=](hipStream_t s) { return g_rccl.AllGather(mine, buf, (size_t)count_per_rank, dt, nc, s); });
}

extern "C" int vct_comm_broadcast(void* comm, void* buf, int64_t count, int dtype, int root, void* after_stream, int order_after) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const int dt = nccl_dtype(dtype);
  if (c == nullptr || buf == nullptr || count < 0 || dt < 0 || root < 0 || root >= c->world) return VCT_E_ARG;
  if (count == 0) return VCT_OK;
  int rc = comm_edge(c, after_stream, order_after != 0);
  if (rc != VCT_OK) return rc;
  ncclComm_t nc = c->nccl;
  return comm_issue(c, [=](hipStream_t s) { return g_rccl.Broadcast(buf, buf, (size_t)count, dt, root, nc, s); });
}

extern "C" int vct_comm_wait(void* comm, void* stream) {
  if (comm == nullptr) return VCT_E_ARG;
  return vct_stream_wait(stream, (void*)reinterpret_cast<Comm*>(comm)->stream);
}
