// The one collective of the data-parallel step as a C-ABI entry point (`allreduce_avg`): the flat fp32
// gradient buffer averaged over the ranks by RCCL over xGMI, on the caller's stream (reference: the DDP wrapper of
// trainers/base/base_trainer.py:92-117 averages every gradient over the ranks before optimizer.step(), :350-361).
// RCCL is resolved at run time (dlopen; the copy torch already mapped when there is one), so libmvk.so has no link-time
// dependency on it and single-GPU processes never touch it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.hpp"

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_load() {
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names)  // the copy already mapped into the process (torch's) first
    if (!g_rccl.h) g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
  for (const char* n : names)
    if (!g_rccl.h) g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl.h) return;
#define MVK_SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.h, name))
  MVK_SYM(GetUniqueId, "ncclGetUniqueId");
  MVK_SYM(CommInitRank, "ncclCommInitRank");
  MVK_SYM(CommDestroy, "ncclCommDestroy");
  MVK_SYM(CommCount, "ncclCommCount");
  MVK_SYM(CommUserRank, "ncclCommUserRank");
  MVK_SYM(AllReduce, "ncclAllReduce");
  MVK_SYM(GroupStart, "ncclGroupStart");
  MVK_SYM(GroupEnd, "ncclGroupEnd");
#undef MVK_SYM
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.CommCount && g_rccl.CommUserRank &&
              g_rccl.AllReduce && g_rccl.GroupStart && g_rccl.GroupEnd;
}
bool rccl() {
  std::call_once(g_rccl_once, rccl_load);
  return g_rccl.ok;
}
}  // namespace

extern "C" {

int mvk_comm_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

/* 1 when RCCL could be resolved in this process (the library never links it), else 0: lets every rank tell the others BEFORE the
 * collective mvk_comm_init whether it can take part (a rank that raised there would leave the others waiting in the rendezvous) */
int mvk_comm_available(void) { return rccl() ? 1 : 0; }

/* rank 0: a fresh rendezvous id (mvk_comm_id_bytes() bytes) that the caller hands to every rank (any side channel) */
int mvk_comm_unique_id(void* id) {
  if (!id) return MVK_EINVAL;
  if (!rccl()) return MVK_ELAUNCH;
  return g_rccl.GetUniqueId(static_cast<ncclUniqueId*>(id)) == ncclSuccess ? MVK_OK : MVK_ELAUNCH;
}

/* collective over the `world` ranks: every rank calls it with the same id, on the device its gradient buffer lives on */
int mvk_comm_init(void** comm, int world, int rank, const void* id) {
  if (!comm || !id || world < 1 || rank < 0 || rank >= world) return MVK_EINVAL;
  if (!rccl()) return MVK_ELAUNCH;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  if (g_rccl.CommInitRank(&c, world, uid, rank) != ncclSuccess) return MVK_ELAUNCH;
  *comm = c;
  return MVK_OK;
}

int mvk_comm_destroy(void* comm) {
  if (!comm) return MVK_OK;
  if (!rccl()) return MVK_ELAUNCH;
  return g_rccl.CommDestroy(static_cast<ncclComm_t>(comm)) == ncclSuccess ? MVK_OK : MVK_ELAUNCH;
}

/* what the COMMUNICATOR says: the number of ranks that took part in mvk_comm_init and this process's rank among them (bench.py
 * reports n_gpus from here, not from the launcher's environment) */
int mvk_comm_size(void* comm, int* world, int* rank) {
  if (!comm || (!world && !rank)) return MVK_EINVAL;
  if (!rccl()) return MVK_ELAUNCH;
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  if (world && g_rccl.CommCount(c, world) != ncclSuccess) return MVK_ELAUNCH;
  if (rank && g_rccl.CommUserRank(c, rank) != ncclSuccess) return MVK_ELAUNCH;
  return MVK_OK;
}

/* Averages `nrange` disjoint ranges of one buffer (offsets / counts in floats, HOST arrays) as ONE RCCL group on `stream`: the
 * form the overlapped step uses — the ranges whose gradients are final early go first, on a communication stream that waits
 * for an event recorded inside the replayed graph, the rest behind the end of the backward pass.  A range longer than
 * `seg_floats` (> 0) is cut into segments of that size (per-link pipelining over xGMI). */
int mvk_allreduce_avg_ranges(float* buf, const int64_t* off, const int64_t* cnt, int nrange, int64_t seg_floats, void* comm,
                             void* stream) {
  if (!buf || !off || !cnt || nrange < 0 || !comm) return MVK_EINVAL;
  for (int i = 0; i < nrange; ++i)
    if (off[i] < 0 || cnt[i] < 0) return MVK_EINVAL;
  if (nrange == 0) return MVK_OK;
  if (!rccl()) return MVK_ELAUNCH;
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  hipStream_t s = mvk_stream(stream);
  const int64_t seg = seg_floats > 0 ? ((seg_floats + 63) & ~int64_t(63)) : INT64_MAX;
  bool ok = g_rccl.GroupStart() == ncclSuccess;
  for (int i = 0; i < nrange && ok; ++i)
    for (int64_t o = 0; o < cnt[i] && ok; o += seg) {
      const int64_t n = cnt[i] - o < seg ? cnt[i] - o : seg;
      float* p = buf + off[i] + o;
      ok = g_rccl.AllReduce(p, p, (size_t)n, ncclFloat32, ncclAvg, c, s) == ncclSuccess;
    }
  ok = (g_rccl.GroupEnd() == ncclSuccess) && ok;
  return ok ? MVK_OK : MVK_ELAUNCH;
}

/* buf[i] <- mean over the ranks of buf[i], in place, enqueued on `stream` (never synchronises).  nseg > 1 cuts the buffer into
 * nseg equal parts issued as one RCCL group: per-link pipelining over xGMI without the caller managing buckets. */
int mvk_allreduce_avg(float* buf, int64_t n, int nseg, void* comm, void* stream) {
  if (!buf || n < 0 || !comm || nseg < 1) return MVK_EINVAL;
  if (n == 0) return MVK_OK;
  if (!rccl()) return MVK_ELAUNCH;
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  hipStream_t s = mvk_stream(stream);
  if (nseg == 1) return g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclAvg, c, s) == ncclSuccess ? MVK_OK : MVK_ELAUNCH;
  const int64_t per = ((n + nseg - 1) / nseg + 63) & ~int64_t(63);
  bool ok = g_rccl.GroupStart() == ncclSuccess;
  for (int64_t o = 0; o < n && ok; o += per) {
    const int64_t cnt = n - o < per ? n - o : per;
    ok = g_rccl.AllReduce(buf + o, buf + o, (size_t)cnt, ncclFloat32, ncclAvg, c, s) == ncclSuccess;
  }
  ok = (g_rccl.GroupEnd() == ncclSuccess) && ok;
  return ok ? MVK_OK : MVK_ELAUNCH;
}

/* ---- events that cross the boundary of a replayed hipGraph --------------------------------------------------------------
 * The captured step (trainers/graph.py) records an EXTERNAL event (hipEventRecordWithFlags(hipEventRecordExternal): an event
 * record NODE of the graph, not a capture-internal dependency) at the point where a range of the gradient buffer is final; a
 * communication stream outside the graph waits for it and starts that range's collective while the rest of the backward pass
 * still runs.  torch's Event(external=True) does not reach hipEventRecordWithFlags on ROCm, hence these three entry points. */
int mvk_event_create(void** ev) {
  if (!ev) return MVK_EINVAL;
  hipEvent_t e = nullptr;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return MVK_ELAUNCH;
  *ev = e;
  return MVK_OK;
}
int mvk_event_destroy(void* ev) {
  if (!ev) return MVK_OK;
  return hipEventDestroy(static_cast<hipEvent_t>(ev)) == hipSuccess ? MVK_OK : MVK_ELAUNCH;
}
/* external != 0: an event-record NODE when `stream` is capturing (a plain record otherwise).  The node is added to the graph
 * being captured by hand — hipStreamGetCaptureInfo_v2 (graph + the stream's current dependency set), hipGraphAddEventRecordNode
 * behind those dependencies, hipStreamUpdateCaptureDependencies so that the stream continues behind the node — because
 * hipEventRecordWithFlags(hipEventRecordExternal) returns an error under torch's stream capture on the HIP 7.0 runtime torch
 * ships (measured, tools/extevent_order_probe.py), and a failed call inside a capture invalidates it. */
int mvk_event_record(void* ev, int external, void* stream) {
  if (!ev) return MVK_EINVAL;
  hipEvent_t e = static_cast<hipEvent_t>(ev);
  hipStream_t s = mvk_stream(stream);
  if (external) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo_v2(s, &st, &id, &graph, &deps, &ndeps) != hipSuccess) return MVK_ELAUNCH;
    if (st == hipStreamCaptureStatusActive) {
      hipGraphNode_t node = nullptr;
      hipError_t rc = hipGraphAddEventRecordNode(&node, graph, deps, ndeps, e);
      if (rc == hipSuccess) rc = hipStreamUpdateCaptureDependencies(s, &node, 1, hipStreamSetCaptureDependencies);
      if (rc != hipSuccess) {
        if (getenv("MVK_SYNC_DEBUG")) fprintf(stderr, "[mvk] external event node: %s\n", hipGetErrorString(rc));
        return MVK_ELAUNCH;
      }
      return MVK_OK;
    }
  }
  return hipEventRecord(e, s) == hipSuccess ? MVK_OK : MVK_ELAUNCH;
}
int mvk_stream_wait_event(void* stream, void* ev) {
  if (!ev) return MVK_EINVAL;
  return hipStreamWaitEvent(mvk_stream(stream), static_cast<hipEvent_t>(ev), 0) == hipSuccess ? MVK_OK : MVK_ELAUNCH;
}

}  // extern "C"
