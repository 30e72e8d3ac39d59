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
  MVK_SYM(AllReduce, "ncclAllReduce");
  MVK_SYM(GroupStart, "ncclGroupStart");
  MVK_SYM(GroupEnd, "ncclGroupEnd");
#undef MVK_SYM
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.GroupStart && g_rccl.GroupEnd;
}
bool rccl() {
  std::call_once(g_rccl_once, rccl_load);
  return g_rccl.ok;
}
}  // namespace

extern "C" {

int mvk_comm_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

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

}  // extern "C"
