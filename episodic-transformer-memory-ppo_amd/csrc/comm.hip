// Gradient exchange of the data-parallel optimiser step: a thin wrapper over RCCL (one communicator per
// process, one sum all-reduce of the flat fp32 gradient bucket between backward() and clipping, upstream trainer.py:310-311 has
// no counterpart).  RCCL is resolved at the first call with dlopen -- the library has no link-time dependency on it, and when the
// host framework is already in the process its copy of librccl.so.1 is the one that is found (same soname) -- so single-GPU use
// never touches it.  The collective is enqueued on the caller's stream (capturable in a HIP graph like every other entry).
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include "etm_common.h"

namespace {
struct Rccl {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
} g_rccl;

int load_rccl() {
  if (g_rccl.handle) return ETM_OK;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return ETM_ENOCOMM;
  Rccl r;
  r.get_unique_id = (decltype(r.get_unique_id))dlsym(h, "ncclGetUniqueId");
  r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(h, "ncclCommInitRank");
  r.all_reduce = (decltype(r.all_reduce))dlsym(h, "ncclAllReduce");
  r.comm_destroy = (decltype(r.comm_destroy))dlsym(h, "ncclCommDestroy");
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) return ETM_ENOCOMM;
  r.handle = h;
  g_rccl = r;
  return ETM_OK;
}
inline int rccl_status(ncclResult_t rc) { return rc == ncclSuccess ? ETM_OK : ETM_ERCCL_BASE + (int)rc; }
}  // namespace

static_assert(sizeof(ncclUniqueId) == ETM_COMM_ID_BYTES, "etm_hip.h: ETM_COMM_ID_BYTES");

extern "C" int etm_comm_unique_id(void *id_out) {
  if (!id_out) return ETM_EINVAL;
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  rc = rccl_status(g_rccl.get_unique_id(&id));
  if (rc == ETM_OK) memcpy(id_out, &id, sizeof(id));
  return rc;
}

extern "C" int etm_comm_init(const void *id, int rank, int world, void **comm_out) {
  if (!id || !comm_out || world <= 0 || rank < 0 || rank >= world) return ETM_EINVAL;
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  rc = rccl_status(g_rccl.comm_init_rank(&comm, world, uid, rank));   // binds to the calling thread's current HIP device
  if (rc == ETM_OK) *comm_out = (void *)comm;
  return rc;
}

extern "C" int etm_allreduce_f32(void *comm, const float *send, float *recv, int64_t count, void *stream) {
  if (!comm || !send || !recv || count <= 0) return ETM_EINVAL;
  if (!g_rccl.handle) return ETM_ENOCOMM;
  return rccl_status(g_rccl.all_reduce(send, recv, (size_t)count, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
}

extern "C" int etm_comm_destroy(void *comm) {
  if (!comm) return ETM_EINVAL;
  if (!g_rccl.handle) return ETM_ENOCOMM;
  return rccl_status(g_rccl.comm_destroy((ncclComm_t)comm));
}
