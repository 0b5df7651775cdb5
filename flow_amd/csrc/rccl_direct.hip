// K15b: the all-reduce of flow_comm issued by the library itself.
//
// flow_comm (include/flow_hip.h) asks its owner for ONE primitive, an all-reduce
// (sum) of the head of a device buffer.  flow_amd/parallel.py can bind it to
// torch.distributed.all_reduce -- a Python callback, ~20 us of host time per
// collective that an 8-GPU strong-scaling run cannot hide behind kernels of a
// few microseconds -- or to flow_rccl_allreduce below: ncclAllReduce on the
// library's own stream, on a communicator created beside torch's (same RCCL
// instance: the shared object torch loaded is dlopen'ed by path, nothing is
// linked).  The unique id travels through torch.distributed once, at set-up.
#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>

#include "common.h"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t,
                            ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int fail(const char* what, ncclResult_t r) {
  flow::set_error("%s failed: %s", what,
                  g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return FLOW_HIP_ERROR;
}

}  // namespace

static_assert(NCCL_UNIQUE_ID_BYTES == FLOW_RCCL_ID_BYTES, "unique id size");

extern "C" int flow_rccl_load(const char* path) {
  if (g_rccl.handle) return FLOW_OK;
  FLOW_REQUIRE(path != nullptr, "librccl path");
  void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    flow::set_error("dlopen(%s) failed: %s", path, dlerror());
    return FLOW_HIP_ERROR;
  }
  RcclApi api;
  api.handle = h;
  api.GetUniqueId =
      reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  api.CommInitRank =
      reinterpret_cast<decltype(api.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  api.CommDestroy =
      reinterpret_cast<decltype(api.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  api.AllReduce =
      reinterpret_cast<decltype(api.AllReduce)>(dlsym(h, "ncclAllReduce"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(
      dlsym(h, "ncclGetErrorString"));
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy ||
      !api.AllReduce) {
    flow::set_error("%s does not export the RCCL entry points", path);
    return FLOW_HIP_ERROR;
  }
  g_rccl = api;
  return FLOW_OK;
}

extern "C" int flow_rccl_unique_id(char* id_host) {
  FLOW_REQUIRE(g_rccl.handle && id_host, "RCCL not loaded");
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
  memcpy(id_host, id.internal, NCCL_UNIQUE_ID_BYTES);
  return FLOW_OK;
}

extern "C" int flow_rccl_comm_create(const char* id_host, int rank, int world,
                                     void** comm_out) {
  FLOW_REQUIRE(g_rccl.handle && id_host && comm_out, "RCCL not loaded");
  FLOW_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank / world");
  ncclUniqueId id;
  memcpy(id.internal, id_host, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  const ncclResult_t r = g_rccl.CommInitRank(&comm, world, id, rank);
  if (r != ncclSuccess) return fail("ncclCommInitRank", r);
  *comm_out = comm;
  return FLOW_OK;
}

extern "C" int flow_rccl_comm_destroy(void* comm) {
  if (!comm || !g_rccl.handle) return FLOW_OK;
  const ncclResult_t r = g_rccl.CommDestroy(static_cast<ncclComm_t>(comm));
  if (r != ncclSuccess) return fail("ncclCommDestroy", r);
  return FLOW_OK;
}

// a flow_allreduce_fn: user = const flow_rccl_binding*
extern "C" int flow_rccl_allreduce(void* user, int count) {
  const flow_rccl_binding* b = static_cast<const flow_rccl_binding*>(user);
  if (!b || !b->comm || !b->buf || count <= 0 || !g_rccl.AllReduce) return 1;
  const ncclResult_t r = g_rccl.AllReduce(
      b->buf, b->buf, static_cast<size_t>(count), ncclDouble, ncclSum,
      static_cast<ncclComm_t>(b->comm), static_cast<hipStream_t>(b->stream));
  return r == ncclSuccess ? 0 : 2;
}
