// Data-parallel collectives on RCCL over xGMI, issued on the CALLER'S stream from the C ABI: the gradient all-reduce of
// DistributedDataParallel and the SyncBatchNorm statistics exchange (reference train.py:278-283, 309-314).
//
// Why not torch.distributed: a training step of the VGG model issues 42 dependent statistics all-reduces of a few KB
// (one per BatchNorm layer and direction -- each is needed by the very next kernel, so they cannot be coalesced across
// layers); through ProcessGroupNCCL each costs a hop to the communicator's side stream and back plus Python dispatch.
// ncclAllReduce enqueued directly on the compute stream is one kernel in stream order: no event, no host work beyond the call.
//
// RCCL is bound at run time (dlopen): libsrvp_hip.so carries no link-time dependency on it, single-GPU users never load it.
#include "common.h"
#include "../../include/srvp_hip.h"
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct Api {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_api;

int load_api() {
    if (g_api.lib) return SRVP_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) { h = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (h) break; }
    SRVP_REQUIRE(h, "srvp_comm: cannot load librccl.so: %s", dlerror());
    g_api.GetUniqueId = (decltype(g_api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    g_api.CommInitRank = (decltype(g_api.CommInitRank))dlsym(h, "ncclCommInitRank");
    g_api.CommDestroy = (decltype(g_api.CommDestroy))dlsym(h, "ncclCommDestroy");
    g_api.AllReduce = (decltype(g_api.AllReduce))dlsym(h, "ncclAllReduce");
    g_api.Broadcast = (decltype(g_api.Broadcast))dlsym(h, "ncclBroadcast");
    g_api.GetErrorString = (decltype(g_api.GetErrorString))dlsym(h, "ncclGetErrorString");
    SRVP_REQUIRE(g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllReduce && g_api.Broadcast && g_api.GetErrorString,
                 "srvp_comm: librccl.so lacks an expected symbol");
    g_api.lib = h;
    return SRVP_OK;
}
#define RCCL_CHECK(call, what)                                                                   \
    do {                                                                                         \
        ncclResult_t r__ = (call);                                                               \
        if (r__ != ncclSuccess) { srvp_set_error("%s: %s", what, g_api.GetErrorString(r__)); return SRVP_ERR_LAUNCH; } \
    } while (0)
}  // namespace

extern "C" int srvp_comm_unique_id(void* id128) {
    SRVP_REQUIRE(id128, "srvp_comm_unique_id: null pointer");
    if (int rc = load_api()) return rc;
    ncclUniqueId id;
    RCCL_CHECK(g_api.GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id128, &id, sizeof(id));
    return SRVP_OK;
}

extern "C" int srvp_comm_init(const void* id128, int rank, int world, void** comm_out) {
    SRVP_REQUIRE(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, "srvp_comm_init: bad args");
    if (int rc = load_api()) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    RCCL_CHECK(g_api.CommInitRank(&c, world, id, rank), "ncclCommInitRank");
    *comm_out = (void*)c;
    return SRVP_OK;
}

extern "C" int srvp_comm_destroy(void* comm) {
    if (!comm || !g_api.lib) return SRVP_OK;
    RCCL_CHECK(g_api.CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
    return SRVP_OK;
}

extern "C" int srvp_allreduce_f64(void* comm, double* buf, int64_t n, void* stream) {
    SRVP_REQUIRE(comm && buf && n > 0 && g_api.lib, "srvp_allreduce_f64: bad args / communicator not initialised");
    RCCL_CHECK(g_api.AllReduce(buf, buf, (size_t)n, ncclFloat64, ncclSum, (ncclComm_t)comm, (hipStream_t)stream), "ncclAllReduce(f64)");
    return SRVP_OK;
}

extern "C" int srvp_allreduce_f32(void* comm, float* buf, int64_t n, void* stream) {
    SRVP_REQUIRE(comm && buf && n > 0 && g_api.lib, "srvp_allreduce_f32: bad args / communicator not initialised");
    RCCL_CHECK(g_api.AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream), "ncclAllReduce(f32)");
    return SRVP_OK;
}

extern "C" int srvp_bcast_bytes(void* comm, void* buf, int64_t nbytes, int root, void* stream) {
    SRVP_REQUIRE(comm && buf && nbytes > 0 && g_api.lib, "srvp_bcast_bytes: bad args / communicator not initialised");
    RCCL_CHECK(g_api.Broadcast(buf, buf, (size_t)nbytes, ncclUint8, root, (ncclComm_t)comm, (hipStream_t)stream), "ncclBroadcast");
    return SRVP_OK;
}
