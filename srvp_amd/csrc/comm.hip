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
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
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
    g_api.CommCount = (decltype(g_api.CommCount))dlsym(h, "ncclCommCount");            // (the three below are diagnostics: optional)
    g_api.CommUserRank = (decltype(g_api.CommUserRank))dlsym(h, "ncclCommUserRank");
    g_api.GetVersion = (decltype(g_api.GetVersion))dlsym(h, "ncclGetVersion");
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

// What RCCL itself reports about a communicator: info[0] = ncclCommCount (ranks), info[1] = ncclCommUserRank, info[2] = ncclGetVersion
// (-1 where the loaded librccl lacks the call).  bench.py --gpus N prints it, so that a scaling curve explains itself.
extern "C" int srvp_comm_info(void* comm, int* info3) {
    SRVP_REQUIRE(comm && info3 && g_api.lib, "srvp_comm_info: bad args / communicator not initialised");
    info3[0] = info3[1] = info3[2] = -1;
    if (g_api.CommCount) RCCL_CHECK(g_api.CommCount((ncclComm_t)comm, &info3[0]), "ncclCommCount");
    if (g_api.CommUserRank) RCCL_CHECK(g_api.CommUserRank((ncclComm_t)comm, &info3[1]), "ncclCommUserRank");
    if (g_api.GetVersion) RCCL_CHECK(g_api.GetVersion(&info3[2]), "ncclGetVersion");
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

// Generic form (round 6): dtype 0 = fp32, 1 = fp64, 2 = bf16 (the opt-in compressed gradient payload); op 0 = sum, 1 = average over ranks
// (ncclAvg: RCCL pre-scales by 1 / ranks inside the collective -- the 1/world pass of DistributedDataParallel's averaging costs no
// extra sweep over the 95 MB gradient buffer).
extern "C" int srvp_allreduce(void* comm, void* buf, int64_t n, int dtype, int op, void* stream) {
    SRVP_REQUIRE(comm && buf && n > 0 && g_api.lib && dtype >= 0 && dtype <= 2 && (op == 0 || op == 1),
                 "srvp_allreduce: bad args / communicator not initialised");
    const ncclDataType_t dt = dtype == 0 ? ncclFloat32 : (dtype == 1 ? ncclFloat64 : ncclBfloat16);
    RCCL_CHECK(g_api.AllReduce(buf, buf, (size_t)n, dt, op == 1 ? ncclAvg : ncclSum, (ncclComm_t)comm, (hipStream_t)stream), "ncclAllReduce");
    return SRVP_OK;
}

extern "C" int srvp_bcast_bytes(void* comm, void* buf, int64_t nbytes, int root, void* stream) {
    SRVP_REQUIRE(comm && buf && nbytes > 0 && g_api.lib, "srvp_bcast_bytes: bad args / communicator not initialised");
    RCCL_CHECK(g_api.Broadcast(buf, buf, (size_t)nbytes, ncclUint8, root, (ncclComm_t)comm, (hipStream_t)stream), "ncclBroadcast");
    return SRVP_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// PROTOTYPE (SRVP_COMM=peer): the SyncBatchNorm statistics exchange (reference train.py:278-283) as a one-sided peer read instead of
// an all-reduce.  42 of these per VGG step (21 BatchNorm layers x 2 directions) feed the very next kernel, each a few KB: through RCCL every one is a collective kernel
// with its own rendezvous (~10-20 us on xGMI), which at 24 sequences per GPU is 1-1.7 ms of a ~9 ms step.  Here every rank owns a
// slab (device memory shared with the other ranks of the node through hipIpc); a collective is ONE single-workgroup launch per rank:
// publish the local sums into the own slab (system-scope stores), raise the own flag to the sequence number, wait for the peers'
// flags (bounded spin), read their sums over xGMI and add them in rank order -- the same order on every rank, so all ranks hold
// bit-identical totals.  Two parities per slot: a rank can only reach use n + 2 of a slot after every peer has finished use n.
// Validated on ranks that share one device (tests/test_two_rank_equality.py); the cross-device memory model (uncached slab,
// system-scope accesses) has not been exercised on hardware yet.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct PeerK { double* slab[8]; int rank, world; long long data_off, flag_off; unsigned long long seq; int* err; };

__global__ __launch_bounds__(256) void peer_allreduce_kernel(double* buf, int n, const PeerK k) {
    double* own = (double*)((char*)k.slab[k.rank] + k.data_off);
    for (int i = threadIdx.x; i < n; i += blockDim.x) __hip_atomic_store(own + i, buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store((unsigned long long*)((char*)k.slab[k.rank] + k.flag_off), k.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    if ((int)threadIdx.x < k.world && (int)threadIdx.x != k.rank) {
        const unsigned long long* f = (const unsigned long long*)((char*)k.slab[threadIdx.x] + k.flag_off);
        const long long t0 = wall_clock64();                   // 100 MHz
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < k.seq) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 500000000ll) { bad = 1; break; }       // 5 s: a peer never arrived -- give up, flag the error
        }
    }
    __syncthreads();
    if (bad) {
        // local-only sums must never be consumed as if they were global ones: poison the result (the loss turns NaN at once) and
        // raise the error word the host polls once per step
        for (int i = threadIdx.x; i < n; i += blockDim.x) buf[i] = __longlong_as_double(0x7ff8000000000000ll);
        if (threadIdx.x == 0 && k.err) *k.err = 1;
        return;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double tot = 0.;
        for (int p = 0; p < k.world; ++p)
            tot += p == k.rank ? buf[i] : __hip_atomic_load((const double*)((char*)k.slab[p] + k.data_off) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[i] = tot;
    }
}
}  // namespace

extern "C" int srvp_peer_slab_create(int64_t bytes, void** dev_ptr, void* ipc_handle64) {
    SRVP_REQUIRE(bytes > 0 && dev_ptr && ipc_handle64, "srvp_peer_slab_create: bad args");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&p, (size_t)bytes); }
    SRVP_REQUIRE(e == hipSuccess, "srvp_peer_slab_create: allocation of %lld bytes failed: %s", (long long)bytes, hipGetErrorString(e));
    e = hipMemset(p, 0, (size_t)bytes);
    SRVP_REQUIRE(e == hipSuccess, "srvp_peer_slab_create: memset failed: %s", hipGetErrorString(e));
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, p);
    if (e != hipSuccess) { (void)hipFree(p); }
    SRVP_REQUIRE(e == hipSuccess, "srvp_peer_slab_create: hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)", hipGetErrorString(e));
    memcpy(ipc_handle64, &h, sizeof(h));
    *dev_ptr = p;
    return SRVP_OK;
}
extern "C" int srvp_peer_slab_open(const void* ipc_handle64, void** dev_ptr) {
    SRVP_REQUIRE(ipc_handle64 && dev_ptr, "srvp_peer_slab_open: bad args");
    hipIpcMemHandle_t h;
    memcpy(&h, ipc_handle64, sizeof(h));
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    SRVP_REQUIRE(e == hipSuccess, "srvp_peer_slab_open: hipIpcOpenMemHandle failed: %s", hipGetErrorString(e));
    *dev_ptr = p;
    return SRVP_OK;
}
extern "C" int srvp_peer_slab_close(void* dev_ptr, int owned) {
    if (!dev_ptr) return SRVP_OK;
    hipError_t e = owned ? hipFree(dev_ptr) : hipIpcCloseMemHandle(dev_ptr);
    SRVP_REQUIRE(e == hipSuccess, "srvp_peer_slab_close: %s", hipGetErrorString(e));
    return SRVP_OK;
}
extern "C" int srvp_peer_allreduce_f64(double* buf, int n, int rank, int world, void* const* slabs, int64_t data_off, int64_t flag_off,
                                       uint64_t seq, int* err_flag, void* stream) {
    SRVP_REQUIRE(buf && slabs && n > 0 && world >= 1 && world <= 8 && rank >= 0 && rank < world && data_off % 8 == 0 && flag_off % 8 == 0,
                 "srvp_peer_allreduce_f64: bad args (world <= 8)");
    PeerK k{};
    for (int p = 0; p < world; ++p) { SRVP_REQUIRE(slabs[p], "srvp_peer_allreduce_f64: slab %d not mapped", p); k.slab[p] = (double*)slabs[p]; }
    k.rank = rank; k.world = world; k.data_off = data_off; k.flag_off = flag_off; k.seq = seq; k.err = err_flag;
    hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, buf, n, k);
    SRVP_CHECK_LAUNCH("srvp_peer_allreduce_f64");
    return SRVP_OK;
}
