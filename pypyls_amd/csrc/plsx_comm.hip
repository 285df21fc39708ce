// plsx_comm.hip -- the exported collective of the sharded resampling loops (include/plsx.h, "The collective").
//
// One process per GPU; the per-rank results of a front-end call are packed into one fp64 buffer and gathered once
// (pypyls_amd/parallel.py).  The communicator is RCCL over xGMI.  libplsx.so does not LINK a communication runtime:
// the five entry points it needs are bound with dlopen / dlsym, so that a host that already carries an RCCL (PyTorch-ROCm
// does) keeps a single copy in the process, and a host without any peer never loads one.
#include "plsx_internal.h"
#include <rccl/rccl.h>          // types and enumerators only; no symbol of librccl is referenced at link time
#include <dlfcn.h>
#include <mutex>

using namespace plsxi;

namespace {

struct RcclApi {
    void* handle = nullptr;
    std::string path;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    // one process driving several devices (plsx_comm_init_all / plsx_allgather_all); optional: a librccl without them
    // still serves the one-process-per-GPU entries above
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};

std::mutex g_api_lock;
RcclApi g_api;                  // one binding per process

bool bind_all(void* h, RcclApi& a)
{
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    return a.GetUniqueId && a.CommInitRank && a.AllGather && a.CommDestroy && a.GetErrorString;
}

// path == nullptr: the copy already in the process, then the loader path, then the ROCm tree.
int load_api(plsx_ctx* ctx, const char* path)
{
    std::lock_guard<std::mutex> guard(g_api_lock);
    if (g_api.handle) {
        if (path && g_api.path != path)
            return fail(ctx, PLSX_ERR_STATE, "plsx_comm_load: this process is already bound to " + g_api.path);
        return PLSX_OK;
    }
    static const char* const names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1",
                                        "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    std::string used;
    if (path) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        used = path;
        if (!h) return fail(ctx, PLSX_ERR_STATE, std::string("plsx_comm_load: ") + dlerror());
    } else {
        for (const char* n : names) {
            h = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (h) { used = std::string(n) + " (already loaded)"; break; }
        }
        for (size_t i = 0; !h && i < sizeof(names) / sizeof(names[0]); i++) {
            h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
            if (h) used = names[i];
        }
        if (!h) return fail(ctx, PLSX_ERR_STATE, "plsx_comm_load: no librccl.so found (pass its path)");
    }
    RcclApi a;
    if (!bind_all(h, a)) {
        dlclose(h);
        return fail(ctx, PLSX_ERR_STATE, "plsx_comm_load: " + used + " lacks an nccl* entry point");
    }
    a.handle = h;
    a.path = path ? std::string(path) : used;
    g_api = a;
    return PLSX_OK;
}

int nccl_fail(plsx_ctx* ctx, const char* what, ncclResult_t r)
{
    return fail(ctx, PLSX_ERR_HIP, std::string(what) + ": " + (g_api.GetErrorString ? g_api.GetErrorString(r) : "?"));
}

}  // namespace

extern "C" {

int plsx_comm_load(plsx_ctx* ctx, const char* librccl_path)
try {
    if (!ctx) return PLSX_ERR_ARG;
    return load_api(ctx, librccl_path);
} PLSX_CATCH(ctx)

int plsx_comm_unique_id(plsx_ctx* ctx, void* id128)
try {
    if (!ctx || !id128) return PLSX_ERR_ARG;
    int rc = load_api(ctx, nullptr);
    if (rc) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the id over as 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = g_api.GetUniqueId(&id);
    if (r != ncclSuccess) return nccl_fail(ctx, "ncclGetUniqueId", r);
    std::memcpy(id128, &id, sizeof(id));
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_comm_init(plsx_ctx* ctx, const void* id128, int rank, int world)
try {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world)
        return fail(ctx, PLSX_ERR_ARG, "plsx_comm_init: bad arguments");
    if (ctx->comm) return fail(ctx, PLSX_ERR_STATE, "plsx_comm_init: the context already has a communicator");
    int rc = load_api(ctx, nullptr);
    if (rc) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t r = g_api.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return nccl_fail(ctx, "ncclCommInitRank", r);
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return PLSX_OK;
} PLSX_CATCH(ctx)

int plsx_comm_rank(const plsx_ctx* ctx, int* rank, int* world)
{
    if (!ctx) return PLSX_ERR_ARG;
    if (rank) *rank = ctx->comm_rank;
    if (world) *world = ctx->comm_world;
    return PLSX_OK;
}

int plsx_comm_transport(const plsx_ctx* ctx)
{
    if (!ctx || ctx->comm_world <= 1) return 0;
    return ctx->comm ? PLSX_TRANSPORT_RCCL : PLSX_TRANSPORT_PEER;
}

int plsx_allgather(plsx_ctx* ctx, const void* d_send, void* d_recv, long long bytes_per_rank, void* stream)
try {
    if (!ctx || bytes_per_rank < 0 || (bytes_per_rank > 0 && (!d_send || !d_recv)))
        return fail(ctx, PLSX_ERR_ARG, "plsx_allgather: bad arguments");
    if (bytes_per_rank == 0) return PLSX_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!ctx->comm) {                                   // a world of one
        if (d_send != d_recv)
            HIPCHK(hipMemcpyAsync(d_recv, d_send, static_cast<size_t>(bytes_per_rank), hipMemcpyDeviceToDevice, st));
        return PLSX_OK;
    }
    HIPCHK(hipSetDevice(ctx->device));
    // the packed buffer is fp64 throughout; a byte count that is not a multiple of 8 travels as bytes
    const bool f64 = bytes_per_rank % 8 == 0;
    ncclResult_t r = g_api.AllGather(d_send, d_recv, static_cast<size_t>(f64 ? bytes_per_rank / 8 : bytes_per_rank),
                                     f64 ? ncclFloat64 : ncclUint8, static_cast<ncclComm_t>(ctx->comm), st);
    if (r != ncclSuccess) return nccl_fail(ctx, "ncclAllGather", r);
    return PLSX_OK;
} PLSX_CATCH(ctx)

// ---- one process, several devices (SURVEY 8(b) "plsx_allgather(ctx[], nranks, ...)", 8(e) "single process driving
// ---- 8 devices (ncclCommInitAll) is sufficient; no torch.distributed") ------------------------------------------------
int plsx_comm_init_all(plsx_ctx** ctxs, int n, int transport)
try {
    if (!ctxs || n < 1 || !ctxs[0]) return PLSX_ERR_ARG;
    plsx_ctx* c0 = ctxs[0];
    for (int r = 0; r < n; r++) {
        if (!ctxs[r]) return fail(c0, PLSX_ERR_ARG, "plsx_comm_init_all: null context");
        if (ctxs[r]->comm || ctxs[r]->comm_world > 1)
            return fail(c0, PLSX_ERR_STATE, "plsx_comm_init_all: a context already belongs to a communicator");
        for (int q = 0; q < r; q++)
            if (ctxs[q] == ctxs[r]) return fail(c0, PLSX_ERR_ARG, "plsx_comm_init_all: the same context twice");
    }
    bool distinct = true;
    for (int r = 0; r < n && distinct; r++)
        for (int q = 0; q < r; q++)
            if (ctxs[q]->device == ctxs[r]->device) { distinct = false; break; }
    if (transport == PLSX_TRANSPORT_RCCL && !distinct)
        return fail(c0, PLSX_ERR_ARG, "plsx_comm_init_all: RCCL needs one rank per device (a device is listed twice)");
    const bool rccl = n > 1 && (transport == PLSX_TRANSPORT_RCCL || (transport == PLSX_TRANSPORT_AUTO && distinct));
    if (rccl) {
        int rc = load_api(c0, nullptr);
        if (rc) return rc;
        if (!g_api.CommInitAll || !g_api.GroupStart || !g_api.GroupEnd)
            return fail(c0, PLSX_ERR_STATE, "plsx_comm_init_all: " + g_api.path + " lacks ncclCommInitAll / ncclGroupStart / ncclGroupEnd");
        std::vector<int> devs(n);
        std::vector<ncclComm_t> comms(n, nullptr);
        for (int r = 0; r < n; r++) devs[r] = ctxs[r]->device;
        ncclResult_t res = g_api.CommInitAll(comms.data(), n, devs.data());
        if (res != ncclSuccess) return nccl_fail(c0, "ncclCommInitAll", res);
        for (int r = 0; r < n; r++) ctxs[r]->comm = comms[r];
    }
    for (int r = 0; r < n; r++) {
        ctxs[r]->comm_rank = r;
        ctxs[r]->comm_world = n;
        ctxs[r]->comm_team = 1;
    }
    return PLSX_OK;
} PLSX_CATCH(ctxs && ctxs[0] ? ctxs[0] : nullptr)

int plsx_allgather_all(plsx_ctx** ctxs, int n, const void* const* d_send, void* const* d_recv, long long bytes_per_rank,
                       void* const* streams)
try {
    if (!ctxs || n < 1 || !ctxs[0]) return PLSX_ERR_ARG;
    plsx_ctx* c0 = ctxs[0];
    plsx_ctx* ctx = c0;                                 // (HIPCHK reports through rank 0's context)
    if (bytes_per_rank < 0 || (bytes_per_rank > 0 && (!d_send || !d_recv)))
        return fail(c0, PLSX_ERR_ARG, "plsx_allgather_all: bad arguments");
    for (int r = 0; r < n; r++)
        if (!ctxs[r] || ctxs[r]->comm_world != n || ctxs[r]->comm_rank != r || !ctxs[r]->comm_team)
            return fail(c0, PLSX_ERR_STATE, "plsx_allgather_all: contexts are not the ranks 0 .. n-1 of one plsx_comm_init_all");
    if (bytes_per_rank == 0) return PLSX_OK;
    const size_t nb = static_cast<size_t>(bytes_per_rank);
    auto stream_of = [&](int r) { return streams ? static_cast<hipStream_t>(streams[r]) : static_cast<hipStream_t>(nullptr); };
    if (c0->comm) {                                     // RCCL: the n calls of one group, issued by this one thread
        const bool f64 = bytes_per_rank % 8 == 0;
        ncclResult_t res = g_api.GroupStart();
        if (res != ncclSuccess) return nccl_fail(c0, "ncclGroupStart", res);
        ncclResult_t bad = ncclSuccess;
        for (int r = 0; r < n; r++) {
            HIPCHK(hipSetDevice(ctxs[r]->device));
            res = g_api.AllGather(d_send[r], d_recv[r], f64 ? nb / 8 : nb, f64 ? ncclFloat64 : ncclUint8,
                                  static_cast<ncclComm_t>(ctxs[r]->comm), stream_of(r));
            if (res != ncclSuccess && bad == ncclSuccess) bad = res;
        }
        res = g_api.GroupEnd();
        if (bad != ncclSuccess) return nccl_fail(c0, "ncclAllGather", bad);
        if (res != ncclSuccess) return nccl_fail(c0, "ncclGroupEnd", res);
        return PLSX_OK;
    }
    // peer copies: rank r's stream waits for every sender's buffer to be complete on ITS stream, then pulls the n
    // blocks into its own receive buffer (hipMemcpyPeerAsync: xGMI between devices, a device copy inside one)
    std::vector<hipEvent_t> ready(n, nullptr);
    int rc = PLSX_OK;
    for (int q = 0; q < n && rc == PLSX_OK; q++) {
        if (hipSetDevice(ctxs[q]->device) != hipSuccess ||
            hipEventCreateWithFlags(&ready[q], hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(ready[q], stream_of(q)) != hipSuccess)
            rc = fail(c0, PLSX_ERR_HIP, "plsx_allgather_all: could not record the senders' events");
    }
    for (int r = 0; r < n && rc == PLSX_OK; r++) {
        hipError_t e = hipSetDevice(ctxs[r]->device);
        for (int q = 0; q < n && e == hipSuccess; q++) {
            if (q != r) e = hipStreamWaitEvent(stream_of(r), ready[q], 0);
            char* dst = static_cast<char*>(d_recv[r]) + static_cast<size_t>(q) * nb;
            if (e == hipSuccess && dst != d_send[q])
                e = hipMemcpyPeerAsync(dst, ctxs[r]->device, d_send[q], ctxs[q]->device, nb, stream_of(r));
        }
        if (e != hipSuccess) rc = fail(c0, PLSX_ERR_HIP, std::string("plsx_allgather_all: ") + hipGetErrorString(e));
    }
    for (int q = 0; q < n; q++)
        if (ready[q]) (void)hipEventDestroy(ready[q]);  // (released when the recorded work completes)
    return rc;
} PLSX_CATCH(ctxs && ctxs[0] ? ctxs[0] : nullptr)

int plsx_comm_destroy(plsx_ctx* ctx)
try {
    if (!ctx) return PLSX_ERR_ARG;
    if (!ctx->comm) { ctx->comm_rank = 0; ctx->comm_world = 1; ctx->comm_team = 0; }
    if (ctx->comm) {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
        ncclResult_t r = g_api.CommDestroy(static_cast<ncclComm_t>(ctx->comm));
        ctx->comm = nullptr;
        ctx->comm_rank = 0;
        ctx->comm_world = 1;
        ctx->comm_team = 0;
        if (r != ncclSuccess) return nccl_fail(ctx, "ncclCommDestroy", r);
    }
    return PLSX_OK;
} PLSX_CATCH(ctx)

}  // extern "C"
