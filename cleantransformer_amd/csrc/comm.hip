// ctmi_ddp_*: the data-parallel collectives of the SFT step issued straight on RCCL (include/ctmi355.h; reference call sites
// examples/ft_bloom_DDP.py:99 `DDP(model, device_ids=[local_rank])` and :183 `init_process_group("nccl")`).
//
// Why a path beside torch.distributed: the collectives share the GPU with the backward's GEMMs, and how many CUs they may take has to
// be decided TOGETHER with the GEMM launch policy (ctmi_set_launch_policy: persistent GEMMs on 256 - R CUs).  An RCCL communicator
// created here carries its own channel cap (ncclConfig_t.maxCTAs = R: one channel = one workgroup = one CU) instead of a process-wide
// NCCL_MAX_NCHANNELS that has to be exported before anything initialises RCCL.  Each communicator owns ONE stream; a collective is
// ordered behind everything the caller's compute stream holds at the call (event), and ctmi_ddp_wait() makes a compute stream wait for
// everything issued so far — the same fencing torch's ProcessGroupNCCL does, without its per-call Python / c10d layers.
//
// librccl is opened with dlopen at first use: a single-GPU process never loads it, and libctmi355.so has no link-time dependency on it.
#include "common.h"
#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <string>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define CTMI_HAVE_RCCL_HEADER 1
#else
#define CTMI_HAVE_RCCL_HEADER 0      // a build host without the RCCL development headers: the ctmi_ddp_* entry points report CTMI_ERR_UNSUPPORTED
#endif

#if CTMI_HAVE_RCCL_HEADER
#define CTMI_HIP_OK(call, what) do { hipError_t e__ = (call); if (e__ != hipSuccess) { \
    ctmi_set_error("%s: %s", what, hipGetErrorString(e__)); return CTMI_ERR_LAUNCH; } } while (0)
#define RC(call) do { int rc__ = (call); if (rc__ != CTMI_OK) return rc__; } while (0)

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRankConfig)(ncclComm_t*, int, ncclUniqueId, int, ncclConfig_t*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_why;                                 // why librccl could not be used (dlerror() text, captured ONCE: a second call returns NULL)

template <typename F> bool sym(void* h, const char* name, F& out) {
    out = reinterpret_cast<F>(dlsym(h, name));
    return out != nullptr;
}
Rccl* rccl() {
    std::call_once(g_rccl_once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
            const char* e = dlerror();
            if (e != nullptr) g_rccl_why = e;
        }
        if (!g_rccl.handle) { if (g_rccl_why.empty()) g_rccl_why = "dlopen failed"; return; }
        Rccl& r = g_rccl;
        r.ok = sym(r.handle, "ncclGetUniqueId", r.GetUniqueId) && sym(r.handle, "ncclCommInitRankConfig", r.CommInitRankConfig) &&
               sym(r.handle, "ncclCommDestroy", r.CommDestroy) && sym(r.handle, "ncclAllReduce", r.AllReduce) &&
               sym(r.handle, "ncclAllGather", r.AllGather) && sym(r.handle, "ncclBroadcast", r.Broadcast) &&
               sym(r.handle, "ncclGetErrorString", r.GetErrorString);
        if (!r.ok) g_rccl_why = "librccl.so lacks one of the nccl* entry points this library binds";
    });
    return g_rccl.ok ? &g_rccl : nullptr;
}

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t st = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    std::mutex mu;                                  // one collective is enqueued at a time (events are reused)
};

#define RCCL_OK(call, what) do { const ncclResult_t r__ = (call); if (r__ != ncclSuccess) { \
    ctmi_set_error("%s: RCCL: %s", what, R->GetErrorString ? R->GetErrorString(r__) : "error"); return CTMI_ERR_LAUNCH; } } while (0)

// order the communicator's stream behind the caller's compute stream as it stands now
int fence_in(Comm* c, hipStream_t compute, const char* what) {
    CTMI_HIP_OK(hipEventRecord(c->ev_in, compute), what);
    CTMI_HIP_OK(hipStreamWaitEvent(c->st, c->ev_in, 0), what);
    return CTMI_OK;
}
}  // namespace

extern "C" int ctmi_ddp_unique_id(void* id128) {
    CTMI_REQUIRE(id128 != nullptr, "ddp_unique_id: null output");
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the id over as 128 bytes");
    Rccl* R = rccl();
    CTMI_REQUIRE(R != nullptr, "ddp_unique_id: librccl.so could not be loaded (%s)", g_rccl_why.c_str());
    ncclUniqueId id;
    RCCL_OK(R->GetUniqueId(&id), "ddp_unique_id");
    std::memcpy(id128, &id, sizeof(id));
    return CTMI_OK;
}

extern "C" int ctmi_ddp_create(const void* id128, int rank, int world, int max_channels, void** comm_out) {
    CTMI_REQUIRE(id128 != nullptr && comm_out != nullptr, "ddp_create: null argument");
    CTMI_REQUIRE(world >= 1 && rank >= 0 && rank < world, "ddp_create: rank %d of %d", rank, world);
    Rccl* R = rccl();
    CTMI_REQUIRE(R != nullptr, "ddp_create: librccl.so could not be loaded (%s)", g_rccl_why.c_str());
    Comm* c = new Comm();
    c->rank = rank; c->world = world;
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
    if (max_channels > 0) cfg.maxCTAs = max_channels;                       // one channel = one workgroup: the collectives' CU budget
    const ncclResult_t r = R->CommInitRankConfig(&c->comm, world, id, rank, &cfg);
    if (r != ncclSuccess) {
        ctmi_set_error("ddp_create: ncclCommInitRankConfig: %s", R->GetErrorString(r));
        delete c;
        return CTMI_ERR_LAUNCH;
    }
    if (hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming) != hipSuccess) {
        ctmi_set_error("ddp_create: stream / event creation failed");
        R->CommDestroy(c->comm);
        delete c;
        return CTMI_ERR_LAUNCH;
    }
    *comm_out = c;
    return CTMI_OK;
}

extern "C" int ctmi_ddp_destroy(void* comm) {
    if (comm == nullptr) return CTMI_OK;
    Comm* c = reinterpret_cast<Comm*>(comm);
    Rccl* R = rccl();
    (void)hipStreamSynchronize(c->st);
    if (R != nullptr && c->comm != nullptr) (void)R->CommDestroy(c->comm);
    (void)hipEventDestroy(c->ev_in);
    (void)hipEventDestroy(c->ev_out);
    (void)hipStreamDestroy(c->st);
    delete c;
    return CTMI_OK;
}

extern "C" int ctmi_ddp_all_reduce(void* comm, void* buf, int64_t count, int dtype, void* compute_stream) {
    CTMI_REQUIRE(comm != nullptr && buf != nullptr && count >= 0, "ddp_all_reduce: bad argument");
    CTMI_REQUIRE(dtype == CTMI_F32 || dtype == CTMI_BF16, "ddp_all_reduce: dtype %d", dtype);
    Comm* c = reinterpret_cast<Comm*>(comm);
    Rccl* R = rccl();
    CTMI_REQUIRE(R != nullptr && c->comm != nullptr, "ddp_all_reduce: no RCCL communicator behind this handle");
    std::lock_guard<std::mutex> lk(c->mu);
    RC(fence_in(c, as_stream(compute_stream), "ddp_all_reduce: fence"));
    RCCL_OK(R->AllReduce(buf, buf, (size_t)count, dtype == CTMI_F32 ? ncclFloat : ncclBfloat16, ncclSum, c->comm, c->st), "ddp_all_reduce");
    return CTMI_OK;
}

extern "C" int ctmi_ddp_all_gather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* compute_stream) {
    CTMI_REQUIRE(comm != nullptr && send != nullptr && recv != nullptr && bytes_per_rank >= 0, "ddp_all_gather: bad argument");
    Comm* c = reinterpret_cast<Comm*>(comm);
    Rccl* R = rccl();
    CTMI_REQUIRE(R != nullptr && c->comm != nullptr, "ddp_all_gather: no RCCL communicator behind this handle");
    std::lock_guard<std::mutex> lk(c->mu);
    RC(fence_in(c, as_stream(compute_stream), "ddp_all_gather: fence"));
    RCCL_OK(R->AllGather(send, recv, (size_t)bytes_per_rank, ncclInt8, c->comm, c->st), "ddp_all_gather");
    return CTMI_OK;
}

extern "C" int ctmi_ddp_broadcast(void* comm, void* buf, int64_t bytes, int root, void* compute_stream) {
    CTMI_REQUIRE(comm != nullptr && buf != nullptr && bytes >= 0, "ddp_broadcast: bad argument");
    Comm* c = reinterpret_cast<Comm*>(comm);
    CTMI_REQUIRE(root >= 0 && root < c->world, "ddp_broadcast: root %d of %d", root, c->world);
    Rccl* R = rccl();
    CTMI_REQUIRE(R != nullptr && c->comm != nullptr, "ddp_broadcast: no RCCL communicator behind this handle");
    std::lock_guard<std::mutex> lk(c->mu);
    RC(fence_in(c, as_stream(compute_stream), "ddp_broadcast: fence"));
    RCCL_OK(R->Broadcast(buf, buf, (size_t)bytes, ncclInt8, root, c->comm, c->st), "ddp_broadcast");
    return CTMI_OK;
}

extern "C" int ctmi_ddp_wait(void* comm, void* compute_stream) {
    CTMI_REQUIRE(comm != nullptr, "ddp_wait: null communicator");
    Comm* c = reinterpret_cast<Comm*>(comm);
    std::lock_guard<std::mutex> lk(c->mu);
    CTMI_HIP_OK(hipEventRecord(c->ev_out, c->st), "ddp_wait: record");
    CTMI_HIP_OK(hipStreamWaitEvent(as_stream(compute_stream), c->ev_out, 0), "ddp_wait: wait");
    return CTMI_OK;
}

#else   // !CTMI_HAVE_RCCL_HEADER
#define CTMI_NO_RCCL(name) do { ctmi_set_error(name ": this libctmi355.so was built without <rccl/rccl.h>; rebuild it on a ROCm image with RCCL"); return CTMI_ERR_UNSUPPORTED; } while (0)
extern "C" int ctmi_ddp_unique_id(void*) { CTMI_NO_RCCL("ddp_unique_id"); }
extern "C" int ctmi_ddp_create(const void*, int, int, int, void**) { CTMI_NO_RCCL("ddp_create"); }
extern "C" int ctmi_ddp_destroy(void*) { return CTMI_OK; }
extern "C" int ctmi_ddp_all_reduce(void*, void*, int64_t, int, void*) { CTMI_NO_RCCL("ddp_all_reduce"); }
extern "C" int ctmi_ddp_all_gather(void*, const void*, void*, int64_t, void*) { CTMI_NO_RCCL("ddp_all_gather"); }
extern "C" int ctmi_ddp_broadcast(void*, void*, int64_t, int, void*) { CTMI_NO_RCCL("ddp_broadcast"); }
extern "C" int ctmi_ddp_wait(void*, void*) { CTMI_NO_RCCL("ddp_wait"); }
#endif
