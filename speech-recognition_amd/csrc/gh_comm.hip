// C ABI: the ONE exchange step of the training path -- the all-reduce of the EM sufficient statistics -- through RCCL,
// enqueued on the context's stream (gh_comm_create, gh_stats_allreduce).  No reference counterpart (the reference is a
// single process); what is reduced are the sums of hmm_state.py:134-148.
//
// librccl is 570 MB: it is NOT a link-time dependency of libgmmhmm.so but opened on the first gh_comm_* call, and from
// the directory of the HIP runtime this library itself runs on (dladdr of hipGetDeviceCount): a process that also
// imported PyTorch holds a second HIP runtime and a second RCCL bound to it, and a collective enqueued on OUR stream
// through THEIR runtime would be a collective on a stream that runtime has never seen.
#include "gh_internal.h"
#include "gh_host.h"
#include <dlfcn.h>
#include <chrono>
#include <mutex>
#include <unistd.h>
#include <rccl/rccl.h>

struct gh_comm {
    gh_ctx* ctx;
    ncclComm_t comm;
    int rank, world;
    double* d_one;      // one double on the device: the barrier's payload
    bool aborted = false;   // ncclCommAbort has run (failure path): every later call on the handle returns GH_ERR_COMM
};

namespace {

struct rccl_api {
    void* handle = nullptr;
    std::string path, error;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

rccl_api g_rccl;
std::once_flag g_rccl_once;

void open_rccl() {
    rccl_api& r = g_rccl;
    std::vector<std::string> tries;
    if (const char* e = getenv("GMMHMM_RCCL_LIB")) tries.push_back(e);
    Dl_info info;
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {   // next to the HIP runtime we are bound to
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) {
            dir.resize(slash + 1);
            tries.push_back(dir + "librccl.so.1");
            tries.push_back(dir + "librccl.so");
        }
    }
    tries.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string& p : tries) {
        r.handle = dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (r.handle) { r.path = p; break; }
        const char* why = dlerror();
        r.error += p + ": " + (why ? why : "?") + "; ";
    }
    if (!r.handle) return;
#define GH_SYM(field, name)                                              \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name)); \
    if (!r.field) { r.error += std::string("missing symbol ") + name + "; "; }
    GH_SYM(GetUniqueId, "ncclGetUniqueId")
    GH_SYM(CommInitRank, "ncclCommInitRank")
    GH_SYM(CommDestroy, "ncclCommDestroy")
    GH_SYM(CommAbort, "ncclCommAbort")
    GH_SYM(CommCount, "ncclCommCount")
    GH_SYM(CommGetAsyncError, "ncclCommGetAsyncError")
    GH_SYM(AllReduce, "ncclAllReduce")
    GH_SYM(GetVersion, "ncclGetVersion")
    GH_SYM(GetErrorString, "ncclGetErrorString")
#undef GH_SYM
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommCount || !r.AllReduce || !r.GetErrorString) {
        dlclose(r.handle);
        r.handle = nullptr;
    }
}

int need_rccl(const char* who) {
    std::call_once(g_rccl_once, open_rccl);
    if (!g_rccl.handle) {
        gh_set_error("%s: librccl not available (%s)", who, g_rccl.error.c_str());
        return GH_ERR_UNSUPPORTED;
    }
    return GH_OK;
}

#define GH_RCCL(call, who)                                                                   \
    do {                                                                                     \
        ncclResult_t r_ = (call);                                                            \
        if (r_ != ncclSuccess) {                                                             \
            gh_set_error("%s: %s -> %s", who, #call, g_rccl.GetErrorString(r_));             \
            return GH_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

}  // namespace

// ---- failure path ("Failure detection": the C ABI returns error codes).  A rank that dies before
// or inside a collective leaves its peers inside ncclAllReduce: a bare hipStreamSynchronize behind it never returns.
// Every wait behind a collective therefore polls the stream, asks RCCL for asynchronous errors (a closed peer connection
// shows up there) and gives up at a deadline -- GMMHMM_COMM_TIMEOUT seconds, default 300 -- by aborting the communicator
// (ncclCommAbort: the collective's kernels leave, the stream drains) and returning GH_ERR_COMM.
static double comm_timeout_s() {
    if (const char* e = getenv("GMMHMM_COMM_TIMEOUT")) {
        const double v = atof(e);
        if (v > 0) return v;
    }
    return 300.0;
}

static void comm_abort_now(gh_comm* c) {
    if (c->aborted) return;
    c->aborted = true;
    if (c->comm) {
        if (g_rccl.CommAbort) g_rccl.CommAbort(c->comm);
        else if (g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
        c->comm = nullptr;
    }
    // the aborted collective's kernels leave on their own; give the stream a bounded time to drain
    const auto t0 = std::chrono::steady_clock::now();
    while (hipStreamQuery(c->ctx->stream) == hipErrorNotReady &&
           std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 10.0)
        usleep(1000);
}

extern "C" int gh_comm_abort(gh_comm* c) {
    GH_REQUIRE(c, "gh_comm_abort: NULL argument");
    hipSetDevice(c->ctx->device);
    comm_abort_now(c);
    return GH_OK;
}

// wait for the context's stream; with a communicator: deadline + RCCL's asynchronous errors (see above)
int gh_stream_wait(gh_ctx* ctx, gh_comm* c, const char* who) {
    if (!c) {
        GH_HIP(hipStreamSynchronize(ctx->stream));
        return GH_OK;
    }
    if (c->aborted) {
        gh_set_error("%s: the communicator has been aborted", who);
        return GH_ERR_COMM;
    }
    const double limit = comm_timeout_s();
    const auto t0 = std::chrono::steady_clock::now();
    for (int polls = 0;; ++polls) {
        const hipError_t q = hipStreamQuery(ctx->stream);
        if (q == hipSuccess) return GH_OK;
        if (q != hipErrorNotReady) {
            gh_set_error("%s: hipStreamQuery -> %s", who, hipGetErrorString(q));
            return GH_ERR_HIP;
        }
        if ((polls & 63) == 63 || polls > 4096) {
            ncclResult_t ar = ncclSuccess;
            if (g_rccl.CommGetAsyncError && g_rccl.CommGetAsyncError(c->comm, &ar) == ncclSuccess && ar != ncclSuccess &&
                ar != ncclInProgress) {
                gh_set_error("%s: RCCL reports an asynchronous error (%s) -- a peer rank is gone; communicator aborted", who,
                             g_rccl.GetErrorString(ar));
                comm_abort_now(c);
                return GH_ERR_COMM;
            }
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el > limit) {
                gh_set_error("%s: rank %d of %d waited %.0f s behind a collective (GMMHMM_COMM_TIMEOUT) -- a peer rank never "
                             "arrived; communicator aborted", who, c->rank, c->world, el);
                comm_abort_now(c);
                return GH_ERR_COMM;
            }
        }
        if (polls > 4096) usleep(50);      // (the first polls spin: a collective of this path takes tens of microseconds)
    }
}

extern "C" int gh_comm_unique_id(char* out_id /*[128]*/) {
    GH_REQUIRE(out_id, "gh_comm_unique_id: NULL argument");
    static_assert(sizeof(ncclUniqueId) == GH_COMM_ID_BYTES, "ncclUniqueId size");
    int rc = need_rccl("gh_comm_unique_id");
    if (rc) return rc;
    ncclUniqueId id;
    GH_RCCL(g_rccl.GetUniqueId(&id), "gh_comm_unique_id");
    memcpy(out_id, id.internal, GH_COMM_ID_BYTES);
    return GH_OK;
}

extern "C" int gh_comm_create(gh_ctx* ctx, int rank, int world, const char* unique_id, gh_comm** out) {
    GH_REQUIRE(ctx && unique_id && out, "gh_comm_create: NULL argument");
    GH_REQUIRE(world >= 1 && rank >= 0 && rank < world, "gh_comm_create: rank %d of %d", rank, world);
    *out = nullptr;
    int rc = need_rccl("gh_comm_create");
    if (rc) return rc;
    GH_HIP(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(id.internal, unique_id, GH_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    GH_RCCL(g_rccl.CommInitRank(&comm, world, id, rank), "gh_comm_create");
    gh_comm* c = new gh_comm();
    c->ctx = ctx; c->comm = comm; c->rank = rank; c->world = world; c->d_one = nullptr;
    if (hipMalloc((void**)&c->d_one, 256) != hipSuccess) {
        g_rccl.CommDestroy(comm);
        delete c;
        gh_set_error("gh_comm_create: hipMalloc failed");
        return GH_ERR_NOMEM;
    }
    *out = c;
    return GH_OK;
}

extern "C" void gh_comm_destroy(gh_comm* c) {
    if (!c) return;
    hipSetDevice(c->ctx->device);
    if (!c->aborted) gh_stream_wait(c->ctx, c, "gh_comm_destroy");     // (a hung collective aborts the communicator here)
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->d_one) hipFree(c->d_one);
    delete c;
}

extern "C" int gh_comm_count(const gh_comm* c) {
    if (!c || !g_rccl.CommCount || c->aborted) return 0;
    int n = 0;
    return g_rccl.CommCount(c->comm, &n) == ncclSuccess ? n : 0;
}

extern "C" int gh_comm_rank(const gh_comm* c) { return c ? c->rank : -1; }

gh_ctx* gh_comm_context(const gh_comm* c) { return c ? c->ctx : nullptr; }

extern "C" const char* gh_comm_library(void) {
    std::call_once(g_rccl_once, open_rccl);
    return g_rccl.handle ? g_rccl.path.c_str() : "";
}

extern "C" int gh_comm_version(void) {
    std::call_once(g_rccl_once, open_rccl);
    int v = 0;
    if (g_rccl.handle && g_rccl.GetVersion) g_rccl.GetVersion(&v);
    return v;
}

// internal: the collective every trainer path uses (gh_em.hip enqueues it between the statistics and the M-step)
int gh_comm_allreduce_enqueue(gh_comm* c, double* dev, int64_t n) {
    if (c->aborted) {
        gh_set_error("gh_stats_allreduce: the communicator has been aborted");
        return GH_ERR_COMM;
    }
    if (n <= 0) return GH_OK;
    GH_RCCL(g_rccl.AllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum, c->comm, c->ctx->stream), "gh_stats_allreduce");
    return GH_OK;
}

extern "C" int gh_stats_allreduce(gh_ctx* ctx, gh_comm* c, double* stats_dev, int64_t n) {
    GH_REQUIRE(ctx && c && (stats_dev || n == 0), "gh_stats_allreduce: NULL argument");
    GH_REQUIRE(c->ctx == ctx, "gh_stats_allreduce: the communicator belongs to another context");
    GH_REQUIRE(n >= 0, "gh_stats_allreduce: n = %lld", (long long)n);
    GH_HIP(hipSetDevice(ctx->device));
    return gh_comm_allreduce_enqueue(c, stats_dev, n);
}

extern "C" int gh_comm_barrier(gh_ctx* ctx, gh_comm* c) {
    GH_REQUIRE(ctx && c && c->ctx == ctx, "gh_comm_barrier: NULL argument / foreign context");
    GH_HIP(hipSetDevice(ctx->device));
    GH_HIP(hipMemsetAsync(c->d_one, 0, 8, ctx->stream));
    int rc = gh_comm_allreduce_enqueue(c, c->d_one, 1);
    if (rc) return rc;
    return gh_stream_wait(ctx, c, "gh_comm_barrier");
}

// host buffer in, reduced host buffer out (staged through the context's scratch): the lock-step trainer's cluster sums
// and the bench's max-over-ranks timing use it; the statistics of the EM path never take it (they are born on the device)
extern "C" int gh_comm_allreduce_host(gh_ctx* ctx, gh_comm* c, double* host_io, int64_t n, int op_max) {
    GH_REQUIRE(ctx && c && c->ctx == ctx && (host_io || n == 0), "gh_comm_allreduce_host: NULL argument / foreign context");
    if (n <= 0) return GH_OK;
    GH_HIP(hipSetDevice(ctx->device));
    void* base;
    int rc = gh_scratch(ctx, (size_t)n * 8, &base);
    if (rc) return rc;
    if (c->aborted) {
        gh_set_error("gh_comm_allreduce_host: the communicator has been aborted");
        return GH_ERR_COMM;
    }
    GH_HIP(hipMemcpyAsync(base, host_io, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    GH_RCCL(g_rccl.AllReduce(base, base, (size_t)n, ncclDouble, op_max ? ncclMax : ncclSum, c->comm, ctx->stream),
            "gh_comm_allreduce_host");
    // (the wait BEFORE the copy back: a device-to-host copy into pageable memory blocks inside hipMemcpyAsync until the
    //  stream gets there -- behind a collective a lost peer would turn that into a hang no deadline can end)
    const int rcw = gh_stream_wait(ctx, c, "gh_comm_allreduce_host");
    if (rcw) return rcw;
    GH_HIP(hipMemcpyAsync(host_io, base, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
    GH_HIP(hipStreamSynchronize(ctx->stream));
    return GH_OK;
}
