// C ABI: error reporting, per-GPU context, scratch / pinned arenas.
#include "gh_internal.h"
#include "gh_host.h"

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;

void gh_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

extern "C" const char* gh_last_error(void) { return g_err.c_str(); }
extern "C" int gh_version(void) { return 1; }

// ------------------------------------------------------------------ context
extern "C" int gh_ctx_create(int device, gh_ctx** out) {
    GH_REQUIRE(out, "gh_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        gh_set_error("gh_ctx_create: no HIP device (%s)", hipGetErrorString(e));
        return GH_ERR_NODEVICE;
    }
    GH_REQUIRE(device >= 0 && device < n, "gh_ctx_create: device %d out of range [0,%d)", device, n);
    GH_HIP(hipSetDevice(device));
    gh_ctx* c = new gh_ctx();
    c->device = device;
    c->scratch = nullptr;
    c->scratch_bytes = 0;
    c->pinned = nullptr;
    c->pinned_bytes = 0;
    c->last_chunks = 0;
    c->compat = 1;     // the reference's linear-domain underflow rule (gh_ctx_set_compat); GMMHMM_COMPAT=0: log domain throughout
    if (const char* e = getenv("GMMHMM_COMPAT")) c->compat = strstr(e, "underflow") ? 1 : atoi(e);
    if (const char* e = getenv("GMMHMM_LSE")) if (!strcmp(e, "f32exp")) c->compat |= 2;   // fp32 exponentials in the fp64 log-sum-exp
    hipDeviceProp_t prop;
    GH_HIP(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount;
    GH_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    GH_HIP(hipMalloc((void**)&c->d_flag, sizeof(int)));
    GH_HIP(hipMemset(c->d_flag, 0, sizeof(int)));
    {   // tables of the fp64 exp/log used by the likelihood epilogue, computed in long double
        double t[384];
        for (int j = 0; j < 128; ++j) {
            t[j] = (double)exp2l((long double)j / 128.0L);
            const double inv = (double)(1.0L / (0.5L + ((long double)j + 0.5L) / 256.0L));
            t[128 + j] = inv;
            t[256 + j] = (double)(-logl((long double)inv) * (long double)GH_LSE_SCALE64);  // scaled log domain
        }
        GH_HIP(hipMalloc((void**)&c->d_fp64_tables, sizeof t));
        GH_HIP(hipMemcpy(c->d_fp64_tables, t, sizeof t, hipMemcpyHostToDevice));
    }
    *out = c;
    return GH_OK;
}

extern "C" void gh_ctx_destroy(gh_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->scratch) hipFree(c->scratch);
    if (c->pinned) hipHostFree(c->pinned);
    if (c->fit_arena) hipFree(c->fit_arena);
    if (c->fit_pin) hipHostFree(c->fit_pin);
    if (c->fit_act) hipHostFree(c->fit_act);
    hipFree(c->d_flag);
    hipFree(c->d_fp64_tables);
    hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int gh_ctx_sync(gh_ctx* c) {
    GH_REQUIRE(c, "gh_ctx_sync: ctx is NULL");
    GH_HIP(hipStreamSynchronize(c->stream));
    return GH_OK;
}

extern "C" int gh_device_sync(gh_ctx* c) {
    GH_REQUIRE(c, "gh_device_sync: ctx is NULL");
    GH_HIP(hipSetDevice(c->device));
    GH_HIP(hipDeviceSynchronize());
    return GH_OK;
}

extern "C" int gh_ctx_set_compat(gh_ctx* c, int flags) {
    GH_REQUIRE(c, "gh_ctx_set_compat: ctx is NULL");
    c->compat = flags;
    return GH_OK;
}

extern "C" int gh_ctx_last_chunks(const gh_ctx* c) { return c ? c->last_chunks : 0; }

extern "C" void* gh_ctx_stream(gh_ctx* c) { return c ? (void*)c->stream : nullptr; }

extern "C" int gh_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

// ------------------------------------------------------------------- events
extern "C" int gh_event_create(gh_ctx* c, void** out_event) {
    GH_REQUIRE(c && out_event, "gh_event_create: NULL argument");
    GH_HIP(hipSetDevice(c->device));
    hipEvent_t e;
    GH_HIP(hipEventCreate(&e));
    *out_event = (void*)e;
    return GH_OK;
}

extern "C" void gh_event_destroy(void* event) {
    if (event) hipEventDestroy((hipEvent_t)event);
}

extern "C" int gh_event_record(gh_ctx* c, void* event) {
    GH_REQUIRE(c && event, "gh_event_record: NULL argument");
    GH_HIP(hipEventRecord((hipEvent_t)event, c->stream));
    return GH_OK;
}

extern "C" int gh_ctx_wait_event(gh_ctx* c, void* event) {
    GH_REQUIRE(c && event, "gh_ctx_wait_event: NULL argument");
    GH_HIP(hipStreamWaitEvent(c->stream, (hipEvent_t)event, 0));
    return GH_OK;
}

extern "C" int gh_event_elapsed_ms(void* start, void* stop, float* out_ms) {
    GH_REQUIRE(start && stop && out_ms, "gh_event_elapsed_ms: NULL argument");
    GH_HIP(hipEventSynchronize((hipEvent_t)stop));
    GH_HIP(hipEventElapsedTime(out_ms, (hipEvent_t)start, (hipEvent_t)stop));
    return GH_OK;
}

// Bytes of DP scratch (back-pointers / decision words, alpha columns) one launch may take; larger batches are chunked.
// GMMHMM_SCRATCH_BUDGET=<bytes>[K|M|G] (read at every call: the tests force several chunks with it); else a quarter of
// what is free on the device (counting the arena this context already holds, which is reused), within [256 MiB, 24 GiB]:
// several contexts or several ranks on one GPU each take their share instead of a fixed 24 GiB.
size_t gh_scratch_budget(gh_ctx* ctx, bool fresh) {
    if (const char* e = getenv("GMMHMM_SCRATCH_BUDGET")) {
        char* end = nullptr;
        double v = strtod(e, &end);
        if (end && (*end == 'K' || *end == 'k')) v *= 1024.0;
        else if (end && (*end == 'M' || *end == 'm')) v *= 1048576.0;
        else if (end && (*end == 'G' || *end == 'g')) v *= 1073741824.0;
        if (v >= 1.0) return (size_t)v;
    }
    if (!fresh && ctx->budget_cache && ++ctx->budget_age < 64) return ctx->budget_cache;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t)4 << 30;
    const size_t avail = free_b + ctx->scratch_bytes;
    ctx->budget_cache = std::min((size_t)24 << 30, std::max((size_t)256 << 20, avail / 4));
    ctx->budget_age = 0;
    return ctx->budget_cache;
}

int gh_scratch(gh_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->scratch_bytes) {
        GH_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) GH_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        size_t want = bytes + bytes / 8 + (1u << 20);
        ctx->budget_cache = 0;                     // (the arena changes size: the next budget comes from the driver again)
        GH_HIP(hipMalloc(&ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return GH_OK;
}

int gh_pinned(gh_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->pinned_bytes) {
        GH_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->pinned) GH_HIP(hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        GH_HIP(hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault));
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return GH_OK;
}

