// C ABI: packed emission model (gh_gmm), resident utterance batches (gh_batch), gh_loglik entry.
#include "gh_internal.h"
#include <atomic>
#include "gh_host.h"

// -------------------------------------------------------------------- model
uint64_t gh_next_serial() {
    static std::atomic<uint64_t> counter{0};
    return ++counter;
}

extern "C" int gh_gmm_create(gh_ctx* ctx, int S, int M, int D, const double* mean, const double* var,
                             const double* weight, gh_gmm** out) {
    GH_REQUIRE(ctx && out && mean && var && weight, "gh_gmm_create: NULL argument");
    GH_REQUIRE(S > 0 && M > 0 && D > 0, "gh_gmm_create: S=%d M=%d D=%d must be positive", S, M, D);
    *out = nullptr;
    GH_HIP(hipSetDevice(ctx->device));
    const int G = S * M;
    // feature length padded to what the matrix-core kernels are instantiated for (k-steps KS = KP / 2 in {2, 4, 8, 12, 20,
    // 24, 32}: gh_loglik_mfma.hip; the pad columns are zero operands): every D <= 64 runs on the matrix cores -- until round
    // 4 only D with (D + 3) & ~3 in that list did, the rest (9-12, 17-20, 25-36, > 40) fell to the vector kernel
    int KP = (D + 3) & ~3;
    for (const int kp : {4, 8, 16, 24, 40, 48, 64})
        if (KP <= kp) { KP = kp; break; }
    gh_gmm* g = new gh_gmm();
    g->ctx = ctx;
    g->d_arena = nullptr;
    g->S = S; g->M = M; g->D = D; g->KP = KP;
    g->hA.assign((size_t)G * KP, 0.0);
    g->hB.assign((size_t)G * KP, 0.0);
    g->hC.assign(G, 0.0);
    std::vector<double> ivar((size_t)G * D), logc(G);
    const double log2pi = std::log(2.0 * M_PI);
    for (int i = 0; i < G; ++i) {
        double sum_logv = 0, sum_m2 = 0;
        for (int d = 0; d < D; ++d) {
            const double v = var[(size_t)i * D + d], mu = mean[(size_t)i * D + d];
            if (v == 0) {  // np.linalg.inv(diag(var)) raises LinAlgError on a zero pivot (hmm_state.py:17);
                           // NaN / negative variances pass and poison the likelihood, as in the reference
                delete g;
                gh_set_error("gh_gmm_create: var[%d,%d,%d]=%g (singular covariance)",
                             i / M, i % M, d, v);
                return GH_ERR_INVALID;
            }
            const double iv = 1.0 / v;
            ivar[(size_t)i * D + d] = iv;
            g->hA[(size_t)i * KP + d] = -0.5 * iv;
            g->hB[(size_t)i * KP + d] = mu * iv;
            sum_logv += std::log(v);
            sum_m2 += mu * mu * iv;
        }
        logc[i] = std::log(weight[i]) - 0.5 * (D * log2pi + sum_logv);  // log(0) = -inf: component off
        g->hC[i] = logc[i] - 0.5 * sum_m2;
    }
    // fp32 operands: packed for centred features (gh_internal.h, dCen32)
    std::vector<float> cen32(KP, 0.f);
    for (int d = 0; d < D; ++d) {
        long double acc = 0;
        int cnt = 0;
        for (int i = 0; i < G; ++i) {
            const double mu = mean[(size_t)i * D + d];
            if (std::isfinite(mu)) { acc += mu; ++cnt; }
        }
        cen32[d] = cnt ? (float)(double)(acc / cnt) : 0.f;
        if (!std::isfinite(cen32[d])) cen32[d] = 0.f;
    }
    std::vector<double> hB32((size_t)G * KP, 0.0), hC32(G, 0.0);   // fp64 master copies of the centred B and C
    for (int i = 0; i < G; ++i) {
        double sum_m2 = 0;
        for (int d = 0; d < D; ++d) {
            const double iv = ivar[(size_t)i * D + d], mu = mean[(size_t)i * D + d] - (double)cen32[d];
            hB32[(size_t)i * KP + d] = mu * iv;
            sum_m2 += mu * mu * iv;
        }
        hC32[i] = logc[i] - 0.5 * sum_m2;
    }
    std::vector<float> fA(g->hA.begin(), g->hA.end()), fB(hB32.begin(), hB32.end()), fC(hC32.begin(), hC32.end());
    // ---- MFMA operand packing (see gh_loglik_mfma.hip) ----
    int M_pad = 1;
    if (M <= 16) { while (M_pad < M) M_pad <<= 1; } else { M_pad = (M + 15) & ~15; }
    const int n_tiles = (S * M_pad + 15) / 16, KS = KP / 2;
    g->M_pad = M_pad;
    g->n_tiles = n_tiles;
    g->dApk64 = nullptr; g->dCpk64 = nullptr; g->dApk32 = nullptr; g->dCpk32 = nullptr; g->dCen32 = nullptr;
    // (+2 all-zero tiles: the kernel's run-ahead operand loads stay in bounds)
    std::vector<double> apk64((size_t)(n_tiles + 2) * KS * 64, 0.0), cpk64((size_t)(n_tiles + 2) * 16, GH_LSE_OFF64);
    std::vector<float> apk32(apk64.size(), 0.f), cpk32(cpk64.size(), GH_LSE_OFF32);
    for (int t = 0; t < n_tiles; ++t)
        for (int j = 0; j < 16; ++j) {  // j = natural position inside the tile
            const int gp = t * 16 + j, s = gp / M_pad, m = gp % M_pad;
            if (s >= S || m >= M) continue;  // padding component: switched off (C = OFF, P = 0)
            const int go = s * M + m;
            // scaled log domain; a weight-0 component (C = -inf) gets the finite OFF constant, NaN stays NaN
            const double c = g->hC[go], c32 = hC32[go];
            cpk64[gp] = (c == -INFINITY) ? GH_LSE_OFF64 : std::max(c * GH_LSE_SCALE64, GH_LSE_OFF64);
            cpk32[gp] = (c32 == -INFINITY) ? GH_LSE_OFF32 : (float)std::max(c32 * GH_LSE_SCALE32, (double)GH_LSE_OFF32);
            // accumulator row that makes lane group q = j/4 hold this component in register j%4:
            // f64 16x16x4: row = (lane>>4) + 4*reg  ->  row = j/4 + 4*(j%4); f32: row = 4*(lane>>4) + reg = j
            const int row64 = (j >> 2) + 4 * (j & 3), row32 = j;
            for (int ks = 0; ks < KS; ++ks)
                for (int kq = 0; kq < 4; ++kq) {
                    const int kk = 4 * ks + kq;
                    const double v = kk < KP ? g->hA[(size_t)go * KP + kk] : g->hB[(size_t)go * KP + kk - KP];
                    const double v32 = kk < KP ? v : hB32[(size_t)go * KP + kk - KP];
                    apk64[((size_t)t * KS + ks) * 64 + kq * 16 + row64] = v * GH_LSE_SCALE64;
                    apk32[((size_t)t * KS + ks) * 64 + kq * 16 + row32] = (float)(v32 * GH_LSE_SCALE32);
                }
        }
    std::vector<double> vmean(mean, mean + (size_t)G * D);
    std::vector<int> anypos(1, 0);
    for (int i = 0; i < G; ++i) if (logc[i] > 0) anypos[0] = 1;
    UploadArena ar;
    ar.add(&g->dA64, g->hA); ar.add(&g->dB64, g->hB); ar.add(&g->dC64, g->hC);
    ar.add(&g->dA32, fA); ar.add(&g->dB32, fB); ar.add(&g->dC32, fC); ar.add(&g->dCen32, cen32);
    ar.add(&g->dMean, vmean); ar.add(&g->dIvar, ivar); ar.add(&g->dLogc, logc); ar.add(&g->dAnyPos, anypos);
    g->any_pos_host = anypos[0];
    ar.add(&g->dApk64, apk64); ar.add(&g->dCpk64, cpk64); ar.add(&g->dApk32, apk32); ar.add(&g->dCpk32, cpk32);
    const int rc = ar.commit(&g->d_arena);
    if (rc) {
        gh_gmm_destroy(g);
        return rc;
    }
    g->serial = gh_next_serial();
    *out = g;
    return GH_OK;
}

// ---- the same packing on the device: a trainer re-estimates the model every iteration and the new parameters are
// born in HBM (M-step kernel); packing them on the host meant a D2H, ~0.2 ms of host arithmetic, an allocation and an
// H2D per EM iteration.  One thread per Gaussian writes the plain arrays, one thread per operand element the MFMA
// fragments; mixture / tile padding keeps what gh_gmm_create put there (P = 0, C = OFF).
// one block per dimension: the finite means of the G Gaussians summed by 1024 threads (strided, a shuffle tree per wave,
// the 16 wave sums in order) -- with 64 lanes the 32 768 Gaussians of the configs[3] model took 0.18 ms per model update
__global__ __launch_bounds__(1024) void gmm_centre_kernel(const double* __restrict__ mean, int G, int D, int KP, float* __restrict__ cen32,
                                                          int* __restrict__ any_pos) {
    __shared__ double s_acc[16];
    __shared__ int s_cnt[16];
    const int d = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (d == 0 && tid == 0) *any_pos = 0;      // (raised again by gmm_pack_plain_kernel, next on the stream)
    double acc = 0;
    int cnt = 0;
    if (d < D)
        for (int i = tid; i < G; i += 1024) {
            const double mu = mean[(size_t)i * D + d];
            if (mu - mu == 0.0) { acc += mu; ++cnt; }   // finite
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { acc += __shfl_xor(acc, o); cnt += __shfl_xor(cnt, o); }
    if (lane == 0) { s_acc[wave] = acc; s_cnt[wave] = cnt; }
    __syncthreads();
    if (tid == 0) {
        double a = 0;
        int n = 0;
        for (int w = 0; w < 16; ++w) { a += s_acc[w]; n += s_cnt[w]; }
        float c = n ? (float)(a / n) : 0.f;
        if (!(c - c == 0.f)) c = 0.f;
        cen32[d] = c;
    }
}

struct gmm_dev_view {
    int* any_pos;
    int S, M, D, KP, M_pad, n_tiles;
    double *A64, *B64, *C64, *Mean, *Ivar, *Logc, *Apk64, *Cpk64;
    float *A32, *B32, *C32, *Apk32, *Cpk32;
    const float* cen32;
};

__global__ void gmm_pack_plain_kernel(gmm_dev_view v, const double* __restrict__ mean, const double* __restrict__ var,
                                      const double* __restrict__ weight, int* __restrict__ flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int G = v.S * v.M;
    if (i >= G) return;
    const int D = v.D, KP = v.KP;
    const double log2pi = 1.8378770664093454836;
    double sum_logv = 0, sum_m2 = 0, sum_m2c = 0;
    for (int d = 0; d < D; ++d) {
        const double vv = var[(size_t)i * D + d], mu = mean[(size_t)i * D + d];
        if (vv == 0) atomicOr(flag, 16);     // singular covariance (np.linalg.inv raises, hmm_state.py:17)
        const double iv = 1.0 / vv;
        const double muc = mu - (double)v.cen32[d];
        v.Mean[(size_t)i * D + d] = mu;
        v.Ivar[(size_t)i * D + d] = iv;
        v.A64[(size_t)i * KP + d] = -0.5 * iv;
        v.B64[(size_t)i * KP + d] = mu * iv;
        v.A32[(size_t)i * KP + d] = (float)(-0.5 * iv);
        v.B32[(size_t)i * KP + d] = (float)(muc * iv);
        sum_logv += log(vv);
        sum_m2 += mu * mu * iv;
        sum_m2c += muc * muc * iv;
    }
    const double logc = log(weight[i]) - 0.5 * (D * log2pi + sum_logv);
    v.Logc[i] = logc;
    if (logc > 0) atomicOr(v.any_pos, 1);
    const double c = logc - 0.5 * sum_m2, c32 = logc - 0.5 * sum_m2c;
    v.C64[i] = c;
    v.C32[i] = (float)c32;
    // position of the component in the padded tile layout
    const int s = i / v.M, m = i - s * v.M;
    const int gp = s * v.M_pad + m;
    v.Cpk64[gp] = (c == -INFINITY) ? GH_LSE_OFF64 : fmax(c * GH_LSE_SCALE64, GH_LSE_OFF64) + (c != c ? c : 0.0);
    v.Cpk32[gp] = (c32 == -INFINITY) ? GH_LSE_OFF32 : (float)(fmax(c32 * GH_LSE_SCALE32, (double)GH_LSE_OFF32) + (c32 != c32 ? c32 : 0.0));
}

__global__ void gmm_pack_operands_kernel(gmm_dev_view v) {
    // one thread per (padded Gaussian gp, k index kk in [0, 2 KP))
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K2 = 2 * v.KP;
    const int64_t total = (int64_t)v.n_tiles * 16 * K2;
    if (idx >= total) return;
    const int gp = (int)(idx / K2), kk = (int)(idx - (int64_t)gp * K2);
    const int s = gp / v.M_pad, m = gp - s * v.M_pad;
    if (s >= v.S || m >= v.M) return;
    const int go = s * v.M + m;
    const int t = gp >> 4, j = gp & 15;
    const int KS = v.KP / 2, ks = kk >> 2, kq = kk & 3;
    const int row64 = (j >> 2) + 4 * (j & 3), row32 = j;
    const double a64 = kk < v.KP ? v.A64[(size_t)go * v.KP + kk] : v.B64[(size_t)go * v.KP + kk - v.KP];
    double a32 = a64;                   // the centred B of the fp32 operands, from its fp64 ingredients (as the host packs it)
    if (kk >= v.KP) {
        const int d = kk - v.KP;
        a32 = d < v.D ? (v.Mean[(size_t)go * v.D + d] - (double)v.cen32[d]) * v.Ivar[(size_t)go * v.D + d] : 0.0;
    }
    v.Apk64[((size_t)t * KS + ks) * 64 + kq * 16 + row64] = a64 * GH_LSE_SCALE64;
    v.Apk32[((size_t)t * KS + ks) * 64 + kq * 16 + row32] = (float)(a32 * GH_LSE_SCALE32);
}

// new parameters (device arrays [S,M,D], [S,M,D], [S,M]) -> every device array of the handle, on the context's stream.
// A zero variance raises bit 16 of *d_flag (a device int of the caller's).
int gh_gmm_update_dev(gh_ctx* ctx, gh_gmm* g, const double* d_mean, const double* d_var, const double* d_weight, int* d_flag) {
    const int G = g->S * g->M;
    hipStream_t st = ctx->stream;
    gmm_dev_view v;
    v.S = g->S; v.M = g->M; v.D = g->D; v.KP = g->KP; v.M_pad = g->M_pad; v.n_tiles = g->n_tiles;
    v.A64 = g->dA64; v.B64 = g->dB64; v.C64 = g->dC64; v.Mean = g->dMean; v.Ivar = g->dIvar; v.Logc = g->dLogc;
    v.Apk64 = g->dApk64; v.Cpk64 = g->dCpk64; v.A32 = g->dA32; v.B32 = g->dB32; v.C32 = g->dC32;
    v.Apk32 = g->dApk32; v.Cpk32 = g->dCpk32; v.cen32 = g->dCen32; v.any_pos = g->dAnyPos;
    g->any_pos_host = -1;      // (the flag is rewritten on the device: the host no longer knows it)
    hipLaunchKernelGGL(gmm_centre_kernel, dim3(g->KP), dim3(1024), 0, st, d_mean, G, g->D, g->KP, g->dCen32, g->dAnyPos);
    hipLaunchKernelGGL(gmm_pack_plain_kernel, dim3((G + 63) / 64), dim3(64), 0, st, v, d_mean, d_var, d_weight, d_flag);
    const int64_t total = (int64_t)g->n_tiles * 16 * 2 * g->KP;
    hipLaunchKernelGGL(gmm_pack_operands_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, v);
    GH_HIP(hipGetLastError());
    g->host_stale = true;
    g->serial = gh_next_serial();      // likelihood matrices computed before this point belong to an older model
    return GH_OK;
}

extern "C" int gh_gmm_update(gh_ctx* ctx, gh_gmm* g, const double* mean, const double* var, const double* weight) {
    GH_REQUIRE(ctx && g && mean && var && weight, "gh_gmm_update: NULL argument");
    GH_HIP(hipSetDevice(ctx->device));
    const size_t G = (size_t)g->S * g->M, nd = G * g->D;
    for (size_t i = 0; i < nd; ++i)
        GH_REQUIRE(var[i] != 0, "gh_gmm_update: var[%zu,%zu,%zu]=0 (singular covariance)", i / g->D / g->M, i / g->D % g->M, i % g->D);
    void* base;
    int rc = gh_scratch(ctx, (2 * nd + G) * 8, &base);
    if (rc) return rc;
    double* d = static_cast<double*>(base);
    GH_HIP(hipMemcpyAsync(d, mean, nd * 8, hipMemcpyHostToDevice, ctx->stream));
    GH_HIP(hipMemcpyAsync(d + nd, var, nd * 8, hipMemcpyHostToDevice, ctx->stream));
    GH_HIP(hipMemcpyAsync(d + 2 * nd, weight, G * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = gh_gmm_update_dev(ctx, g, d, d + nd, d + 2 * nd, ctx->d_flag);   // (zero variances were rejected above: the flag stays clear)
    if (rc) return rc;
    GH_HIP(hipStreamSynchronize(ctx->stream));   // the host arrays are the caller's, the staging area the context's
    return GH_OK;
}

extern "C" void gh_gmm_destroy(gh_gmm* g) {
    if (!g) return;
    hipSetDevice(g->ctx->device);
    hipFree(g->d_arena);
    delete g;
}

// -------------------------------------------------------------------- batch
static int batch_common(gh_ctx* ctx, gh_dtype dtype, int D, int64_t N, int64_t U, const int64_t* off,
                        gh_batch** out) {
    GH_REQUIRE(ctx && out && off, "gh_batch: NULL argument");
    GH_REQUIRE(dtype == GH_F32 || dtype == GH_F64, "gh_batch: bad dtype %d", (int)dtype);
    GH_REQUIRE(D > 0 && N >= 0 && U >= 0, "gh_batch: D=%d N=%lld U=%lld", D, (long long)N, (long long)U);
    GH_REQUIRE(off[0] == 0 && off[U] == N, "gh_batch: utt_offsets must run from 0 to N");
    for (int64_t u = 0; u < U; ++u)
        GH_REQUIRE(off[u + 1] >= off[u], "gh_batch: utt_offsets not monotone at %lld", (long long)u);
    GH_HIP(hipSetDevice(ctx->device));
    gh_batch* b = new gh_batch();
    b->ctx = ctx; b->dtype = dtype; b->D = D; b->N = N; b->U = U;
    b->feats = nullptr; b->owns_feats = false; b->nll = nullptr; b->nll_S = 0; b->d_offsets = nullptr;
    b->occ = nullptr;
    b->occ_S = 0;
    b->gam = nullptr;
    b->occ_valid = false;
    b->d_occ_states = nullptr;
    b->offsets.assign(off, off + U + 1);
    b->max_T = 0;
    for (int64_t u = 0; u < U; ++u) {
        b->max_T = std::max(b->max_T, off[u + 1] - off[u]);
        if (off[u + 1] - off[u] == 1) b->any_T1 = true;
    }
    b->perm.resize(U);
    std::iota(b->perm.begin(), b->perm.end(), (int64_t)0);
    std::stable_sort(b->perm.begin(), b->perm.end(), [&](int64_t x, int64_t y) {
        return off[x + 1] - off[x] > off[y + 1] - off[y];
    });
    b->d_perm = nullptr;
    int rc = upload(&b->d_offsets, b->offsets);
    if (!rc) rc = upload(&b->d_perm, b->perm);
    if (rc) { delete b; return rc; }
    *out = b;
    return GH_OK;
}

extern "C" int gh_batch_create(gh_ctx* ctx, gh_dtype dtype, int D, int64_t N, int64_t U, const void* feats,
                               const int64_t* off, gh_batch** out) {
    GH_REQUIRE(feats || N == 0, "gh_batch_create: feats is NULL");
    int rc = batch_common(ctx, dtype, D, N, U, off, out);
    if (rc) return rc;
    gh_batch* b = *out;
    const size_t bytes = (size_t)N * D * (dtype == GH_F64 ? 8 : 4);
    b->owns_feats = true;
    if (bytes) {
        hipError_t e = hipMalloc(&b->feats, bytes);
        if (e == hipSuccess) e = hipMemcpy(b->feats, feats, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            gh_set_error("gh_batch_create: %s", hipGetErrorString(e));
            gh_batch_destroy(b);
            *out = nullptr;
            return e == hipErrorOutOfMemory ? GH_ERR_NOMEM : GH_ERR_HIP;
        }
    }
    return GH_OK;
}

// fp32 on the wire, the batch's dtype in HBM: the upload is half the bytes of an fp64 one and a kernel widens it on the
// device (every value exactly: fp32 -> fp64 is lossless; what the caller gives up is the fp64 features' low 29 bits BEFORE
// the call).  pin != 0: the host buffer is page-locked for the copy (hipHostRegister): the copy then runs at the link's
// rate instead of through the runtime's pageable staging.
// page-locking of caller buffers, counted per base address: two host threads may upload from the same array at once
// (bench.py's two lanes), and registering a range twice -- or unregistering it under the other thread's copy -- is an error
#include <map>
#include <mutex>
static std::mutex g_pin_mutex;
static std::map<const void*, std::pair<size_t, int>> g_pinned;
static bool pin_acquire(const void* p, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_pin_mutex);
    auto it = g_pinned.find(p);
    if (it != g_pinned.end()) {
        if (it->second.first < bytes) return false;      // (a longer range at the same base: leave it pageable)
        ++it->second.second;
        return true;
    }
    if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
    g_pinned[p] = {bytes, 1};
    return true;
}
static void pin_release(const void* p, bool keep) {
    std::lock_guard<std::mutex> lk(g_pin_mutex);
    auto it = g_pinned.find(p);
    if (it == g_pinned.end()) return;
    if (--it->second.second == 0 && !keep) {
        hipHostUnregister(const_cast<void*>(p));
        g_pinned.erase(it);
    }
}

// pin == 2 leaves the caller's buffer page-locked after the call (registering 150 MB costs about what the faster copy
// saves: a caller that uploads from the same buffer again and again pays it once); gh_host_unpin ends that -- before the
// buffer is freed
extern "C" int gh_host_unpin(const void* host_ptr) {
    std::lock_guard<std::mutex> lk(g_pin_mutex);
    auto it = g_pinned.find(host_ptr);
    if (it == g_pinned.end() || it->second.second > 0) return GH_OK;
    hipHostUnregister(const_cast<void*>(host_ptr));
    g_pinned.erase(it);
    return GH_OK;
}

__global__ void widen_f32_kernel(const float* __restrict__ src, int64_t n, double* __restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = (double)src[i];
}

extern "C" int gh_batch_create_wire(gh_ctx* ctx, gh_dtype dtype, gh_dtype wire_dtype, int pin, int D, int64_t N, int64_t U,
                                    const void* feats, const int64_t* off, gh_batch** out) {
    GH_REQUIRE(feats || N == 0, "gh_batch_create_wire: feats is NULL");
    GH_REQUIRE(wire_dtype == dtype || (wire_dtype == GH_F32 && dtype == GH_F64),
               "gh_batch_create_wire: the wire format is the batch's dtype, or fp32 for an fp64 batch");
    int rc = batch_common(ctx, dtype, D, N, U, off, out);
    if (rc) return rc;
    gh_batch* b = *out;
    const size_t n = (size_t)N * D, wire_bytes = n * (wire_dtype == GH_F64 ? 8 : 4), bytes = n * (dtype == GH_F64 ? 8 : 4);
    b->owns_feats = true;
    if (!n) return GH_OK;
    auto fail = [&](hipError_t e, const char* what) {
        gh_set_error("gh_batch_create_wire: %s -> %s", what, hipGetErrorString(e));
        gh_batch_destroy(b);
        *out = nullptr;
        return e == hipErrorOutOfMemory ? GH_ERR_NOMEM : GH_ERR_HIP;
    };
    hipError_t e = hipMalloc(&b->feats, bytes);
    if (e != hipSuccess) return fail(e, "hipMalloc");
    void* dst = b->feats;
    if (wire_dtype != dtype) {
        rc = gh_scratch(ctx, wire_bytes, &dst);
        if (rc) { gh_batch_destroy(b); *out = nullptr; return rc; }
    }
    const bool pinned = pin && pin_acquire(feats, wire_bytes);
    e = hipMemcpyAsync(dst, feats, wire_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && wire_dtype != dtype) {
        hipLaunchKernelGGL(widen_f32_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, (size_t)ctx->n_cu * 32)), dim3(256), 0, ctx->stream,
                           (const float*)dst, (int64_t)n, (double*)b->feats);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (pinned) pin_release(feats, pin == 2);
    if (e != hipSuccess) return fail(e, "upload");
    return GH_OK;
}

extern "C" int gh_batch_wrap(gh_ctx* ctx, gh_dtype dtype, int D, int64_t N, int64_t U, void* feats_dev,
                             const int64_t* off, gh_batch** out) {
    GH_REQUIRE(feats_dev || N == 0, "gh_batch_wrap: feats_dev is NULL");
    int rc = batch_common(ctx, dtype, D, N, U, off, out);
    if (rc) return rc;
    (*out)->feats = feats_dev;
    return GH_OK;
}

// rows `idx` of a resident batch as a new resident batch (no host round trip): what the regrouping of continuous_train
// (continuous_speech.py:107-113: np.vstack of every state's segments) is on the device
template <typename T>
__global__ void gather_rows_kernel(const T* __restrict__ src, const int64_t* __restrict__ idx, int64_t n, int D, T* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    const int64_t r = i / D;
    dst[i] = src[idx[r] * D + (i - r * D)];
}

extern "C" int gh_batch_gather(gh_ctx* ctx, const gh_batch* src, const int64_t* idx, int64_t n, int64_t U, const int64_t* off,
                               gh_batch** out) {
    GH_REQUIRE(ctx && src && out && (idx || n == 0), "gh_batch_gather: NULL argument");
    for (int64_t i = 0; i < n; ++i) GH_REQUIRE(idx[i] >= 0 && idx[i] < src->N, "gh_batch_gather: row %lld out of range", (long long)idx[i]);
    int rc = batch_common(ctx, src->dtype, src->D, n, U, off, out);
    if (rc) return rc;
    gh_batch* b = *out;
    const size_t es = src->dtype == GH_F64 ? 8 : 4;
    b->owns_feats = true;
    if (n > 0) {
        int64_t* d_idx = nullptr;
        hipError_t e = hipMalloc(&b->feats, (size_t)n * src->D * es);
        if (e == hipSuccess) e = hipMalloc((void**)&d_idx, (size_t)n * 8);
        if (e == hipSuccess) e = hipMemcpyAsync(d_idx, idx, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            const int64_t total = n * src->D;
            const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
            if (src->dtype == GH_F64) hipLaunchKernelGGL(gather_rows_kernel<double>, grid, blk, 0, ctx->stream, (const double*)src->feats, d_idx, n, src->D, (double*)b->feats);
            else hipLaunchKernelGGL(gather_rows_kernel<float>, grid, blk, 0, ctx->stream, (const float*)src->feats, d_idx, n, src->D, (float*)b->feats);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (d_idx) hipFree(d_idx);
        if (e != hipSuccess) {
            gh_set_error("gh_batch_gather: %s", hipGetErrorString(e));
            gh_batch_destroy(b);
            *out = nullptr;
            return e == hipErrorOutOfMemory ? GH_ERR_NOMEM : GH_ERR_HIP;
        }
    }
    return GH_OK;
}

// Runs of consecutive rows of a resident batch as a new resident batch: run r = rows [start[r], start[r] + len[r]) of
// `src`, copied to rows [dest[r], ...) of the result (n = the result's rows; the runs must tile it).  What gh_batch_gather
// does row by row from 8 bytes of index per row, from 24 bytes per run -- the regrouped frames of continuous_train are
// ~20-frame runs (gh_align_runs).  One workgroup per run: a contiguous copy.
template <typename T>
__global__ __launch_bounds__(256) void gather_runs_kernel(const T* __restrict__ src, const int64_t* __restrict__ runs /*[R,3]*/, int D,
                                                          T* __restrict__ dst) {
    const int64_t* r = runs + (int64_t)blockIdx.x * 3;
    const T* s = src + r[0] * D;
    T* d = dst + r[2] * D;
    const int64_t total = r[1] * D;
    for (int64_t i = threadIdx.x; i < total; i += 256) d[i] = s[i];
}

extern "C" int gh_batch_gather_runs(gh_ctx* ctx, const gh_batch* src, int64_t n_runs, const int64_t* start, const int64_t* len,
                                    const int64_t* dest, int64_t n, int64_t U, const int64_t* off, gh_batch** out) {
    GH_REQUIRE(ctx && src && out && ((start && len && dest) || n_runs == 0) && n_runs >= 0 && n >= 0, "gh_batch_gather_runs: NULL argument");
    int64_t covered = 0;
    for (int64_t r = 0; r < n_runs; ++r) {
        GH_REQUIRE(len[r] >= 0 && start[r] >= 0 && start[r] + len[r] <= src->N && dest[r] >= 0 && dest[r] + len[r] <= n,
                   "gh_batch_gather_runs: run %lld out of range", (long long)r);
        covered += len[r];
    }
    GH_REQUIRE(covered == n, "gh_batch_gather_runs: the runs hold %lld rows, the result %lld", (long long)covered, (long long)n);
    int rc = batch_common(ctx, src->dtype, src->D, n, U, off, out);
    if (rc) return rc;
    gh_batch* b = *out;
    const size_t es = src->dtype == GH_F64 ? 8 : 4;
    b->owns_feats = true;
    if (n > 0) {
        std::vector<int64_t> tab((size_t)n_runs * 3);
        for (int64_t r = 0; r < n_runs; ++r) { tab[3 * r] = start[r]; tab[3 * r + 1] = len[r]; tab[3 * r + 2] = dest[r]; }
        int64_t* d_tab = nullptr;
        hipError_t e = hipMalloc(&b->feats, (size_t)n * src->D * es);
        if (e == hipSuccess) e = hipMalloc((void**)&d_tab, tab.size() * 8);
        if (e == hipSuccess) e = hipMemcpyAsync(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            const dim3 grid((unsigned)n_runs), blk(256);
            if (src->dtype == GH_F64) hipLaunchKernelGGL(gather_runs_kernel<double>, grid, blk, 0, ctx->stream, (const double*)src->feats, d_tab, src->D, (double*)b->feats);
            else hipLaunchKernelGGL(gather_runs_kernel<float>, grid, blk, 0, ctx->stream, (const float*)src->feats, d_tab, src->D, (float*)b->feats);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (d_tab) hipFree(d_tab);
        if (e != hipSuccess) {
            gh_set_error("gh_batch_gather_runs: %s", hipGetErrorString(e));
            gh_batch_destroy(b);
            *out = nullptr;
            return e == hipErrorOutOfMemory ? GH_ERR_NOMEM : GH_ERR_HIP;
        }
    }
    return GH_OK;
}

// `reps` copies of a resident batch back to back (device-to-device): a large batch from a small upload
extern "C" int gh_batch_tile(gh_ctx* ctx, const gh_batch* src, int reps, gh_batch** out) {
    GH_REQUIRE(ctx && src && out && reps >= 1, "gh_batch_tile: NULL argument / reps=%d", reps);
    const int64_t U = src->U * reps, N = src->N * reps;
    std::vector<int64_t> off((size_t)U + 1, 0);
    for (int r = 0; r < reps; ++r)
        for (int64_t u = 0; u < src->U; ++u) off[(size_t)r * src->U + u + 1] = (int64_t)r * src->N + src->offsets[u + 1];
    int rc = batch_common(ctx, src->dtype, src->D, N, U, off.data(), out);
    if (rc) return rc;
    gh_batch* b = *out;
    const size_t bytes = (size_t)src->N * src->D * (src->dtype == GH_F64 ? 8 : 4);
    b->owns_feats = true;
    if (bytes) {
        hipError_t e = hipMalloc(&b->feats, bytes * reps);
        for (int r = 0; e == hipSuccess && r < reps; ++r)
            e = hipMemcpyAsync((char*)b->feats + (size_t)r * bytes, src->feats, bytes, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            gh_set_error("gh_batch_tile: %s", hipGetErrorString(e));
            gh_batch_destroy(b);
            *out = nullptr;
            return e == hipErrorOutOfMemory ? GH_ERR_NOMEM : GH_ERR_HIP;
        }
    }
    return GH_OK;
}

// x += scale * z, z ~ N(0, 1) independent per feature, from a counter-based generator keyed on (seed, element index):
// every copy of a tiled batch becomes an utterance of its own without a trip through the host (bench.py's configs[4]
// legs: 125 000 utterances from 5 000 synthesised ones).  Deterministic for a given seed; likelihoods / occupancies held
// by the batch are stale afterwards.
template <typename T>
__global__ void batch_jitter_kernel(T* __restrict__ x, int64_t n, uint64_t seed, double scale) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1);      // splitmix64 of the element's counter
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const double u1 = ((double)(uint32_t)(z >> 32) + 1.0) * (1.0 / 4294967296.0);   // (0, 1]
    const double u2 = (double)(uint32_t)z * (1.0 / 4294967296.0);
    const double g = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);            // Box-Muller
    x[i] = (T)((double)x[i] + scale * g);
}

extern "C" int gh_batch_jitter(gh_ctx* ctx, gh_batch* b, uint64_t seed, double scale) {
    GH_REQUIRE(ctx && b && b->owns_feats, "gh_batch_jitter: NULL argument / the batch wraps memory it does not own");
    GH_HIP(hipSetDevice(ctx->device));
    const int64_t n = b->N * (int64_t)b->D;
    if (n == 0) return GH_OK;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    if (b->dtype == GH_F64) hipLaunchKernelGGL(batch_jitter_kernel<double>, grid, blk, 0, ctx->stream, (double*)b->feats, n, seed, scale);
    else hipLaunchKernelGGL(batch_jitter_kernel<float>, grid, blk, 0, ctx->stream, (float*)b->feats, n, seed, scale);
    GH_HIP(hipGetLastError());
    b->nll_serial = 0;
    b->occ_valid = false;
    return GH_OK;
}

extern "C" void gh_batch_destroy(gh_batch* b) {
    if (!b) return;
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    if (b->owns_feats && b->feats) hipFree(b->feats);
    if (b->nll) hipFree(b->nll);
    if (b->occ) hipFree(b->occ);
    if (b->gam) hipFree(b->gam);
    if (b->d_occ_states) hipFree(b->d_occ_states);
    if (b->d_offsets) hipFree(b->d_offsets);
    if (b->d_perm) hipFree(b->d_perm);
    if (b->d_clusters) hipFree(b->d_clusters);
    delete b;
}

// ------------------------------------------------------------------- loglik
// ---- the linear-domain underflow rule, exactly (gh_ctx_set_compat bit 0).  GMM.evaluate sums w * pdf with
// pdf = norm * np.exp(-q/2) (hmm_state.py:36-45,114-120): a term is 0 once exp(-q/2) is -- below ln 2^-1075, WHATEVER the
// normaliser -- or once the product is.  The likelihood kernels test the largest total logarithm a = log(w norm) - q/2
// against ln 2^-1075, which is the whole rule while every log(w norm) <= 0 (ordinary variances).  With a component whose
// w * norm > 1 (variances below ~1/2pi on average) a frame can have a > ln 2^-1075 and -q/2 below it: this pass finds the
// entries in that band (cost within max(log(w norm), 0) + log M of the threshold: next to none) and re-tests them per component.
template <typename T>
__global__ void loglik_underflow_fix_kernel(const T* __restrict__ X, int64_t N, int S, int M, int D, const double* __restrict__ mean,
                                            const double* __restrict__ ivar, const double* __restrict__ logc,
                                            const float* __restrict__ cen32, const int* __restrict__ any_pos, T* __restrict__ nll) {
    if (*any_pos == 0) return;     // ordinary models: a few thousand threads read one word and leave (a fixed, small grid:
                                   // one thread per entry made the launch itself cost 0.4 ms at 5e8 entries)
    const double thr = 745.1332191019412;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < N * S; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = idx / S;
        const int s = (int)(idx - t * S);
        double P = 0.0;
        for (int m = 0; m < M; ++m) P = fmax(P, logc[s * M + m]);     // (-inf for a switched-off component)
        if (!(P > 0.0)) continue;
        const double v = (double)nll[idx];
        // a term whose exponent underflows has a = log(w norm) - q/2 < P - thr; with all M terms like that the cost
        // -logsumexp(a) can still be as low as thr - P - log M (the band used to stop at thr - P)
        if (!(v > thr - P - log((double)M)) || v == INFINITY) continue;
        bool survives = false;
        for (int m = 0; m < M && !survives; ++m) {
            const double lc = logc[s * M + m];
            double q2 = 0.0;
            for (int d = 0; d < D; ++d) {
                const double dx = (double)X[t * D + d] - mean[(size_t)(s * M + m) * D + d];
                q2 += dx * dx * ivar[(size_t)(s * M + m) * D + d];
            }
            q2 *= 0.5;
            if (!(q2 > thr) && !(q2 - lc > thr)) survives = true;     // this term survives in the reference: the cost is finite
        }
        if (!survives) nll[idx] = (T)INFINITY;
    }
}

int gh_loglik_underflow_fix(gh_ctx* ctx, const gh_gmm* g, gh_batch* b) {
    if (!(ctx->compat & 1) || b->N == 0 || !b->nll || !g->dAnyPos) return GH_OK;
    // a model packed on the host whose log-normalisers are all <= 0 cannot trip the rule: no launch at all (the
    // early-exit launch still cost 42 us of the 1.26 ms headline step)
    if (g->any_pos_host == 0) return GH_OK;
    const int64_t total = b->N * (int64_t)g->S;
    const unsigned grid = (unsigned)std::min<int64_t>((total + 255) / 256, (int64_t)ctx->n_cu * 16);     // grid-stride loop
    if (b->dtype == GH_F64)
        hipLaunchKernelGGL(loglik_underflow_fix_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, (const double*)b->feats, b->N, g->S,
                           g->M, g->D, g->dMean, g->dIvar, g->dLogc, g->dCen32, g->dAnyPos, (double*)b->nll);
    else
        hipLaunchKernelGGL(loglik_underflow_fix_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, (const float*)b->feats, b->N, g->S,
                           g->M, g->D, g->dMean, g->dIvar, g->dLogc, g->dCen32, g->dAnyPos, (float*)b->nll);
    GH_HIP(hipGetLastError());
    return GH_OK;
}

extern "C" int gh_loglik(gh_ctx* ctx, const gh_gmm* g, gh_batch* b, void* out_host) {
    GH_REQUIRE(ctx && g && b, "gh_loglik: NULL argument");
    GH_REQUIRE(g->D == b->D, "gh_loglik: feature dim %d != model dim %d (hmm_state.py:45)", b->D, g->D);
    GH_HIP(hipSetDevice(ctx->device));
    const size_t esz = b->dtype == GH_F64 ? 8 : 4;
    if (b->nll && b->nll_S != g->S) {
        GH_HIP(hipStreamSynchronize(ctx->stream));
        GH_HIP(hipFree(b->nll));
        b->nll = nullptr;
    }
    if (!b->nll && b->N > 0) {
        GH_HIP(hipMalloc(&b->nll, (size_t)b->N * g->S * esz));
        b->nll_S = g->S;
    }
    // matrix-core kernel when the shape is covered, vector kernel otherwise (GMMHMM_LOGLIK=valu forces it)
    static const bool force_valu = [] { const char* e = getenv("GMMHMM_LOGLIK"); return e && !strcmp(e, "valu"); }();
    int rc = force_valu ? 1 : gh_launch_loglik_mfma(ctx, g, b);
    if (rc == 1) rc = gh_launch_loglik(ctx, g, b);
    if (!rc) rc = gh_loglik_underflow_fix(ctx, g, b);
    if (rc) return rc;
    if (out_host && b->N > 0) {
        GH_HIP(hipMemcpyAsync(out_host, b->nll, (size_t)b->N * g->S * esz, hipMemcpyDeviceToHost, ctx->stream));
        GH_HIP(hipStreamSynchronize(ctx->stream));
    }
    return GH_OK;
}

// the resident [N, S] likelihood matrix of a batch, (re)allocated for S states; zero: entries nobody computes stay defined
int gh_batch_ensure_nll(gh_ctx* ctx, gh_batch* b, int S, bool zero) {
    const size_t esz = b->dtype == GH_F64 ? 8 : 4;
    if (b->nll && b->nll_S != S) {
        GH_HIP(hipStreamSynchronize(ctx->stream));
        GH_HIP(hipFree(b->nll));
        b->nll = nullptr;
    }
    if (!b->nll && b->N > 0) {
        GH_HIP(hipMalloc(&b->nll, (size_t)b->N * S * esz));
        if (zero) GH_HIP(hipMemsetAsync(b->nll, 0, (size_t)b->N * S * esz, ctx->stream));
        b->nll_S = S;
    }
    return GH_OK;
}

static int loglik_subset_impl(gh_ctx* ctx, const gh_gmm* g, gh_batch* b, const int32_t* state_lo, const int32_t* state_hi,
                              const int64_t* range_off, const char* who) {
    GH_REQUIRE(ctx && g && b && state_lo && state_hi, "%s: NULL argument", who);
    GH_REQUIRE(g->D == b->D, "%s: feature dim %d != model dim %d (hmm_state.py:45)", who, b->D, g->D);
    if (range_off) {
        GH_REQUIRE(range_off[0] == 0, "%s: range_off must start at 0", who);
        for (int64_t u = 0; u < b->U; ++u) GH_REQUIRE(range_off[u + 1] >= range_off[u], "%s: range_off not monotone at %lld", who, (long long)u);
    }
    GH_HIP(hipSetDevice(ctx->device));
    const size_t esz = b->dtype == GH_F64 ? 8 : 4;
    if (b->nll && b->nll_S != g->S) {
        GH_HIP(hipStreamSynchronize(ctx->stream));
        GH_HIP(hipFree(b->nll));
        b->nll = nullptr;
    }
    if (!b->nll && b->N > 0) {
        GH_HIP(hipMalloc(&b->nll, (size_t)b->N * g->S * esz));
        GH_HIP(hipMemsetAsync(b->nll, 0, (size_t)b->N * g->S * esz, ctx->stream));   // entries outside the ranges stay defined
        b->nll_S = g->S;
    }
    int rc = gh_launch_loglik_mfma(ctx, g, b, state_lo, state_hi, range_off);
    if (rc == 1) return gh_loglik(ctx, g, b, nullptr);   // shape / range not covered: the full matrix is a superset
    if (!rc) rc = gh_loglik_underflow_fix(ctx, g, b);    // (entries outside the ranges: zeros / stale values, nobody reads them)
    return rc;
}

extern "C" int gh_loglik_subset(gh_ctx* ctx, const gh_gmm* g, gh_batch* b, const int32_t* state_lo, const int32_t* state_hi) {
    return loglik_subset_impl(ctx, g, b, state_lo, state_hi, nullptr, "gh_loglik_subset");
}

extern "C" int gh_loglik_sets(gh_ctx* ctx, const gh_gmm* g, gh_batch* b, const int64_t* range_off, const int32_t* state_lo,
                              const int32_t* state_hi) {
    GH_REQUIRE(range_off, "gh_loglik_sets: NULL argument");
    return loglik_subset_impl(ctx, g, b, state_lo, state_hi, range_off, "gh_loglik_sets");
}

extern "C" void* gh_loglik_dev_ptr(gh_batch* b) { return b ? b->nll : nullptr; }

extern "C" int gh_loglik_fetch(gh_ctx* ctx, const gh_batch* b, void* out_host) {
    GH_REQUIRE(ctx && b && out_host, "gh_loglik_fetch: NULL argument");
    GH_REQUIRE(b->nll || b->N == 0, "gh_loglik_fetch: gh_loglik has not been run on this batch");
    GH_HIP(hipSetDevice(ctx->device));
    if (b->N == 0) return GH_OK;
    GH_HIP(hipMemcpyAsync(out_host, b->nll, (size_t)b->N * b->nll_S * (b->dtype == GH_F64 ? 8 : 4), hipMemcpyDeviceToHost,
                          ctx->stream));
    GH_HIP(hipStreamSynchronize(ctx->stream));
    return GH_OK;
}

