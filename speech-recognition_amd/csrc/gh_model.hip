// C ABI: packed emission model (gh_gmm), resident utterance batches (gh_batch), gh_loglik entry.
#include "gh_internal.h"
#include "gh_host.h"

// -------------------------------------------------------------------- model
extern "C" int gh_gmm_create(gh_ctx* ctx, int S, int M, int D, const double* mean, const double* var,
                             const double* weight, gh_gmm** out) {
    GH_REQUIRE(ctx && out && mean && var && weight, "gh_gmm_create: NULL argument");
    GH_REQUIRE(S > 0 && M > 0 && D > 0, "gh_gmm_create: S=%d M=%d D=%d must be positive", S, M, D);
    *out = nullptr;
    GH_HIP(hipSetDevice(ctx->device));
    const int G = S * M;
    const int KP = (D + 3) & ~3;
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
    UploadArena ar;
    ar.add(&g->dA64, g->hA); ar.add(&g->dB64, g->hB); ar.add(&g->dC64, g->hC);
    ar.add(&g->dA32, fA); ar.add(&g->dB32, fB); ar.add(&g->dC32, fC); ar.add(&g->dCen32, cen32);
    ar.add(&g->dMean, vmean); ar.add(&g->dIvar, ivar); ar.add(&g->dLogc, logc);
    ar.add(&g->dApk64, apk64); ar.add(&g->dCpk64, cpk64); ar.add(&g->dApk32, apk32); ar.add(&g->dCpk32, cpk32);
    const int rc = ar.commit(&g->d_arena);
    if (rc) {
        gh_gmm_destroy(g);
        return rc;
    }
    *out = g;
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
    for (int64_t u = 0; u < U; ++u) b->max_T = std::max(b->max_T, off[u + 1] - off[u]);
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
    if (rc) return rc;
    if (out_host && b->N > 0) {
        GH_HIP(hipMemcpyAsync(out_host, b->nll, (size_t)b->N * g->S * esz, hipMemcpyDeviceToHost, ctx->stream));
        GH_HIP(hipStreamSynchronize(ctx->stream));
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

