// Host side of the C ABI: handles, packing, launch orchestration.
#include "gh_internal.h"
#include "gh_viterbi.h"
#include "gh_dtw.h"
#include "gh_fb.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

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
            t[256 + j] = (double)(-logl((long double)inv));
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

extern "C" void* gh_ctx_stream(gh_ctx* c) { return c ? (void*)c->stream : nullptr; }

int gh_scratch(gh_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->scratch_bytes) {
        GH_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) GH_HIP(hipFree(ctx->scratch));
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        size_t want = bytes + bytes / 8 + (1u << 20);
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

namespace {

// carve 256-byte aligned pieces out of the context scratch
struct Carver {
    size_t total = 0;
    std::vector<std::pair<void**, size_t>> items;  // (destination pointer, offset)
    template <typename T> void add(T** dst, size_t count) {
        items.push_back({reinterpret_cast<void**>(dst), total});
        total += (count * sizeof(T) + 255) & ~size_t(255);
    }
    int commit(gh_ctx* ctx) {
        void* base = nullptr;
        int rc = gh_scratch(ctx, total ? total : 256, &base);
        if (rc) return rc;
        for (auto& it : items) *it.first = static_cast<char*>(base) + it.second;
        return GH_OK;
    }
};

template <typename T> int upload(T** dst, const std::vector<T>& src) {
    *dst = nullptr;
    if (src.empty()) return GH_OK;
    GH_HIP(hipMalloc((void**)dst, src.size() * sizeof(T)));
    GH_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return GH_OK;
}

}  // namespace

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
    std::vector<float> fA(g->hA.begin(), g->hA.end()), fB(g->hB.begin(), g->hB.end()),
        fC(g->hC.begin(), g->hC.end());
    // ---- MFMA operand packing (see gh_loglik_mfma.hip) ----
    int M_pad = 1;
    if (M <= 16) { while (M_pad < M) M_pad <<= 1; } else { M_pad = (M + 15) & ~15; }
    const int n_tiles = (S * M_pad + 15) / 16, KS = KP / 2;
    g->M_pad = M_pad;
    g->n_tiles = n_tiles;
    g->dApk64 = nullptr; g->dCpk64 = nullptr; g->dApk32 = nullptr; g->dCpk32 = nullptr;
    // (+2 all-zero tiles: the kernel's run-ahead operand loads stay in bounds)
    std::vector<double> apk64((size_t)(n_tiles + 2) * KS * 64, 0.0), cpk64((size_t)(n_tiles + 2) * 16, -INFINITY);
    std::vector<float> apk32(apk64.size(), 0.f), cpk32(cpk64.size(), -INFINITY);
    for (int t = 0; t < n_tiles; ++t)
        for (int j = 0; j < 16; ++j) {  // j = natural position inside the tile
            const int gp = t * 16 + j, s = gp / M_pad, m = gp % M_pad;
            if (s >= S || m >= M) continue;  // padding component: weight 0 (C = -inf, P = 0)
            const int go = s * M + m;
            cpk64[gp] = g->hC[go];
            cpk32[gp] = (float)g->hC[go];
            // accumulator row that makes lane group q = j/4 hold this component in register j%4:
            // f64 16x16x4: row = (lane>>4) + 4*reg  ->  row = j/4 + 4*(j%4); f32: row = 4*(lane>>4) + reg = j
            const int row64 = (j >> 2) + 4 * (j & 3), row32 = j;
            for (int ks = 0; ks < KS; ++ks)
                for (int kq = 0; kq < 4; ++kq) {
                    const int kk = 4 * ks + kq;
                    const double v = kk < KP ? g->hA[(size_t)go * KP + kk] : g->hB[(size_t)go * KP + kk - KP];
                    apk64[((size_t)t * KS + ks) * 64 + kq * 16 + row64] = v;
                    apk32[((size_t)t * KS + ks) * 64 + kq * 16 + row32] = (float)v;
                }
        }
    std::vector<double> vmean(mean, mean + (size_t)G * D);
    int rc = GH_OK;
    if ((rc = upload(&g->dA64, g->hA)) || (rc = upload(&g->dB64, g->hB)) || (rc = upload(&g->dC64, g->hC)) ||
        (rc = upload(&g->dA32, fA)) || (rc = upload(&g->dB32, fB)) || (rc = upload(&g->dC32, fC)) ||
        (rc = upload(&g->dMean, vmean)) || (rc = upload(&g->dIvar, ivar)) || (rc = upload(&g->dLogc, logc)) ||
        (rc = upload(&g->dApk64, apk64)) || (rc = upload(&g->dCpk64, cpk64)) ||
        (rc = upload(&g->dApk32, apk32)) || (rc = upload(&g->dCpk32, cpk32))) {
        gh_gmm_destroy(g);
        return rc;
    }
    *out = g;
    return GH_OK;
}

extern "C" void gh_gmm_destroy(gh_gmm* g) {
    if (!g) return;
    hipSetDevice(g->ctx->device);
    hipFree(g->dA64); hipFree(g->dB64); hipFree(g->dC64);
    hipFree(g->dA32); hipFree(g->dB32); hipFree(g->dC32);
    hipFree(g->dMean); hipFree(g->dIvar); hipFree(g->dLogc);
    hipFree(g->dApk64); hipFree(g->dCpk64); hipFree(g->dApk32); hipFree(g->dCpk32);
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

extern "C" void gh_batch_destroy(gh_batch* b) {
    if (!b) return;
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    if (b->owns_feats && b->feats) hipFree(b->feats);
    if (b->nll) hipFree(b->nll);
    if (b->occ) hipFree(b->occ);
    if (b->d_offsets) hipFree(b->d_offsets);
    if (b->d_perm) hipFree(b->d_perm);
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

extern "C" void* gh_loglik_dev_ptr(gh_batch* b) { return b ? b->nll : nullptr; }

// ----------------------------------------------------------------- lattices
extern "C" int gh_lattices_create(gh_ctx* ctx, int L, const int64_t* row_off, const int32_t* row_state,
                                  const int64_t* arc_off, const int32_t* arc_to, const int32_t* arc_from,
                                  const double* arc_cost, const int64_t* start_off, const int32_t* start_rows,
                                  const int64_t* end_off, const int32_t* end_rows, gh_lattices** out) {
    GH_REQUIRE(ctx && out && row_off && row_state && arc_off && start_off && end_off && end_rows,
               "gh_lattices_create: NULL argument");
    GH_REQUIRE(L > 0, "gh_lattices_create: L=%d", L);
    *out = nullptr;
    GH_HIP(hipSetDevice(ctx->device));
    const int64_t Rtot = row_off[L], Atot = arc_off[L], Etot = end_off[L];
    GH_REQUIRE(row_off[0] == 0 && arc_off[0] == 0 && start_off[0] == 0 && end_off[0] == 0,
               "gh_lattices_create: offsets must start at 0");
    std::vector<int32_t> h_state(row_state, row_state + Rtot), h_ptr, h_order(Rtot), h_lev, h_narrow, h_end;
    std::vector<uint8_t> h_start(Rtot, 0);
    std::vector<uint32_t> h_prow(Atot), h_srow(Atot);
    std::vector<double> h_pcost(Atot), h_scost(Atot);
    std::vector<int32_t> h_sptr;
    gh_lattices* lt = new gh_lattices();
    lt->ctx = ctx; lt->L = L; lt->max_R = 0; lt->max_nlev = 0; lt->has_nan_arc = false; lt->has_self_arc = false;
    for (int64_t k = 0; k < Atot; ++k) lt->has_nan_arc = lt->has_nan_arc || std::isnan(arc_cost[k]);
    lt->d_row_state = nullptr; lt->d_row_start = nullptr; lt->d_pred_ptr = nullptr; lt->d_pred_row = nullptr;
    lt->d_pred_cost = nullptr; lt->d_order = nullptr; lt->d_succ_ptr = nullptr; lt->d_succ_row = nullptr;
    lt->d_succ_cost = nullptr; lt->d_level_ptr = nullptr; lt->d_end_rows = nullptr; lt->d_level_narrow = nullptr;
    lt->chain_ok = false; lt->chain_skip = false; lt->chain_groups = 0;
    lt->d_ch_cost0 = lt->d_ch_cost1 = lt->d_ch_cost2 = nullptr; lt->d_ch_info = nullptr;
    lt->d_ch_end_slot = nullptr; lt->d_ch_group_row0 = nullptr;
    lt->d_desc = nullptr;
    h_end.assign(end_rows, end_rows + Etot);
    for (int l = 0; l < L; ++l) {
        const int64_t r0 = row_off[l], a0 = arc_off[l];
        const int R = (int)(row_off[l + 1] - r0), A = (int)(arc_off[l + 1] - a0);
        const int ns = (int)(start_off[l + 1] - start_off[l]), ne = (int)(end_off[l + 1] - end_off[l]);
#define GH_LFAIL(...) do { gh_set_error(__VA_ARGS__); delete lt; return GH_ERR_INVALID; } while (0)
        if (R <= 0 || R > 0x7FFE) GH_LFAIL("gh_lattices_create: graph %d has %d rows (1..32766 supported)", l, R);
        if (ne <= 0) GH_LFAIL("gh_lattices_create: graph %d has no end row", l);
        for (int k = 0; k < ns; ++k) {
            const int r = start_rows[start_off[l] + k];
            if (r < 0 || r >= R) GH_LFAIL("gh_lattices_create: start row %d out of range", r);
            h_start[r0 + r] |= 1;
        }
        for (int k = 0; k < ne; ++k) {
            const int r = end_rows[end_off[l] + k];
            if (r < 0 || r >= R) GH_LFAIL("gh_lattices_create: end row %d out of range", r);
            h_start[r0 + r] |= 2;  // bit1 = end row (forward-backward)
        }
        // CSR by destination, ascending origin (tie-break contract, decode.py:105-118)
        std::vector<int> idx(A);
        std::iota(idx.begin(), idx.end(), 0);
        for (int k = 0; k < A; ++k) {
            const int to = arc_to[a0 + k], fr = arc_from[a0 + k];
            if (to < 0 || to >= R || fr < 0 || fr >= R) GH_LFAIL("gh_lattices_create: arc %d out of range", k);
            if (std::isinf(arc_cost[a0 + k]))  // NaN is kept: the reference treats it as an arc (isinf(nan) is False)
                GH_LFAIL("gh_lattices_create: arc %d has an infinite cost (omit absent arcs, decode.py:106)", k);
        }
        std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) {
            const int tx = arc_to[a0 + x], ty = arc_to[a0 + y];
            if (tx != ty) return tx < ty;
            return arc_from[a0 + x] < arc_from[a0 + y];
        });
        const size_t ptr_base = h_ptr.size();
        h_ptr.resize(ptr_base + R + 1, 0);
        int32_t* ptr = h_ptr.data() + ptr_base;
        std::vector<int> level(R, 0);
        for (int k = 0; k < A; ++k) ptr[arc_to[a0 + idx[k]] + 1]++;
        for (int r = 0; r < R; ++r) ptr[r + 1] += ptr[r];
        for (int k = 0; k < A; ++k) {
            const int to = arc_to[a0 + idx[k]], fr = arc_from[a0 + idx[k]];
            uint32_t w = (uint32_t)fr;
            const bool same = h_state[r0 + to] < 0 || h_state[r0 + fr] < 0;  // decode.py:109
            if (same) {
                if (fr == to) lt->has_self_arc = true;
                w |= GH_ARC_SAME;
                if (fr >= to) w |= GH_ARC_DEAD;  // not yet computed in this column: still +inf
            }
            h_prow[a0 + k] = w;
            h_pcost[a0 + k] = arc_cost[a0 + idx[k]];
        }
        // transposed CSR (by origin) for the backward pass; h_sptr shares ptr_base with h_ptr
        {
            h_sptr.resize(ptr_base + R + 1, 0);
            int32_t* sp = h_sptr.data() + ptr_base;
            for (int r = 0; r < R; ++r)
                for (int p = ptr[r]; p < ptr[r + 1]; ++p) sp[(h_prow[a0 + p] & GH_ARC_ROW) + 1]++;
            for (int r = 0; r < R; ++r) sp[r + 1] += sp[r];
            std::vector<int> fill(sp, sp + R);
            for (int r = 0; r < R; ++r)
                for (int p = ptr[r]; p < ptr[r + 1]; ++p) {
                    const uint32_t w = h_prow[a0 + p];
                    const int o = (int)(w & GH_ARC_ROW), k = fill[o]++;
                    h_srow[a0 + k] = (uint32_t)r | (w & ~GH_ARC_ROW);
                    h_scost[a0 + k] = h_pcost[a0 + p];
                }
        }
        // levels: rows ascending, so every live same-column origin (< row) is already levelled
        int nlev = 1;
        for (int r = 0; r < R; ++r) {
            int lv = 0;
            for (int p = ptr[r]; p < ptr[r + 1]; ++p) {
                const uint32_t w = h_prow[a0 + p];
                if ((w & GH_ARC_SAME) && !(w & GH_ARC_DEAD)) lv = std::max(lv, level[w & GH_ARC_ROW] + 1);
            }
            level[r] = lv;
            nlev = std::max(nlev, lv + 1);
        }
        const size_t lev_base = h_lev.size();
        h_lev.resize(lev_base + nlev + 1, 0);
        int32_t* lp = h_lev.data() + lev_base;
        for (int r = 0; r < R; ++r) lp[level[r] + 1]++;
        int max_level_rows = 0;
        for (int k = 0; k < nlev; ++k) { max_level_rows = std::max(max_level_rows, lp[k + 1]); lp[k + 1] += lp[k]; }
        // inside a level: rows with <= 2 arcs first, then the wide rows (the lean kernel gives those
        // 16 lanes each); ascending row index inside both groups
        h_narrow.resize(lev_base + nlev + 1, 0);
        int lean_lanes = 0;
        {
            std::vector<int> fill(lp, lp + nlev), nwide(nlev, 0);
            for (int pass = 0; pass < 2; ++pass)
                for (int r = 0; r < R; ++r) {
                    const bool wide = ptr[r + 1] - ptr[r] > 2;
                    if (wide != (pass == 1)) continue;
                    h_order[r0 + fill[level[r]]++] = r;
                    if (wide) nwide[level[r]]++; else h_narrow[lev_base + level[r]]++;
                }
            for (int k = 0; k < nlev; ++k) lean_lanes = std::max(lean_lanes, h_narrow[lev_base + k] + 16 * nwide[k]);
        }
        int max_state = -1;
        for (int r = 0; r < R; ++r) max_state = std::max(max_state, h_state[r0 + r]);
        gh_lattice_host lh;
        lh.R = R; lh.A = A; lh.nlev = nlev; lh.n_start = ns; lh.n_end = ne;
        lh.row_base = r0; lh.arc_base = a0; lh.end_base = end_off[l]; lh.max_state = max_state;
        lt->lat.push_back(lh);
        gh_lattices::desc d;
        d.R = R; d.nlev = nlev; d.n_end = ne; d.pad = max_level_rows; d.lean_lanes = lean_lanes; d.pad2 = 0;
        d.row_base = r0; d.ptr_base = (int64_t)ptr_base; d.arc_base = a0; d.lev_base = (int64_t)lev_base;
        d.end_base = end_off[l];
        lt->h_desc.push_back(d);
        lt->max_R = std::max(lt->max_R, R);
        lt->max_nlev = std::max(lt->max_nlev, nlev);
#undef GH_LFAIL
    }
    // ---- chain form: one graph, one level, arcs only from r, r-1, r-2, distinct end rows ----
    std::vector<double> ch0, ch1, ch2;
    std::vector<uint8_t> chinfo;
    std::vector<int32_t> chslot, chgroups;
    if (L == 1 && lt->max_nlev == 1 && !lt->has_nan_arc) {
        const int R = lt->lat[0].R;
        const int32_t* ptr = h_ptr.data();
        ch0.assign(R, INFINITY); ch1.assign(R, INFINITY); ch2.assign(R, INFINITY);
        chinfo.assign(R, 3); chslot.assign(R, -1);
        bool ok = true, skip = false;
        for (int r = 0; r < R && ok; ++r) {
            int first = 3;
            for (int p = ptr[r]; p < ptr[r + 1]; ++p) {
                const int o = (int)(h_prow[p] & GH_ARC_ROW), dlt = r - o;
                if (dlt < 0 || dlt > 2 || (h_prow[p] & (GH_ARC_SAME | GH_ARC_DEAD))) { ok = false; break; }
                double& slot = dlt == 0 ? ch0[r] : (dlt == 1 ? ch1[r] : ch2[r]);
                if (!std::isinf(slot)) { ok = false; break; }  // duplicate arc
                slot = h_pcost[p];
                if (dlt == 2) skip = true;
                if (first == 3 || dlt > first) first = dlt;  // lowest origin == largest delta comes first
            }
            chinfo[r] = (uint8_t)(first | (h_start[r] & 1 ? 4 : 0));
        }
        for (int k = 0; ok && k < (int)h_end.size(); ++k) {
            if (chslot[h_end[k]] >= 0) ok = false;  // duplicated end row: generic / lean kernels handle it
            else chslot[h_end[k]] = k;
        }
        if (ok) {  // 64-lane groups made of whole chains (a chain starts where no arc arrives from r-1 / r-2)
            chgroups.push_back(0);
            int gs = 0, cs = 0;
            for (int r = 1; r <= R && ok; ++r) {
                const bool chain_start = r == R || (std::isinf(ch1[r]) && std::isinf(ch2[r]) &&
                                                    (r + 1 >= R || std::isinf(ch2[r + 1])));
                if (!chain_start) continue;
                if (r - cs > 64) { ok = false; break; }     // one chain longer than a wave
                if (r - gs > 64) { chgroups.push_back(cs); gs = cs; }
                cs = r;
            }
            chgroups.push_back(R);
        }
        if (ok) { lt->chain_ok = true; lt->chain_skip = skip; lt->chain_groups = (int)chgroups.size() - 1; }
    }
    int rc = GH_OK;
    if (lt->chain_ok &&
        ((rc = upload(&lt->d_ch_cost0, ch0)) || (rc = upload(&lt->d_ch_cost1, ch1)) || (rc = upload(&lt->d_ch_cost2, ch2)) ||
         (rc = upload(&lt->d_ch_info, chinfo)) || (rc = upload(&lt->d_ch_end_slot, chslot)) ||
         (rc = upload(&lt->d_ch_group_row0, chgroups)))) {
        gh_lattices_destroy(lt);
        return rc;
    }
    if ((rc = upload(&lt->d_row_state, h_state)) || (rc = upload(&lt->d_row_start, h_start)) ||
        (rc = upload(&lt->d_pred_ptr, h_ptr)) || (rc = upload(&lt->d_pred_row, h_prow)) ||
        (rc = upload(&lt->d_pred_cost, h_pcost)) || (rc = upload(&lt->d_order, h_order)) ||
        (rc = upload(&lt->d_succ_ptr, h_sptr)) || (rc = upload(&lt->d_succ_row, h_srow)) ||
        (rc = upload(&lt->d_succ_cost, h_scost)) ||
        (rc = upload(&lt->d_level_ptr, h_lev)) || (rc = upload(&lt->d_level_narrow, h_narrow)) ||
        (rc = upload(&lt->d_end_rows, h_end)) ||
        (rc = upload(&lt->d_desc, lt->h_desc))) {
        gh_lattices_destroy(lt);
        return rc;
    }
    *out = lt;
    return GH_OK;
}

extern "C" void gh_lattices_destroy(gh_lattices* l) {
    if (!l) return;
    hipSetDevice(l->ctx->device);
    hipFree(l->d_row_state); hipFree(l->d_row_start); hipFree(l->d_pred_ptr); hipFree(l->d_pred_row);
    hipFree(l->d_pred_cost); hipFree(l->d_order); hipFree(l->d_level_ptr); hipFree(l->d_end_rows);
    hipFree(l->d_level_narrow);
    hipFree(l->d_ch_cost0); hipFree(l->d_ch_cost1); hipFree(l->d_ch_cost2); hipFree(l->d_ch_info);
    hipFree(l->d_ch_end_slot); hipFree(l->d_ch_group_row0);
    hipFree(l->d_succ_ptr); hipFree(l->d_succ_row); hipFree(l->d_succ_cost);
    hipFree(l->d_desc);
    delete l;
}

extern "C" int64_t gh_viterbi_path_cap(const gh_lattices* lat, int l, int64_t T) {
    if (!lat || l < 0 || l >= lat->L) return -1;
    return T * lat->lat[l].nlev;  // at most one cell per (column, level)
}

// ------------------------------------------------------------------ viterbi
extern "C" int gh_viterbi(gh_ctx* ctx, const gh_lattices* lat, const gh_batch* b, const int32_t* utt_lattice,
                          double* out_end_cost, int32_t* out_best_end, int32_t* out_path,
                          const int64_t* path_off, int32_t* out_path_len, double* out_costs,
                          const int64_t* costs_off) {
    GH_REQUIRE(ctx && lat && b, "gh_viterbi: NULL argument");
    GH_REQUIRE(b->nll || b->N == 0, "gh_viterbi: gh_loglik has not been run on this batch");
    GH_REQUIRE(!out_path || (path_off && out_path_len), "gh_viterbi: out_path needs path_off and out_path_len");
    GH_REQUIRE(!out_costs || costs_off, "gh_viterbi: out_costs needs costs_off");
    GH_HIP(hipSetDevice(ctx->device));
    const int64_t U = b->U;
    if (U == 0) return GH_OK;
    const int S = b->nll_S;
    const bool want_path = out_path != nullptr;
    for (int l = 0; l < lat->L; ++l)
        GH_REQUIRE(lat->lat[l].max_state < S, "gh_viterbi: graph %d uses state %d but the model has %d", l,
                   lat->lat[l].max_state, S);
    // per-utterance bookkeeping (host); with one graph for all utterances everything is implicit
    const bool uniform = utt_lattice == nullptr;
    const std::vector<int64_t>& perm = b->perm;
    std::vector<int64_t> end_off;
    int64_t n_end_total = 0;
    if (uniform) {
        n_end_total = U * lat->lat[0].n_end;
    } else {
        end_off.assign(U + 1, 0);
        for (int64_t u = 0; u < U; ++u) {
            const int l = utt_lattice[u];
            GH_REQUIRE(l >= 0 && l < lat->L, "gh_viterbi: utt_lattice[%lld]=%d out of range", (long long)u, l);
            end_off[u + 1] = end_off[u] + lat->lat[l].n_end;
        }
        n_end_total = end_off[U];
    }
    if (want_path)
        for (int64_t u = 0; u < U; ++u) {
            const int l = uniform ? 0 : utt_lattice[u];
            const int64_t T = b->offsets[u + 1] - b->offsets[u];
            GH_REQUIRE(path_off[u + 1] - path_off[u] >= (T > 1 ? T * lat->lat[l].nlev : 0),
                       "gh_viterbi: path capacity of utterance %lld too small", (long long)u);
        }
    // back-pointer scratch is chunked (<= 4 GiB per launch)
    const size_t BP_BUDGET = (size_t)4 << 30;
    std::vector<int64_t> bp_off(U, 0);
    std::vector<int64_t> chunk_begin{0};
    size_t bp_max = 0;
    // chain kernel: one left-to-right graph for the whole batch, no single-frame utterance (T == 1 has
    // the reference's wrap-around semantics, which only the lean / generic kernels implement)
    bool use_chain = lat->chain_ok && uniform;
    {
        static const bool no_chain = [] { const char* e = getenv("GMMHMM_VITERBI"); return e && (!strcmp(e, "generic") || !strcmp(e, "lean")); }();
        if (no_chain) use_chain = false;
        for (int64_t u = 0; use_chain && u < U; ++u)
            if (b->offsets[u + 1] - b->offsets[u] == 1) use_chain = false;
    }
    const bool want_bp = want_path || (use_chain && out_costs);
    if (want_bp) {
        size_t acc = 0;
        for (int64_t k = 0; k < U; ++k) {
            const int64_t u = perm[k];
            const int l = utt_lattice ? utt_lattice[u] : 0;
            const size_t need = (size_t)(b->offsets[u + 1] - b->offsets[u]) * lat->lat[l].R;
            if (acc && (acc + need) * 2 > BP_BUDGET) {
                chunk_begin.push_back(k);
                bp_max = std::max(bp_max, acc);
                acc = 0;
            }
            bp_off[k] = (int64_t)acc;
            acc += need;
        }
        bp_max = std::max(bp_max, acc);
    }
    chunk_begin.push_back(U);
    const int64_t n_path = want_path ? path_off[U] : 0;
    const int64_t n_costs = out_costs ? costs_off[U] : 0;

    gh_vit_args a;
    memset(&a, 0, sizeof a);
    int64_t *d_bpoff = nullptr, *d_endoff = nullptr, *d_pathoff = nullptr, *d_costsoff = nullptr;
    int32_t *d_uttlat = nullptr, *d_bestend, *d_path = nullptr, *d_pathlen = nullptr;
    double *d_endcost, *d_costs = nullptr;
    uint16_t* d_bp = nullptr;
    Carver cv;
    int* d_flag2;  // [flag | best_end | end_cost] are carved back to back: ONE D2H copy into pinned memory
    cv.add(&d_flag2, 64); cv.add(&d_bestend, U); cv.add(&d_endcost, n_end_total);
    const size_t small_bytes = cv.total;
    if (want_bp) cv.add(&d_bpoff, U);
    if (!uniform) cv.add(&d_endoff, U + 1);
    if (utt_lattice) cv.add(&d_uttlat, U);
    if (want_path) { cv.add(&d_pathoff, U + 1); cv.add(&d_path, 2 * n_path); cv.add(&d_pathlen, U); }
    if (want_bp) cv.add(&d_bp, bp_max);
    if (out_costs) { cv.add(&d_costsoff, U + 1); cv.add(&d_costs, n_costs); }
    int rc = cv.commit(ctx);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    GH_HIP(hipMemsetAsync(d_flag2, 0, sizeof(int), st));
    if (want_bp) GH_HIP(hipMemcpyAsync(d_bpoff, bp_off.data(), U * 8, hipMemcpyHostToDevice, st));
    if (!uniform) GH_HIP(hipMemcpyAsync(d_endoff, end_off.data(), (U + 1) * 8, hipMemcpyHostToDevice, st));
    if (utt_lattice) GH_HIP(hipMemcpyAsync(d_uttlat, utt_lattice, U * 4, hipMemcpyHostToDevice, st));
    if (want_path) GH_HIP(hipMemcpyAsync(d_pathoff, path_off, (U + 1) * 8, hipMemcpyHostToDevice, st));
    if (out_costs) GH_HIP(hipMemcpyAsync(d_costsoff, costs_off, (U + 1) * 8, hipMemcpyHostToDevice, st));

    a.descs = lat->d_desc; a.row_state = lat->d_row_state; a.row_start = lat->d_row_start;
    a.pred_ptr = lat->d_pred_ptr; a.pred_row = lat->d_pred_row; a.pred_cost = lat->d_pred_cost;
    a.order = lat->d_order; a.level_ptr = lat->d_level_ptr; a.end_rows = lat->d_end_rows;
    a.level_narrow = lat->d_level_narrow;
    a.nll = b->nll; a.S = S; a.r_pad = (lat->max_R + 1) & ~1;
    a.utt_off = b->d_offsets; a.utt_lat = d_uttlat; a.perm = b->d_perm;
    a.bp = d_bp; a.bp_off = d_bpoff; a.end_cost = d_endcost; a.end_off = d_endoff; a.best_end = d_bestend;
    a.path = d_path; a.path_off = d_pathoff; a.path_len = d_pathlen; a.costs = d_costs; a.costs_off = d_costsoff;
    a.flag = d_flag2;

    if (use_chain) {
        gh_chain_args c;
        memset(&c, 0, sizeof c);
        c.cost0 = lat->d_ch_cost0; c.cost1 = lat->d_ch_cost1; c.cost2 = lat->d_ch_cost2; c.row_info = lat->d_ch_info;
        c.row_state = lat->d_row_state; c.end_slot = lat->d_ch_end_slot; c.end_rows = lat->d_end_rows;
        c.group_row0 = lat->d_ch_group_row0; c.n_groups = lat->chain_groups; c.R = lat->lat[0].R; c.S = S;
        c.n_end = lat->lat[0].n_end; c.nll = b->nll; c.utt_off = b->d_offsets; c.perm = b->d_perm;
        c.bp = reinterpret_cast<uint8_t*>(d_bp); c.bp_off = d_bpoff; c.end_cost = d_endcost; c.best_end = d_bestend;
        c.path = d_path; c.path_off = d_pathoff; c.path_len = d_pathlen; c.costs = d_costs; c.costs_off = d_costsoff;
        c.flag = d_flag2;
        for (size_t k = 0; k + 1 < chunk_begin.size(); ++k) {
            const int64_t u0 = chunk_begin[k], nu = chunk_begin[k + 1] - u0;
            rc = gh_launch_viterbi_chain(ctx, c, u0, nu, b->dtype == GH_F64, want_bp, out_costs != nullptr, lat->chain_skip);
            if (!rc) rc = gh_launch_chain_backtrace(ctx, c, u0, nu);  // end selection (+ path when requested)
            if (rc) return rc;
        }
    }
    int max_level_rows = 1;
    for (auto& d : lat->h_desc) max_level_rows = std::max(max_level_rows, d.pad);
    int block = std::min(512, std::max(64, (max_level_rows + 63) & ~63));
    // lean kernel: <= 3 levels (the same number in every graph), one row per lane per level, arc lists
    // that fit LDS, no NaN arc cost, no same-column self arc (GMMHMM_VITERBI=generic forces the generic one)
    int lean_levels = 0, max_arcs = 0;
    for (auto& lh : lat->lat) max_arcs = std::max(max_arcs, lh.A);
    {
        static const bool no_lean = [] { const char* e = getenv("GMMHMM_VITERBI"); return e && !strcmp(e, "generic"); }();
        int lean_lanes = 1;
        for (auto& d : lat->h_desc) lean_lanes = std::max(lean_lanes, d.lean_lanes);
        const int lb = std::max(64, (lean_lanes + 63) & ~63);
        bool same_nlev = true;
        for (auto& d : lat->h_desc) same_nlev = same_nlev && d.nlev == lat->max_nlev;
        if (!no_lean && !(out_costs && !want_path) && !lat->has_nan_arc && !lat->has_self_arc && same_nlev && lat->max_nlev <= 3 && lb <= 1024 &&
            S <= 8 * lb && max_arcs <= 4096) {
            lean_levels = lat->max_nlev;
            block = lb;
        }
    }
    size_t lds;
    if (lean_levels) {
        a.em_chunk = std::max(1, std::min(8, 8 * block / std::max(S, 1)));
        if (const char* e = getenv("GMMHMM_EMCHUNK")) a.em_chunk = std::max(1, std::min(a.em_chunk, atoi(e)));  // tuning knob
        a.arc_cap = (max_arcs + 1) & ~1;
        lds = (size_t)2 * a.r_pad * 8 + (size_t)2 * a.em_chunk * (S + 1) * 8 + (size_t)a.arc_cap * 12 + 16;
    } else {
        a.em_chunk = 1;
        lds = ((size_t)2 * a.r_pad + S) * sizeof(double);
    }
    if (lds > 160 * 1024) {
        gh_set_error("gh_viterbi: %d rows + %d states need %zu B of LDS (> 160 KiB)", lat->max_R, S, lds);
        return GH_ERR_UNSUPPORTED;
    }
    for (size_t c = 0; !use_chain && c + 1 < chunk_begin.size(); ++c) {
        a.u_begin = chunk_begin[c];
        const int64_t nu = chunk_begin[c + 1] - chunk_begin[c];
        rc = lean_levels ? gh_launch_viterbi_lean(ctx, a, nu, block, lds, b->dtype == GH_F64, want_path, lean_levels)
                         : gh_launch_viterbi(ctx, a, nu, block, lds, b->dtype == GH_F64, want_path);
        if (rc) return rc;
    }
    char* pin;
    rc = gh_pinned(ctx, small_bytes, (void**)&pin);
    if (rc) return rc;
    GH_HIP(hipMemcpyAsync(pin, d_flag2, small_bytes, hipMemcpyDeviceToHost, st));
    if (want_path) {
        GH_HIP(hipMemcpyAsync(out_path, d_path, 2 * n_path * 4, hipMemcpyDeviceToHost, st));
        GH_HIP(hipMemcpyAsync(out_path_len, d_pathlen, U * 4, hipMemcpyDeviceToHost, st));
    }
    if (out_costs) GH_HIP(hipMemcpyAsync(out_costs, d_costs, n_costs * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    const int flag = *reinterpret_cast<int*>(pin);
    if (out_best_end) memcpy(out_best_end, pin + ((char*)d_bestend - (char*)d_flag2), U * 4);
    if (out_end_cost) memcpy(out_end_cost, pin + ((char*)d_endcost - (char*)d_flag2), n_end_total * 8);
    if (flag & 1) {
        gh_set_error("gh_viterbi: a DP cell chose itself as its origin (decode.py:120-121)");
        return GH_ERR_SELF_POINTER;
    }
    if (flag & 4) {
        gh_set_error("gh_viterbi: back-trace does not terminate (cycle of same-column arcs between unreachable cells)");
        return GH_ERR_INVALID;
    }
    if (flag & 2) {
        gh_set_error("gh_viterbi: back-trace reached a cell without predecessor");
        return GH_ERR_INVALID;
    }
    return GH_OK;
}

// ---------------------------------------------------------------------- dtw
extern "C" int gh_dtw(gh_ctx* ctx, const gh_batch* b, int n, const double* y, const double* var,
                      const double* trans, int beam, const double* dist_host, double* out_costs,
                      int32_t* out_path, int32_t* out_path_len) {
    GH_REQUIRE(ctx && b && trans, "gh_dtw: NULL argument");
    GH_REQUIRE(dist_host || y, "gh_dtw: need template rows or a distance matrix");
    GH_REQUIRE(n > 1 && n <= 64, "gh_dtw: n=%d (2..64 supported; decode.py:22 asserts n > 1)", n);
    GH_REQUIRE(b->dtype == GH_F64 || dist_host, "gh_dtw: built-in distances need an fp64 batch");
    GH_REQUIRE(!out_path || out_path_len, "gh_dtw: out_path needs out_path_len");
    GH_HIP(hipSetDevice(ctx->device));
    const int64_t U = b->U, N = b->N;
    const int D = b->D;
    if (U == 0) return GH_OK;
    for (int64_t u = 0; u < U; ++u)
        GH_REQUIRE(b->offsets[u + 1] - b->offsets[u] > 1, "gh_dtw: utterance %lld has fewer than 2 frames (decode.py:22)",
                   (long long)u);
    std::vector<double> logdet(n, 0.0);
    if (var && !dist_host)
        for (int i = 0; i < n; ++i) {
            double prod = 1.0;
            for (int d = 0; d < D; ++d) prod *= var[(size_t)i * D + d];
            logdet[i] = 0.5 * std::log(std::pow(2.0 * M_PI, D) * prod);  // hmm_state.py:58
        }
    std::vector<int64_t> moff(U + 1);  // n * frame offset: [n,T] blocks
    for (int64_t u = 0; u <= U; ++u) moff[u] = (int64_t)n * b->offsets[u];
    double *d_y = nullptr, *d_var = nullptr, *d_ld = nullptr, *d_tr, *d_E = nullptr, *d_costs = nullptr;
    int64_t* d_moff;
    uint8_t* d_bp;
    int32_t *d_path = nullptr, *d_plen = nullptr;
    Carver cv;
    cv.add(&d_tr, (size_t)n * n); cv.add(&d_moff, U + 1); cv.add(&d_bp, (size_t)n * N);
    if (!dist_host) { cv.add(&d_y, (size_t)n * D); if (var) { cv.add(&d_var, (size_t)n * D); cv.add(&d_ld, n); } }
    if (dist_host) cv.add(&d_E, (size_t)n * N);
    if (out_costs) cv.add(&d_costs, (size_t)n * N);
    if (out_path) { cv.add(&d_path, 2 * (size_t)N); cv.add(&d_plen, U); }
    int rc = cv.commit(ctx);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    GH_HIP(hipMemsetAsync(ctx->d_flag, 0, sizeof(int), st));
    GH_HIP(hipMemcpyAsync(d_tr, trans, (size_t)n * n * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(d_moff, moff.data(), (U + 1) * 8, hipMemcpyHostToDevice, st));
    if (d_y) GH_HIP(hipMemcpyAsync(d_y, y, (size_t)n * D * 8, hipMemcpyHostToDevice, st));
    if (d_var) {
        GH_HIP(hipMemcpyAsync(d_var, var, (size_t)n * D * 8, hipMemcpyHostToDevice, st));
        GH_HIP(hipMemcpyAsync(d_ld, logdet.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    }
    if (d_E) GH_HIP(hipMemcpyAsync(d_E, dist_host, (size_t)n * N * 8, hipMemcpyHostToDevice, st));
    gh_dtw_args a;
    memset(&a, 0, sizeof a);
    a.x = dist_host ? nullptr : (const double*)b->feats;
    a.E = d_E; a.e_off = d_moff; a.utt_off = b->d_offsets; a.n = n; a.D = D; a.beam = beam;
    a.y = d_y; a.var = d_var; a.logdet = d_ld; a.trans = d_tr; a.bp = d_bp; a.bp_off = d_moff;
    a.costs = d_costs; a.costs_off = d_moff; a.path = d_path; a.path_off = b->d_offsets; a.path_len = d_plen;
    a.flag = ctx->d_flag;
    rc = gh_launch_dtw(ctx, a, U);
    if (rc) return rc;
    int flag = 0;
    GH_HIP(hipMemcpyAsync(&flag, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    if (out_costs) GH_HIP(hipMemcpyAsync(out_costs, d_costs, (size_t)n * N * 8, hipMemcpyDeviceToHost, st));
    if (out_path) {
        GH_HIP(hipMemcpyAsync(out_path, d_path, 2 * (size_t)N * 4, hipMemcpyDeviceToHost, st));
        GH_HIP(hipMemcpyAsync(out_path_len, d_plen, U * 4, hipMemcpyDeviceToHost, st));
    }
    GH_HIP(hipStreamSynchronize(st));
    if (flag & 8) {
        gh_set_error("gh_dtw: the beam pruned every origin of a column (np.argmin of an empty list)");
        return GH_ERR_INVALID;
    }
    if (flag & 2) {
        gh_set_error("gh_dtw: back-trace left the matrix: cell (0,0) is not reachable from the end cell");
        return GH_ERR_INVALID;
    }
    return GH_OK;
}

// --------------------------------------------------------- forward-backward
extern "C" int gh_forward_backward(gh_ctx* ctx, const gh_lattices* lat, gh_batch* b, const int32_t* utt_lattice,
                                   int want_occ, double* out_logp, double* out_alpha, double* out_beta,
                                   double* out_gamma, const int64_t* mat_off, double* out_occ) {
    GH_REQUIRE(ctx && lat && b, "gh_forward_backward: NULL argument");
    GH_REQUIRE(b->nll || b->N == 0, "gh_forward_backward: gh_loglik has not been run on this batch");
    GH_REQUIRE(!(out_alpha || out_beta || out_gamma) || mat_off, "gh_forward_backward: matrices need mat_off");
    GH_REQUIRE(!out_occ || want_occ, "gh_forward_backward: out_occ needs want_occ");
    GH_HIP(hipSetDevice(ctx->device));
    const int64_t U = b->U;
    if (U == 0) return GH_OK;
    const int S = b->nll_S;
    for (int l = 0; l < lat->L; ++l)
        GH_REQUIRE(lat->lat[l].max_state < S, "gh_forward_backward: graph %d uses state %d but the model has %d", l,
                   lat->lat[l].max_state, S);
    if (utt_lattice)
        for (int64_t u = 0; u < U; ++u)
            GH_REQUIRE(utt_lattice[u] >= 0 && utt_lattice[u] < lat->L, "gh_forward_backward: utt_lattice[%lld] out of range",
                       (long long)u);
    if (want_occ && !b->occ && b->N > 0) GH_HIP(hipMalloc((void**)&b->occ, (size_t)b->N * S * 8));
    // alpha scratch, chunked (<= 4 GiB per launch), launch order = longest first
    const size_t BUDGET = (size_t)4 << 30;
    std::vector<int64_t> soff(U, 0), chunk_begin{0};
    size_t acc = 0, smax = 0;
    for (int64_t k = 0; k < U; ++k) {
        const int64_t u = b->perm[k];
        const int l = utt_lattice ? utt_lattice[u] : 0;
        const size_t need = (size_t)(b->offsets[u + 1] - b->offsets[u]) * lat->lat[l].R;
        if (acc && (acc + need) * 8 > BUDGET) { chunk_begin.push_back(k); smax = std::max(smax, acc); acc = 0; }
        soff[k] = (int64_t)acc;
        acc += need;
    }
    smax = std::max(smax, acc);
    chunk_begin.push_back(U);
    const bool mats = out_alpha || out_beta || out_gamma;
    const int64_t n_mat = mats ? mat_off[U] : 0;
    int64_t *d_soff, *d_matoff = nullptr;
    int32_t* d_uttlat = nullptr;
    double *d_scratch, *d_logp, *d_alpha = nullptr, *d_beta = nullptr, *d_gamma = nullptr;
    Carver cv;
    cv.add(&d_soff, U); cv.add(&d_logp, U); cv.add(&d_scratch, smax);
    if (utt_lattice) cv.add(&d_uttlat, U);
    if (mats) cv.add(&d_matoff, U + 1);
    if (out_alpha) cv.add(&d_alpha, n_mat);
    if (out_beta) cv.add(&d_beta, n_mat);
    if (out_gamma) cv.add(&d_gamma, n_mat);
    int rc = cv.commit(ctx);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    GH_HIP(hipMemcpyAsync(d_soff, soff.data(), U * 8, hipMemcpyHostToDevice, st));
    if (utt_lattice) GH_HIP(hipMemcpyAsync(d_uttlat, utt_lattice, U * 4, hipMemcpyHostToDevice, st));
    if (mats) GH_HIP(hipMemcpyAsync(d_matoff, mat_off, (U + 1) * 8, hipMemcpyHostToDevice, st));
    gh_fb_args a;
    memset(&a, 0, sizeof a);
    a.descs = lat->d_desc; a.row_state = lat->d_row_state; a.row_flag = lat->d_row_start;
    a.pred_ptr = lat->d_pred_ptr; a.pred_row = lat->d_pred_row; a.pred_cost = lat->d_pred_cost;
    a.succ_ptr = lat->d_succ_ptr; a.succ_row = lat->d_succ_row; a.succ_cost = lat->d_succ_cost;
    a.order = lat->d_order; a.level_ptr = lat->d_level_ptr; a.end_rows = lat->d_end_rows;
    a.nll = b->nll; a.S = S; a.r_pad = (lat->max_R + 1) & ~1;
    a.utt_off = b->d_offsets; a.utt_lat = d_uttlat; a.perm = b->d_perm;
    a.alpha_scratch = d_scratch; a.scratch_off = d_soff; a.logp = d_logp;
    a.out_alpha = d_alpha; a.out_beta = d_beta; a.out_gamma = d_gamma; a.mat_off = d_matoff;
    a.occ = want_occ ? b->occ : nullptr;
    int max_level_rows = 1;
    for (auto& d : lat->h_desc) max_level_rows = std::max(max_level_rows, d.pad);
    const int block = std::min(512, std::max(64, (max_level_rows + 63) & ~63));
    const size_t lds = ((size_t)2 * a.r_pad + 3 * (size_t)S) * sizeof(double);
    if (lds > 150 * 1024) {
        gh_set_error("gh_forward_backward: %d rows + %d states need %zu B of LDS", lat->max_R, S, lds);
        return GH_ERR_UNSUPPORTED;
    }
    for (size_t c = 0; c + 1 < chunk_begin.size(); ++c) {
        a.u_begin = chunk_begin[c];
        rc = gh_launch_fb(ctx, a, chunk_begin[c + 1] - chunk_begin[c], block, lds, b->dtype == GH_F64);
        if (rc) return rc;
    }
    if (out_logp) GH_HIP(hipMemcpyAsync(out_logp, d_logp, U * 8, hipMemcpyDeviceToHost, st));
    if (out_alpha) GH_HIP(hipMemcpyAsync(out_alpha, d_alpha, n_mat * 8, hipMemcpyDeviceToHost, st));
    if (out_beta) GH_HIP(hipMemcpyAsync(out_beta, d_beta, n_mat * 8, hipMemcpyDeviceToHost, st));
    if (out_gamma) GH_HIP(hipMemcpyAsync(out_gamma, d_gamma, n_mat * 8, hipMemcpyDeviceToHost, st));
    if (out_occ) GH_HIP(hipMemcpyAsync(out_occ, b->occ, (size_t)b->N * S * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}
