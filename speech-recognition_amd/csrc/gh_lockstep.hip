// Lock-step training of ALL states at once: the reference refits one state after the other --
// split k-means (kmeans.py:167-193) and mixture EM (hmm_state.py:122-159) on the frames Viterbi alignment gave the
// state, hmm.py:97-124 / continuous_speech.py:114-142 -- one host round trip per state and iteration.  Here the
// frames of every state sit back to back in one resident batch ("gathered" order: state s owns frames
// [seg_off[s], seg_off[s+1])) and one launch advances every state that has not converged yet:
//   * gh_kmeans_assign_multi  -- the N x k distance / arg-min sweep of kmeans.py:180-186 for every active state,
//     optionally with the per-(state, cluster) sums and counts of the centroid update (cluster_centroids,
//     kmeans.py:158-164; kmeans_rowsum_kernel, in numpy's summation order) and the number of assignments that changed
//     (the sharded trainer all-reduces those);
//   * gh_em_accumulate_multi  -- the E-step statistics of hmm_state.py:127-143 for every active state
//     (same centred layout as gh_em_accumulate).
// One WAVE per tile of <= 64 frames of one state (tiles never straddle states): the frames are staged through LDS
// with coalesced loads and end up one frame per lane in registers; the state's parameters are read from LDS by
// broadcast.  Per-tile partial results are summed per state in tile order by a second kernel (deterministic).
#include "gh_internal.h"
#include "gh_host.h"

namespace {

constexpr int LS_MAXD = 64;   // feature dimensions a lane keeps in registers

struct ls_tile { int64_t first; int32_t state, count; };

// stage a tile of <= 64 frames: coalesced global -> LDS [64][D + 1], then lane = frame -> registers
template <int DR>
__device__ __forceinline__ void stage_tile(const double* __restrict__ X, int64_t first, int count, int D, double* tile,
                                           double (&x)[DR]) {
    const int lane = threadIdx.x;
    const double* src = X + first * D;
    const int nelem = count * D;
    for (int i = lane; i < nelem; i += 64) {
        const int f = i / D, d = i - f * D;
        tile[f * (D + 1) + d] = src[i];
    }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < DR; ++d) x[d] = (d < D && lane < count) ? tile[lane * (D + 1) + d] : 0.0;
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ k-means
// Cluster sums in FRAME ORDER: numpy reduces a C-contiguous [n, D] array over axis 0 row after row (pairwise summation
// only applies along the contiguous axis), so cluster_centroids' np.mean(data[clusters == c], axis=0)
// (kmeans.py:158-164) is the sequential sum of the cluster's frames divided by their number; sums / count is then
// BITWISE the reference's centroid (for D >= 2).  A sequential sum is a serial chain per (state, cluster, dimension),
// so everything that is NOT the chain is taken out of it:
//   * the frames of every cluster are first listed in order (stable compaction: per-tile cluster counts from the
//     assignment kernel -> exclusive scan over the state's tiles -> every tile scatters its frame numbers), so a
//     chain walks only its own cluster's frames -- n / k steps instead of n compare-and-select steps;
//   * one wave per (state, cluster), lane = dimension: the frame numbers of 64 list entries are ONE coalesced load
//     (lane j holds entry j's) handed out by v_readlane; a ring of RS_B registers keeps RS_B row loads in flight and
//     refills a slot right after it is summed.
// (History: a thread per (state, cluster, dimension) reading global memory in the chain, 8 loads in flight: 150
// cycles per frame, 1.9 ms per k-means iteration for 50 states of 28 000 frames; staging through LDS made the state's
// waves queue for their CU's LDS; a 64-deep register ring with select: 0.89 ms; compacted lists: see profiles.)
constexpr int RS_B = 64;

// exclusive scan of the per-tile cluster counts over the tiles of every state (in tile = frame order); also the
// number of frames of every (state, cluster) and where its list starts inside the state's segment
__global__ void kmeans_scan_kernel(const int32_t* __restrict__ tile_ptr /*[S+1]*/, int k, int32_t* __restrict__ tilecnt /*[tiles][k] -> offsets*/,
                                   int32_t* __restrict__ counts /*[S,k]*/, int32_t* __restrict__ cbase /*[S,k]*/) {
    const int s = blockIdx.x, c = threadIdx.x;
    __shared__ int tot[64];
    int run = 0;
    if (c < k)
        for (int t = tile_ptr[s]; t < tile_ptr[s + 1]; ++t) {
            const int v = tilecnt[(int64_t)t * k + c];
            tilecnt[(int64_t)t * k + c] = run;
            run += v;
        }
    tot[c] = c < k ? run : 0;
    __syncthreads();
    if (c < k) {
        int base = 0;
        for (int j = 0; j < c; ++j) base += tot[j];
        counts[s * k + c] = run;
        cbase[s * k + c] = base;
    }
}

// every tile writes the (segment-local) numbers of its frames into their clusters' lists, keeping frame order
__global__ __launch_bounds__(64) void kmeans_scatter_kernel(const ls_tile* __restrict__ tiles, int k, const int32_t* __restrict__ clusters,
                                                            const int64_t* __restrict__ seg_off, const int32_t* __restrict__ tileoff,
                                                            const int32_t* __restrict__ cbase, int32_t* __restrict__ lists /*[N]*/) {
    const ls_tile tl = tiles[blockIdx.x];
    const int lane = threadIdx.x;
    const bool act = lane < tl.count;
    const int id = act ? clusters[tl.first + lane] : -1;
    const int64_t seg0 = seg_off[tl.state];
    for (int c = 0; c < k; ++c) {
        const unsigned long long m = __ballot(act && id == c);
        if (act && id == c) {
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            lists[seg0 + cbase[tl.state * k + c] + tileoff[(int64_t)blockIdx.x * k + c] + rank] = (int32_t)(tl.first + lane - seg0);
        }
    }
}

// grid (S, k), block = D rounded up to waves: the chain of (state, cluster, dimension = thread)
__global__ __launch_bounds__(64) void kmeans_rowsum_kernel(const double* __restrict__ X, int D, int k,
                                                           const int64_t* __restrict__ seg_off, const uint8_t* __restrict__ active,
                                                           const int32_t* __restrict__ lists, const int32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ cbase, double* __restrict__ sums) {
    const int s = blockIdx.x, c = blockIdx.y;
    if (active && !active[s]) return;
    const int d = blockIdx.z * 64 + threadIdx.x, lane = threadIdx.x;
    const bool live = d < D;
    const int64_t f0 = seg_off[s];
    const int n = counts[s * k + c];
    if (d == 0) sums[((int64_t)s * k + c) * (D + 1) + D] = (double)n;
    if (n <= 0) return;
    const double* col = X + f0 * D + (live ? d : 0);
    const int32_t* li = lists + f0 + cbase[s * k + c];
    int id = li[min(lane, n - 1)];
    double x[RS_B];
#pragma unroll
    for (int j = 0; j < RS_B; ++j) x[j] = col[(int64_t)__builtin_amdgcn_readlane(id, j) * D];
    double acc = 0.0;
    int b = 0;
    for (; b + RS_B <= n; b += RS_B) {                        // full rounds: the chain is one add per frame
        id = li[min(b + RS_B + lane, n - 1)];                 // the next 64 entries (clamped: re-reads the last frame, never summed)
#pragma unroll
        for (int j = 0; j < RS_B; ++j) {
            acc += x[j];
            x[j] = col[(int64_t)__builtin_amdgcn_readlane(id, j) * D];
        }
    }
    const int left = n - b;                                   // the last, partial round
#pragma unroll
    for (int j = 0; j < RS_B; ++j) acc += (j < left) ? x[j] : 0.0;
    if (live) sums[((int64_t)s * k + c) * (D + 1) + d] = acc;
}

template <int DR>
__global__ __launch_bounds__(64) void kmeans_multi_kernel(const double* __restrict__ X, int D, int k, const ls_tile* __restrict__ tiles,
                                                          const double* __restrict__ cent /*[S,k,D]*/,
                                                          const double* __restrict__ var /*[S,D] or null*/,
                                                          const double* __restrict__ logdet /*[S]*/,
                                                          int32_t* __restrict__ clusters /*[N] in/out*/,
                                                          int32_t* __restrict__ changed /*[S] or null*/,
                                                          int32_t* __restrict__ counts /*[tiles][k] or null*/) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* sc = sm;                  // [k][D]
    double* sv = sc + k * D;          // [D]
    double* tile = sv + D;            // [64][D+1]
    const ls_tile tl = tiles[blockIdx.x];
    const int lane = threadIdx.x;
    const double* c0 = cent + (int64_t)tl.state * k * D;
    for (int i = lane; i < k * D; i += 64) sc[i] = c0[i];
    if (var) for (int i = lane; i < D; i += 64) sv[i] = var[(int64_t)tl.state * D + i];
    double x[DR];
    stage_tile<DR>(X, tl.first, tl.count, D, tile, x);
    double best = 0;
    int bi = 0;
    const double ld = var ? logdet[tl.state] : 0.0;
    for (int c = 0; c < k; ++c) {
        double q = 0, dist;
        if (var) {
#pragma unroll
            for (int d = 0; d < DR; ++d) if (d < D) { const double t = sc[c * D + d] - x[d]; q += t / sv[d] * t; }
            dist = ld + 0.5 * q;                                   // mahalanobis(centroid, x, cov[0]), kmeans.py:183
        } else {
#pragma unroll
            for (int d = 0; d < DR; ++d) if (d < D) { const double t = sc[c * D + d] - x[d]; q = fma(t, t, q); }
            dist = sqrt(q);
        }
        if (c == 0 || dist < best || (dist != dist && best == best)) { best = dist; bi = c; }   // np.argmin
    }
    const bool act = lane < tl.count;
    if (act) {
        int32_t* slot = clusters + tl.first + lane;
        if (changed && *slot != bi) atomicAdd(changed + tl.state, 1);
        *slot = bi;
    }
    if (counts)                                               // frames of this tile per cluster (scanned by kmeans_scan_kernel)
        for (int c = 0; c < k; ++c) {
            const int nc = __popcll(__ballot(act && bi == c));
            if (lane == 0) counts[(int64_t)blockIdx.x * k + c] = nc;
        }
}

// partial[tile][len] summed over the tiles of each state (contiguous, in order) -> out[state][len]
__global__ void tiles_reduce_kernel(const double* __restrict__ partial, const int32_t* __restrict__ tile_ptr /*[S+1]*/, int len,
                                    double* __restrict__ out) {
    const int s = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    double acc = 0;
    for (int t = tile_ptr[s]; t < tile_ptr[s + 1]; ++t) acc += partial[(int64_t)t * len + i];
    out[(int64_t)s * len + i] = acc;
}

// ------------------------------------------------------------------------------------------------ EM E-step
constexpr double LS_LN_UNDERFLOW = -745.1332191019412;   // exp(x) rounds to +0 in fp64 below this (see gh_train.hip)

template <int DR>
__global__ __launch_bounds__(64) void em_multi_kernel(const double* __restrict__ X, int D, int k, const ls_tile* __restrict__ tiles,
                                                      const double* __restrict__ mean /*[S,k,D]*/,
                                                      const double* __restrict__ ivar /*[S,k,D]*/,
                                                      const double* __restrict__ logc /*[S,k]*/,
                                                      double* __restrict__ partial /*[tiles][k*(1+2D) + 1]*/) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* pm = sm;                  // [k][D]
    double* pv = pm + k * D;          // [k][D]
    double* pc = pv + k * D;          // [k]
    double* rt = pc + k;              // [k][64] responsibilities
    double* tile = rt + k * 64;       // [64][D+1]
    const ls_tile tl = tiles[blockIdx.x];
    const int lane = threadIdx.x;
    const int64_t pbase = (int64_t)tl.state * k;
    for (int i = lane; i < k * D; i += 64) { pm[i] = mean[pbase * D + i]; pv[i] = ivar[pbase * D + i]; }
    for (int i = lane; i < k; i += 64) pc[i] = logc[pbase + i];
    double x[DR];
    stage_tile<DR>(X, tl.first, tl.count, D, tile, x);
    double ll = 0;   // (the frames stay in LDS: the centred sums below read them again)
    {
        double mx = -INFINITY;
        bool bad = false;
#pragma unroll 1
        for (int c = 0; c < k; ++c) {
            double q = 0;
#pragma unroll
            for (int d = 0; d < DR; ++d) if (d < D) { const double t = x[d] - pm[c * D + d]; q = fma(t * pv[c * D + d], t, q); }
            double l = pc[c] - 0.5 * q;
            bad |= (l != l);
            if (-0.5 * q < LS_LN_UNDERFLOW || l < LS_LN_UNDERFLOW) l = -INFINITY;   // the reference's linear-domain product is 0
            rt[c * 64 + lane] = l;
            mx = fmax(mx, l);
        }
        double sum = 0;
        for (int c = 0; c < k; ++c) {
            const double e = (mx == -INFINITY) ? 0.0 : exp(rt[c * 64 + lane] - mx);
            rt[c * 64 + lane] = e;
            sum += e;
        }
        const double inv = bad ? NAN : (sum > 0 ? 1.0 / sum : 0.0);
        const bool act = lane < tl.count;
        for (int c = 0; c < k; ++c) rt[c * 64 + lane] = act ? (bad ? NAN : rt[c * 64 + lane] * inv) : 0.0;
        if (act) ll = bad ? NAN : (sum > 0 ? mx + log(sum) : 0.0);
    }
    __syncthreads();
    const int Wd = 1 + 2 * D;
    double* out = partial + (int64_t)blockIdx.x * (k * Wd + 1);
    for (int p = lane; p < k * (D + 1); p += 64) {      // lane = (component, dimension) pair; dimension D = the occupancy
        const int c = p / (D + 1), d = p - c * (D + 1);
        const double* r = rt + c * 64;
        double a1 = 0, a2 = 0;
        if (d == D) {
            for (int f = 0; f < tl.count; ++f) a1 += r[f];
            out[c * Wd] = a1;
        } else {
            const double m = pm[c * D + d];
            for (int f = 0; f < tl.count; ++f) {
                const double xv = tile[f * (D + 1) + d] - m;    // centred on the current mean
                const double rx = r[f] * xv;
                a1 += rx;
                a2 = fma(rx, xv, a2);
            }
            out[c * Wd + 1 + d] = a1;
            out[c * Wd + 1 + D + d] = a2;
        }
    }
    // log-likelihood of the tile
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ll += __shfl_xor(ll, o);
    if (lane == 0) out[k * Wd] = ll;
}

int build_tiles(int S, const int64_t* seg_off, const uint8_t* active, std::vector<ls_tile>& tiles, std::vector<int32_t>& tile_ptr) {
    tile_ptr.assign(S + 1, 0);
    for (int s = 0; s < S; ++s) {
        tile_ptr[s] = (int32_t)tiles.size();
        if (active && !active[s]) continue;
        for (int64_t f = seg_off[s]; f < seg_off[s + 1]; f += 64)
            tiles.push_back(ls_tile{f, s, (int32_t)std::min<int64_t>(64, seg_off[s + 1] - f)});
    }
    tile_ptr[S] = (int32_t)tiles.size();
    return GH_OK;
}

int check_segments(const gh_batch* b, int S, const int64_t* seg_off, const char* who) {
    GH_REQUIRE(b->dtype == GH_F64, "%s: needs an fp64 batch", who);
    GH_REQUIRE(S > 0 && seg_off && seg_off[0] >= 0 && seg_off[S] <= b->N, "%s: segment offsets outside the batch", who);
    for (int s = 0; s < S; ++s) GH_REQUIRE(seg_off[s + 1] >= seg_off[s], "%s: segment offsets not monotone at %d", who, s);
    GH_REQUIRE(b->D <= LS_MAXD, "%s: D=%d (<= %d supported)", who, b->D, LS_MAXD);
    return GH_OK;
}

}  // namespace

static int resident_clusters(gh_ctx* ctx, gh_batch* b, bool reset) {
    if (!b->d_clusters && b->N > 0) {
        GH_HIP(hipMalloc((void**)&b->d_clusters, (size_t)b->N * 4));
        reset = true;
    }
    if (reset && b->N > 0) GH_HIP(hipMemsetAsync(b->d_clusters, 0xFF, (size_t)b->N * 4, ctx->stream));
    return GH_OK;
}

extern "C" int gh_kmeans_resident_clusters(gh_ctx* ctx, gh_batch* b, int reset, int32_t* out) {
    GH_REQUIRE(ctx && b, "gh_kmeans_resident_clusters: NULL argument");
    GH_HIP(hipSetDevice(ctx->device));
    const int rc = resident_clusters(ctx, b, reset != 0);
    if (rc) return rc;
    if (out && b->N > 0) GH_HIP(hipMemcpyAsync(out, b->d_clusters, (size_t)b->N * 4, hipMemcpyDeviceToHost, ctx->stream));
    GH_HIP(hipStreamSynchronize(ctx->stream));
    return GH_OK;
}

extern "C" int gh_kmeans_assign_multi(gh_ctx* ctx, const gh_batch* b, int S, const int64_t* seg_off, const uint8_t* active,
                                      int k, const double* centroids, const double* var, int32_t* clusters_io,
                                      int32_t* out_changed, double* out_sums) {
    GH_REQUIRE(ctx && b && centroids, "gh_kmeans_assign_multi: NULL argument");
    GH_REQUIRE(k > 0 && k <= 64, "gh_kmeans_assign_multi: k=%d (1..64)", k);
    int rc = check_segments(b, S, seg_off, "gh_kmeans_assign_multi");
    if (rc) return rc;
    GH_HIP(hipSetDevice(ctx->device));
    const int D = b->D;
    std::vector<ls_tile> tiles;
    std::vector<int32_t> tile_ptr;
    build_tiles(S, seg_off, active, tiles, tile_ptr);
    if (out_changed) for (int s = 0; s < S; ++s) out_changed[s] = 0;
    if (out_sums) for (int64_t i = 0; i < (int64_t)S * k * (D + 1); ++i) out_sums[i] = 0.0;
    if (tiles.empty()) return GH_OK;
    std::vector<double> logdet(S, 0.0);
    if (var)
        for (int s = 0; s < S; ++s) {
            double prod = 1.0;
            for (int d = 0; d < D; ++d) prod *= var[(size_t)s * D + d];
            logdet[s] = 0.5 * std::log(std::pow(2.0 * M_PI, D) * prod);   // hmm_state.py:58
        }
    const int64_t N = b->N;
    const int plen = k * (D + 1);
    ls_tile* d_tiles; double *d_cent, *d_var = nullptr, *d_ld, *d_part = nullptr, *d_sums = nullptr;
    int32_t *d_cl, *d_changed, *d_tptr;
    Carver cv;
    cv.add(&d_tiles, tiles.size()); cv.add(&d_cent, (size_t)S * k * D); cv.add(&d_ld, S);
    if (clusters_io) cv.add(&d_cl, N);
    cv.add(&d_changed, S); cv.add(&d_tptr, S + 1);
    if (var) cv.add(&d_var, (size_t)S * D);
    int64_t* d_segoff = nullptr;
    uint8_t* d_active = nullptr;
    int32_t *d_counts = nullptr, *d_tilecnt = nullptr, *d_cbase = nullptr, *d_lists = nullptr;
    if (out_sums) {
        cv.add(&d_sums, (size_t)S * plen); cv.add(&d_segoff, S + 1); cv.add(&d_counts, (size_t)S * k); cv.add(&d_cbase, (size_t)S * k);
        cv.add(&d_tilecnt, tiles.size() * (size_t)k); cv.add(&d_lists, N);
        if (active) cv.add(&d_active, S);
    }
    rc = cv.commit(ctx);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    GH_HIP(hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(ls_tile), hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(d_cent, centroids, (size_t)S * k * D * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(d_ld, logdet.data(), (size_t)S * 8, hipMemcpyHostToDevice, st));
    if (clusters_io) GH_HIP(hipMemcpyAsync(d_cl, clusters_io, (size_t)N * 4, hipMemcpyHostToDevice, st));
    else {   // assignments resident in the batch
        rc = resident_clusters(ctx, const_cast<gh_batch*>(b), false);
        if (rc) return rc;
        d_cl = b->d_clusters;
    }
    GH_HIP(hipMemsetAsync(d_changed, 0, (size_t)S * 4, st));
    if (d_counts) { GH_HIP(hipMemsetAsync(d_counts, 0, (size_t)S * k * 4, st)); GH_HIP(hipMemsetAsync(d_cbase, 0, (size_t)S * k * 4, st)); }
    if (var) GH_HIP(hipMemcpyAsync(d_var, var, (size_t)S * D * 8, hipMemcpyHostToDevice, st));
    const size_t lds = ((size_t)k * D + D + 64 * (size_t)(D + 1)) * 8 + 64 * 4 + 16;
    GH_REQUIRE(lds <= 150 * 1024, "gh_kmeans_assign_multi: k=%d x D=%d does not fit LDS", k, D);
    const dim3 grid((unsigned)tiles.size()), blk(64);
#define GH_KM(DR) hipLaunchKernelGGL((kmeans_multi_kernel<DR>), grid, blk, lds, st, (const double*)b->feats, D, k, d_tiles, d_cent, \
                                     d_var, d_ld, d_cl, out_changed ? d_changed : nullptr, d_tilecnt)
    if (D <= 16) GH_KM(16); else if (D <= 40) GH_KM(40); else GH_KM(64);
#undef GH_KM
    GH_HIP(hipGetLastError());
    if (out_sums) {
        GH_HIP(hipMemcpyAsync(d_segoff, seg_off, (size_t)(S + 1) * 8, hipMemcpyHostToDevice, st));
        if (active) GH_HIP(hipMemcpyAsync(d_active, active, (size_t)S, hipMemcpyHostToDevice, st));
        GH_HIP(hipMemsetAsync(d_sums, 0, (size_t)S * plen * 8, st));
        GH_HIP(hipMemcpyAsync(d_tptr, tile_ptr.data(), (size_t)(S + 1) * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(kmeans_scan_kernel, dim3((unsigned)S), dim3(64), 0, st, d_tptr, k, d_tilecnt, d_counts, d_cbase);
        hipLaunchKernelGGL(kmeans_scatter_kernel, grid, blk, 0, st, d_tiles, k, d_cl, d_segoff, d_tilecnt, d_cbase, d_lists);
        hipLaunchKernelGGL(kmeans_rowsum_kernel, dim3((unsigned)S, (unsigned)k, (unsigned)((D + 63) / 64)), dim3(64), 0, st,
                           (const double*)b->feats, D, k, d_segoff, d_active, d_lists, d_counts, d_cbase, d_sums);
        GH_HIP(hipGetLastError());
        GH_HIP(hipMemcpyAsync(out_sums, d_sums, (size_t)S * plen * 8, hipMemcpyDeviceToHost, st));
    }
    if (clusters_io) GH_HIP(hipMemcpyAsync(clusters_io, d_cl, (size_t)N * 4, hipMemcpyDeviceToHost, st));
    if (out_changed) GH_HIP(hipMemcpyAsync(out_changed, d_changed, (size_t)S * 4, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}

extern "C" int gh_em_accumulate_multi(gh_ctx* ctx, const gh_batch* b, int S, const int64_t* seg_off, const uint8_t* active,
                                      int k, const double* mean, const double* var, const double* weight,
                                      double* out_stats, double* out_loglik, double* stats_dev) {
    GH_REQUIRE(ctx && b && mean && var && weight && (out_stats || stats_dev), "gh_em_accumulate_multi: NULL argument");
    GH_REQUIRE(k > 0 && k <= 32, "gh_em_accumulate_multi: k=%d (1..32)", k);
    int rc = check_segments(b, S, seg_off, "gh_em_accumulate_multi");
    if (rc) return rc;
    GH_HIP(hipSetDevice(ctx->device));
    const int D = b->D, Wd = 1 + 2 * D, plen = k * Wd + 1;
    std::vector<ls_tile> tiles;
    std::vector<int32_t> tile_ptr;
    build_tiles(S, seg_off, active, tiles, tile_ptr);
    std::vector<double> ivar((size_t)S * k * D), logc((size_t)S * k);
    const double log2pi = std::log(2.0 * M_PI);
    for (int s = 0; s < S; ++s) {
        if (active && !active[s]) continue;
        for (int c = 0; c < k; ++c) {
            double sl = 0;
            const size_t g = (size_t)s * k + c;
            for (int d = 0; d < D; ++d) {
                const double v = var[g * D + d];
                GH_REQUIRE(v != 0, "gh_em_accumulate_multi: var[%d,%d,%d]=%g (singular covariance)", s, c, d, v);
                ivar[g * D + d] = 1.0 / v;
                sl += std::log(v);
            }
            logc[g] = std::log(weight[g]) - 0.5 * (D * log2pi + sl);
        }
    }
    double *d_mean, *d_ivar, *d_logc, *d_part, *d_out;
    ls_tile* d_tiles;
    int32_t* d_tptr;
    Carver cv;
    cv.add(&d_tiles, std::max<size_t>(1, tiles.size())); cv.add(&d_mean, (size_t)S * k * D); cv.add(&d_ivar, (size_t)S * k * D);
    cv.add(&d_logc, (size_t)S * k); cv.add(&d_part, std::max<size_t>(1, tiles.size()) * plen); cv.add(&d_out, (size_t)S * plen);
    cv.add(&d_tptr, S + 1);
    rc = cv.commit(ctx);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    if (!tiles.empty()) GH_HIP(hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(ls_tile), hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(d_mean, mean, (size_t)S * k * D * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(d_ivar, ivar.data(), ivar.size() * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(d_logc, logc.data(), logc.size() * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(d_tptr, tile_ptr.data(), (size_t)(S + 1) * 4, hipMemcpyHostToDevice, st));
    const size_t lds = ((size_t)2 * k * D + k + (size_t)k * 64 + 64 * (size_t)(D + 1)) * 8 + 16;
    GH_REQUIRE(lds <= 150 * 1024, "gh_em_accumulate_multi: k=%d x D=%d does not fit LDS", k, D);
    if (!tiles.empty()) {
        const dim3 grid((unsigned)tiles.size()), blk(64);
#define GH_EM(DR) hipLaunchKernelGGL((em_multi_kernel<DR>), grid, blk, lds, st, (const double*)b->feats, D, k, d_tiles, d_mean, d_ivar, \
                                     d_logc, d_part)
        if (D <= 16) GH_EM(16); else if (D <= 40) GH_EM(40); else GH_EM(64);
#undef GH_EM
        GH_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(tiles_reduce_kernel, dim3((unsigned)((plen + 127) / 128), (unsigned)S), dim3(128), 0, st, d_part, d_tptr,
                       plen, d_out);
    GH_HIP(hipGetLastError());
    std::vector<double> host((size_t)S * plen);
    GH_HIP(hipMemcpyAsync(host.data(), d_out, host.size() * 8, hipMemcpyDeviceToHost, st));
    if (stats_dev)   // [S, k, 1+2D] packed (without the log-likelihood column) for a device-side all-reduce
        GH_HIP(hipMemcpy2DAsync(stats_dev, (size_t)k * Wd * 8, d_out, (size_t)plen * 8, (size_t)k * Wd * 8, S,
                                hipMemcpyDeviceToDevice, st));
    GH_HIP(hipStreamSynchronize(st));
    for (int s = 0; s < S; ++s) {
        if (out_stats) memcpy(out_stats + (size_t)s * k * Wd, host.data() + (size_t)s * plen, (size_t)k * Wd * 8);
        if (out_loglik) out_loglik[s] = host[(size_t)s * plen + k * Wd];
    }
    return GH_OK;
}
