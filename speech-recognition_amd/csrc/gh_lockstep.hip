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
#include "gh_refit.h"

namespace {

constexpr int LS_MAXD = 64;   // feature dimensions a lane keeps in registers

struct ls_tile { int64_t first; int32_t state, count; };

// stage a tile of <= 64 frames: coalesced global -> LDS [64][D | 1], then lane = frame -> registers.
// Row stride TS = D | 1: odd, so the lane = frame reads below touch 32 banks (stride D + 1 = 40 doubles for D = 39 put them
// on 4: an 8-way conflict per read); for odd D the tile is simply the frames as they lie in memory, element i at slot i.
__device__ __forceinline__ int tile_stride(int D) { return D | 1; }
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
// number of frames of every (state, cluster) and where its list starts inside the state's segment.
// One block of 4 waves per state; a wave takes the clusters c = wave, wave + 4, ... and scans a cluster's tiles 64 at a
// time (lane = tile, shuffle prefix sum, running carry) -- the first version walked the ~440 tiles of a state one after
// the other per cluster thread (44 us per launch, latency bound).
__global__ __launch_bounds__(256) void kmeans_scan_kernel(const int32_t* __restrict__ tile_ptr /*[S+1]*/, int k, int32_t* __restrict__ tilecnt /*[tiles][k] -> offsets*/,
                                   int32_t* __restrict__ counts /*[S,k]*/, int32_t* __restrict__ cbase /*[S,k]*/,
                                   const uint8_t* __restrict__ active = nullptr) {
    const int s = blockIdx.x;
    if (active && !active[s]) return;      // (a state that has stopped keeps its lists and counts)
    __shared__ int tot[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t0 = tile_ptr[s], t1 = tile_ptr[s + 1];
    for (int c = wave; c < k; c += 4) {
        int run = 0;
        for (int t = t0; t < t1; t += 64) {
            const bool in = t + lane < t1;
            const int v = in ? tilecnt[(int64_t)(t + lane) * k + c] : 0;
            int inc = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_up(inc, o);
                if (lane >= o) inc += up;
            }
            if (in) tilecnt[(int64_t)(t + lane) * k + c] = run + inc - v;
            run += __shfl(inc, 63);
        }
        if (lane == 0) tot[c] = run;
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c < k) {
        int base = 0;
        for (int j = 0; j < c; ++j) base += tot[j];
        counts[s * k + c] = tot[c];
        cbase[s * k + c] = base;
    }
}

// every tile writes the (segment-local) numbers of its frames into their clusters' lists, keeping frame order
__global__ __launch_bounds__(64) void kmeans_scatter_kernel(const ls_tile* __restrict__ tiles, int k, const int32_t* __restrict__ clusters,
                                                            const int64_t* __restrict__ seg_off, const int32_t* __restrict__ tileoff,
                                                            const int32_t* __restrict__ cbase, int32_t* __restrict__ lists /*[N]*/,
                                                            const uint8_t* __restrict__ active = nullptr) {
    const ls_tile tl = tiles[blockIdx.x];
    if (active && !active[tl.state]) return;
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

// grid (S, k), block = D rounded up to waves: the chain of (state, cluster, dimension = thread).
// MODE 0: sum of x (and the count in column D); 1: sum of (x - centre)^2 with centre = csum / count (the two-pass
// variance of a partition); 2: sum of x^2.  sstride: doubles per state in `sums` (>= k (D + 1)).
template <int MODE>
__global__ __launch_bounds__(64) void kmeans_rowsum_kernel(const double* __restrict__ X, int D, int k,
                                                           const int64_t* __restrict__ seg_off, const uint8_t* __restrict__ active,
                                                           const int32_t* __restrict__ lists, const int32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ cbase, double* __restrict__ sums, int sstride,
                                                           const double* __restrict__ csum = nullptr, int cstride = 0,
                                                           int chunk = 0, int nch = 0) {
    const int s = blockIdx.x, c = blockIdx.y;
    if (active && !active[s]) return;
    // chunk == 0: the whole list of (state, cluster) in frame order -- numpy's sum, bit for bit (D <= 64: one block);
    // chunk > 0: blockIdx.z-th piece of `chunk` list entries into partial[s][c][z][D+1] (summed by fit_chunk_reduce_kernel:
    // a FIXED order, but not numpy's -- the lock-step iterations use it, the final centroids never)
    const int d = threadIdx.x, lane = threadIdx.x;
    const bool live = d < D;
    const int64_t f0 = seg_off[s];
    const int n_all = counts[s * k + c];
    const int skip = chunk > 0 ? blockIdx.z * chunk : 0;
    const int n = chunk > 0 ? min(chunk, n_all - skip) : n_all;
    double* out = chunk > 0 ? sums + (((int64_t)s * k + c) * nch + blockIdx.z) * (D + 1)
                            : sums + (int64_t)s * sstride + (int64_t)c * (D + 1);
    if (d == 0 && MODE == 0 && chunk == 0) out[D] = (double)n_all;
    if (n <= 0) { if (live && MODE != 0 && chunk == 0) out[d] = 0.0; return; }
    double ctr = 0.0;
    if (MODE == 1) ctr = csum[(int64_t)s * cstride + (int64_t)c * (D + 1) + (live ? d : 0)] / (double)n_all;
    // Rows through BUFFER loads: address = descriptor base (the state's first frame) + scalar offset (the row, handed
    // out of the lane-held list by v_readlane) + vector offset (the lane's dimension) -- readlane, buffer_load, add: three
    // instructions per frame.  With flat loads the 64-bit row address was rebuilt by scalar arithmetic for every frame
    // (~10 instructions): the kernel was ISSUE bound at ~50 cycles per frame (292 us per launch for the longest chains),
    // not memory bound -- fetching the frame numbers a round earlier changed nothing.
    const int64_t seg_bytes = (seg_off[s + 1] - f0) * (int64_t)D * 8;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(X + f0 * D), 0, (int)seg_bytes, 0x00020000);
    const int voff = (live ? d : 0) * 8;
    const int row_bytes = D * 8;
    const int32_t* li = lists + f0 + cbase[s * k + c] + skip;
    int id = li[min(lane, n - 1)] * row_bytes;
    int id_next = li[min(RS_B + lane, n - 1)] * row_bytes;    // the frame numbers travel two rounds ahead of the adds
    typedef int rs_v2i __attribute__((ext_vector_type(2)));
    auto row = [&](int ids, int j) -> double {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, __builtin_amdgcn_readlane(ids, j), 0));
    };
    double x[RS_B];
#pragma unroll
    for (int j = 0; j < RS_B; ++j) x[j] = row(id, j);
    auto term = [&](double v) -> double {
        if (MODE == 0) return v;
        if (MODE == 1) { const double t = v - ctr; return t * t; }
        return v * v;
    };
    double acc = 0.0;
    int b = 0;
    for (; b + RS_B <= n; b += RS_B) {                        // full rounds: the chain is one add per frame
        id = id_next;                                         // entries b + 64 .. (clamped: re-reads the last frame, never summed)
        id_next = li[min(b + 2 * RS_B + lane, n - 1)] * row_bytes;
#pragma unroll
        for (int j = 0; j < RS_B; ++j) {
            acc += term(x[j]);
            x[j] = row(id, j);
        }
    }
    const int left = n - b;                                   // the last, partial round
#pragma unroll
    for (int j = 0; j < RS_B; ++j) acc += (j < left) ? term(x[j]) : 0.0;
    if (live) out[d] = acc;
}

// RECIP: (t * (1/var)) * t instead of the reference's t / var * t (mahalanobis, hmm_state.py:58: `m / variance * m`).  The
// division is ~25 fp64 instructions per (cluster, dimension) -- 8 x 39 of them per frame made this kernel the longest of
// the refit (437 us per sweep of 1.4 M frames).  The two forms differ in the last bit of a term; an assignment changes
// only where two centroids tie to ~1e-16 relative.  Opt-in (GMMHMM_KMEANS_RECIP=1) since round 5: the device-resident refit
// now runs the streaming kernel of gh_refit_mfma.hip (k <= 8), which re-tests near-ties with the division, and what is left
// to this kernel (k > 8, GMMHMM_REFIT=tiles, the call-by-call gh_kmeans_assign_multi) keeps the reference's operations.
// TWO waves per tile, like em_multi_kernel below: the waves share the tile (here only a transposition buffer) and the
// centroids; each takes half of the clusters, and wave 0 joins the two candidates with the rule of the one-wave loop
// (np.argmin: the first minimum, a NaN before everything) -- the same assignment, the chains of 39 dependent terms per
// cluster running two abreast on twice the waves per CU.
template <int DR, bool RECIP = false>
__global__ __launch_bounds__(128) void kmeans_multi_kernel(const double* __restrict__ X, int D, int k, const ls_tile* __restrict__ tiles,
                                                           const double* __restrict__ cent /*[S,k,D]*/,
                                                           const double* __restrict__ var /*[S,D] or null*/,
                                                           const double* __restrict__ logdet /*[S]*/,
                                                           int32_t* __restrict__ clusters /*[N] in/out*/,
                                                           int32_t* __restrict__ changed /*[S] or null*/,
                                                           int32_t* __restrict__ counts /*[tiles][k] or null*/,
                                                           const uint8_t* __restrict__ active = nullptr, int vstride = 0) {
    static_assert(DR % 2 == 0, "the tile is staged by 128 threads");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* sc = sm;                  // [k][D]
    double* sv = sc + k * D;          // [D]
    double* tile = sv + D;            // [64][D | 1]; afterwards: wave 1's candidates
    const ls_tile tl = tiles[blockIdx.x];
    if (active && !active[tl.state]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int TS = tile_stride(D);
    {
        const double* src = X + tl.first * D;
        const int nelem = tl.count * D;
        double r[DR / 2];
#pragma unroll
        for (int it = 0; it < DR / 2; ++it) {
            const int i = tid + 128 * it;
            r[it] = src[i < nelem ? i : nelem - 1];
        }
        const double* c0 = cent + (int64_t)tl.state * k * D;
#pragma unroll 4
        for (int i = tid; i < k * D; i += 128) sc[i] = c0[i];
        if (var && tid < D) {                         // (vstride: [S,k,D] variances, cluster 0's row; D <= 64)
            const double v = var[(int64_t)tl.state * (vstride ? vstride : D) + tid];
            sv[tid] = RECIP ? 1.0 / v : v;
        }
        if (D & 1) {
#pragma unroll
            for (int it = 0; it < DR / 2; ++it) if (tid + 128 * it < nelem) tile[tid + 128 * it] = r[it];
        } else {
            int f = tid / D, d = tid - f * D;
            const int q128 = 128 / D, r128 = 128 - q128 * D;
#pragma unroll
            for (int it = 0; it < DR / 2; ++it) {
                if (tid + 128 * it < nelem) tile[tid + 128 * it + f] = r[it];
                d += r128;
                f += q128;
                if (d >= D) { d -= D; ++f; }
            }
        }
    }
    __syncthreads();
    double x[DR], svr[DR];
#pragma unroll
    for (int d = 0; d < DR; ++d) {
        x[d] = (d < D && lane < tl.count) ? tile[lane * TS + d] : 0.0;
        svr[d] = (var && d < D) ? sv[d] : 1.0;        // the state's (inverse) variances: out of LDS once, not once per cluster
    }
    __syncthreads();                                  // (the tile has been read: it becomes the exchange buffer)
    const double ld = var ? logdet[tl.state] : 0.0;
    const int kh = (k + 1) >> 1, c_lo = wv * kh, c_hi = (c_lo + kh < k) ? c_lo + kh : k;
    double best = 0;
    int bi = -1;                                      // (-1: this wave has no cluster, k = 1)
    for (int c = c_lo; c < c_hi; ++c) {
        double q = 0, dist;
        if (var) {
#pragma unroll
            for (int d = 0; d < DR; ++d) if (d < D) {
                const double t = sc[c * D + d] - x[d];
                if (RECIP) q += t * svr[d] * t; else q += t / svr[d] * t;
            }
            dist = ld + 0.5 * q;                                   // mahalanobis(centroid, x, cov[0]), kmeans.py:183
        } else {
#pragma unroll
            for (int d = 0; d < DR; ++d) if (d < D) { const double t = sc[c * D + d] - x[d]; q = fma(t, t, q); }
            dist = sqrt(q);
        }
        if (c == c_lo || dist < best || (dist != dist && best == best)) { best = dist; bi = c; }   // np.argmin
    }
    double* xb = tile;                                             // [64] wave 1's best distance
    int* xi = reinterpret_cast<int*>(tile + 64);                   // [64] ... and its cluster
    if (wv == 1) { xb[lane] = best; xi[lane] = bi; }
    __syncthreads();
    if (wv == 1) return;
    {
        const double d1 = xb[lane];
        const int b1 = xi[lane];
        if (b1 >= 0 && (d1 < best || (d1 != d1 && best == best))) { best = d1; bi = b1; }
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

// partial[tile][len] summed over the tiles of each state -> out[state][len].  One block per (state, 64 columns): wave w
// takes the tiles t0 + w, t0 + w + 4, ... with four accumulators per thread (16 loads in flight per column instead of
// one thread walking all ~440 tiles: 105 -> ~10 us), and the 16 partial sums are combined in a fixed order --
// deterministic for a given tile list, which is all the EM statistics need (their summation order is ours).
__global__ __launch_bounds__(256) void tiles_reduce_kernel(const double* __restrict__ partial, const int32_t* __restrict__ tile_ptr /*[S+1]*/, int len,
                                    double* __restrict__ out, const uint8_t* __restrict__ active = nullptr) {
    const int s = blockIdx.y;
    if (active && !active[s]) return;
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const int t0 = tile_ptr[s], t1 = tile_ptr[s + 1];
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (i < len) {
        const double* src = partial + i;
        int t = t0 + wave;
        for (; t + 12 < t1; t += 16) {
            a0 += src[(int64_t)t * len];
            a1 += src[(int64_t)(t + 4) * len];
            a2 += src[(int64_t)(t + 8) * len];
            a3 += src[(int64_t)(t + 12) * len];
        }
        for (; t < t1; t += 4) a0 += src[(int64_t)t * len];
    }
    red[wave][lane] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (wave == 0 && i < len) out[(int64_t)s * len + i] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ------------------------------------------------------------------------------------------------ EM E-step
constexpr double LS_LN_UNDERFLOW = -745.1332191019412;   // exp(x) rounds to +0 in fp64 below this (see gh_train.hip)

// TWO waves per tile (round 4, after tools/emv_variants.sh: the kernel is arithmetic on dependent chains -- 39 fma per
// density, 64 adds per sum -- at 1.25 waves per SIMD, held there by its 30 KB of LDS): the waves share the tile and the
// parameters in LDS; each takes half of the components in phase 1 and half of the (component, dimension) sums in phase 2,
// so every chain is as long as before but twice as many run side by side, on twice the waves per CU.  Every number is
// computed by the same operations in the same order as with one wave: the results are bitwise those of the one-wave kernel.
#ifndef GH_EM_WAVES
#define GH_EM_WAVES 2                 // waves per tile (1, 2 or 4; 4: 434 -> 413 us per full pass, nothing at the application)
#endif
template <int DR>
__global__ __launch_bounds__(64 * GH_EM_WAVES) void em_multi_kernel(const double* __restrict__ X, int D, int k, const ls_tile* __restrict__ tiles,
                                                       const double* __restrict__ mean /*[S,k,D]*/,
                                                       const double* __restrict__ ivar /*[S,k,D]*/,
                                                       const double* __restrict__ logc /*[S,k]*/,
                                                       double* __restrict__ partial /*[tiles][k*(1+2D) + 1]*/,
                                                       const uint8_t* __restrict__ active = nullptr) {
    constexpr int NW = GH_EM_WAVES, NT = 64 * NW;
    static_assert(DR % NW == 0, "the tile is staged by all the block's threads");
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* pm = sm;                  // [k][D]
    double* pv = pm + k * D;          // [k][D]
    double* pc = pv + k * D;          // [k]
    double* rt = pc + k;              // [k][64] log-densities, then responsibilities
    double* tile = rt + k * 64;       // [64][D | 1]
    const ls_tile tl = tiles[blockIdx.x];
    if (active && !active[tl.state]) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int TS = tile_stride(D);
    const int64_t pbase = (int64_t)tl.state * k;
    // ---- the tile's frames (all loads in flight first, see stage_tile) and the state's parameters ----
    {
        const double* src = X + tl.first * D;
        const int nelem = tl.count * D;
        double r[DR / NW];
#pragma unroll
        for (int it = 0; it < DR / NW; ++it) {
            const int i = tid + NT * it;
            r[it] = src[i < nelem ? i : nelem - 1];
        }
#pragma unroll 4
        for (int i = tid; i < k * D; i += NT) { pm[i] = mean[pbase * D + i]; pv[i] = ivar[pbase * D + i]; }
        if (tid < k) pc[tid] = logc[pbase + tid];
        if (D & 1) {
#pragma unroll
            for (int it = 0; it < DR / NW; ++it) if (tid + NT * it < nelem) tile[tid + NT * it] = r[it];
        } else {                                 // one pad slot per frame: element i of frame f at slot i + f
            int f = tid / D, d = tid - f * D;
            const int qn = NT / D, rn = NT - qn * D;
#pragma unroll
            for (int it = 0; it < DR / NW; ++it) {
                if (tid + NT * it < nelem) tile[tid + NT * it + f] = r[it];
                d += rn;
                f += qn;
                if (d >= D) { d -= D; ++f; }
            }
        }
    }
    __syncthreads();
    double x[DR];
#pragma unroll
    for (int d = 0; d < DR; ++d) x[d] = (d < D && lane < tl.count) ? tile[lane * TS + d] : 0.0;
    // ---- phase 1: log-densities of this wave's half of the components, frame = lane ----
    const int kh = (k + NW - 1) / NW, c_lo = (wv * kh < k) ? wv * kh : k, c_hi = (c_lo + kh < k) ? c_lo + kh : k;
#pragma unroll 1
    for (int c = c_lo; c < c_hi; ++c) {
        double q = 0;
#pragma unroll
        for (int d = 0; d < DR; ++d) if (d < D) { const double t = x[d] - pm[c * D + d]; q = fma(t * pv[c * D + d], t, q); }
        double l = pc[c] - 0.5 * q;
        const bool nan_l = l != l;
        if (-0.5 * q < LS_LN_UNDERFLOW || l < LS_LN_UNDERFLOW) l = -INFINITY;   // the reference's linear-domain product is 0
        rt[c * 64 + lane] = nan_l ? NAN : l;     // (a NaN stays visible to both waves: everything of the frame is NaN then)
    }
    __syncthreads();
    // ---- both waves: maximum and sum over ALL components, in component order (the one-wave kernel's order) ----
    double mx = -INFINITY, sum = 0, ll = 0;
    bool bad = false;
    for (int c = 0; c < k; ++c) { const double l = rt[c * 64 + lane]; bad |= (l != l); mx = fmax(mx, l); }
    for (int c = 0; c < k; ++c) {
        const double e = (mx == -INFINITY) ? 0.0 : exp(rt[c * 64 + lane] - mx);
        sum += e;
    }
    const double inv = bad ? NAN : (sum > 0 ? 1.0 / sum : 0.0);
    const bool act = lane < tl.count;
    if (act && wv == 0) ll = bad ? NAN : (sum > 0 ? mx + log(sum) : 0.0);
    __syncthreads();                              // (every log-density has been read: the responsibilities may replace them)
    for (int c = c_lo; c < c_hi; ++c) {
        const double e = (mx == -INFINITY) ? 0.0 : exp(rt[c * 64 + lane] - mx);
        rt[c * 64 + lane] = act ? (bad ? NAN : e * inv) : 0.0;
    }
    __syncthreads();
    // ---- phase 2: thread = (component, dimension) pair; dimension D = the occupancy; the sums run over the frames in order ----
    const int Wd = 1 + 2 * D;
    double* out = partial + (int64_t)blockIdx.x * (k * Wd + 1);
    for (int p = tid; p < k * (D + 1); p += NT) {
        const int c = p / (D + 1), d = p - c * (D + 1);
        const double* r = rt + c * 64;
        double a1 = 0, a2 = 0;
        if (d == D) {
            for (int f = 0; f < tl.count; ++f) a1 += r[f];
            out[c * Wd] = a1;
        } else {
            const double m = pm[c * D + d];
            for (int f = 0; f < tl.count; ++f) {
                const double xv = tile[f * TS + d] - m;         // centred on the current mean
                const double rx = r[f] * xv;
                a1 += rx;
                a2 = fma(rx, xv, a2);
            }
            out[c * Wd + 1 + d] = a1;
            out[c * Wd + 1 + D + d] = a2;
        }
    }
    // log-likelihood of the tile (wave 0 holds it)
    if (wv == 0) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ll += __shfl_xor(ll, o);
        if (lane == 0) out[k * Wd] = ll;
    }
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
    for (int s = 0; s < S; ++s)    // (the centroid sums address a state's rows by 31-bit byte offsets from its first frame)
        GH_REQUIRE((seg_off[s + 1] - seg_off[s]) * (int64_t)b->D * 8 < ((int64_t)1 << 31), "%s: state %d holds %lld frames (< 2 GiB per state)",
                   who, s, (long long)(seg_off[s + 1] - seg_off[s]));
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
    const dim3 grid((unsigned)tiles.size()), blk(128);
#define GH_KM(DR) hipLaunchKernelGGL((kmeans_multi_kernel<DR>), grid, blk, lds, st, (const double*)b->feats, D, k, d_tiles, d_cent, \
                                     d_var, d_ld, d_cl, out_changed ? d_changed : nullptr, d_tilecnt, nullptr)
    if (D <= 16) GH_KM(16); else if (D <= 40) GH_KM(40); else GH_KM(64);
#undef GH_KM
    GH_HIP(hipGetLastError());
    if (out_sums) {
        GH_HIP(hipMemcpyAsync(d_segoff, seg_off, (size_t)(S + 1) * 8, hipMemcpyHostToDevice, st));
        if (active) GH_HIP(hipMemcpyAsync(d_active, active, (size_t)S, hipMemcpyHostToDevice, st));
        GH_HIP(hipMemsetAsync(d_sums, 0, (size_t)S * plen * 8, st));
        GH_HIP(hipMemcpyAsync(d_tptr, tile_ptr.data(), (size_t)(S + 1) * 4, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(kmeans_scan_kernel, dim3((unsigned)S), dim3(256), 0, st, d_tptr, k, d_tilecnt, d_counts, d_cbase, nullptr);
        hipLaunchKernelGGL(kmeans_scatter_kernel, grid, dim3(64), 0, st, d_tiles, k, d_cl, d_segoff, d_tilecnt, d_cbase, d_lists, nullptr);   // (one wave per tile)
        hipLaunchKernelGGL(kmeans_rowsum_kernel<0>, dim3((unsigned)S, (unsigned)k, (unsigned)((D + 63) / 64)), dim3(64), 0, st,
                           (const double*)b->feats, D, k, d_segoff, d_active, d_lists, d_counts, d_cbase, d_sums, plen, nullptr, 0);
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
        const dim3 grid((unsigned)tiles.size()), blk(64 * GH_EM_WAVES);
#define GH_EM(DR) hipLaunchKernelGGL((em_multi_kernel<DR>), grid, blk, lds, st, (const double*)b->feats, D, k, d_tiles, d_mean, d_ivar, \
                                     d_logc, d_part, nullptr)
        if (D <= 16) GH_EM(16); else if (D <= 40) GH_EM(40); else GH_EM(64);
#undef GH_EM
        GH_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(tiles_reduce_kernel, dim3((unsigned)((plen + 63) / 64), (unsigned)S), dim3(256), 0, st, d_part, d_tptr,
                       plen, d_out, nullptr);
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

// =====================================================================================================================
// gh_fit: the refit of ALL states -- binary-split k-means + mixture EM, hmm.py:97-124 / continuous_speech.py:114-142 --
// as a DEVICE-RESIDENT session.  Round 2 advanced the states in lock-step but kept the loop on the host: per k-means /
// EM iteration one synchronous call (tile lists rebuilt and uploaded, parameters uploaded, statistics copied back), the
// M-step and convergence test of GMM.em_update per state in numpy, the partition variances as np.cov per (state,
// cluster) -- 0.19 s + 0.3 s of every 0.75 s continuous_train iteration against 75 ms of kernels.  Here the tile list,
// centroids, variances, mixture parameters, "old" parameters of the allclose test, the active mask and the iteration
// counters live in HBM; one k-means / EM iteration is a handful of kernel launches on the context's stream with NO host
// synchronisation, and the host looks at one counter (states still active) every `check_every` iterations.  Per state
// the arithmetic and the stopping rule are those of the sequential algorithm, so a state's result does not depend on
// how many iterations the others take.
//   * partition variances (kmeans.py:171-177): two passes per (state, cluster) over the cluster's frames in frame order
//     -- mean (the sum numpy's np.average takes, bit for bit), then sum (x - mean)^2 / (n - 1).  np.cov forms the whole
//     D x D matrix with a BLAS product to read its diagonal: O(N D^2) instead of O(N D), and the diagonal's last bits
//     are the BLAS library's summation order (callers that want exactly those bits keep np.cov: compat_cov=True).
//   * sharded (comm != NULL): cluster sums + changed counts, cluster counts, partition sums and EM statistics are summed
//     over the ranks with gh_comm's all-reduce ON THE DEVICE BUFFERS, between the kernels that produce and consume them.
constexpr int FIT_CHUNK = 512;   // list entries per wave of the chunked cluster sums (see gh_fit_kmeans)
constexpr int FIT_TAIL_ITERS = 64;   // iterations a TAIL launch may run (its workgroups leave when their state stops)

struct gh_fit {
    gh_ctx* ctx;
    const gh_batch* b;
    int S, D, kmax, n_tiles;
    int64_t N;
    void* d_arena;
    size_t arena_bytes, act_bytes;      // sizes of d_arena / h_act (both go back to the context when the session closes)
    ls_tile* d_tiles;
    int32_t* d_tptr;
    int64_t* d_segoff;
    uint8_t* d_active;
    int32_t *d_ids, *d_tilecnt, *d_counts, *d_cbase, *d_lists, *d_changed, *d_iters;
    double *d_cent, *d_cov, *d_logdet, *d_sums, *d_sq, *d_psum;
    int nch;             // pieces of FIT_CHUNK list entries the longest state's lists are cut into (d_psum [S][kmax][nch][D+1])
    int* d_counter;      // [0] states still active after the last iteration, [1] error bits (16: zero variance)
    double *d_mean, *d_var, *d_weight, *d_ivar, *d_logc, *d_old_mu, *d_old_sigma, *d_old_w, *d_nframes, *d_part, *d_stats;
    int* h_pin;          // pinned [16]
    // streaming matrix-core form of the two per-iteration kernels (gh_refit_mfma.hip): items, packed operands, slabs
    bool mfma;           // k <= 8 and not switched off (GMMHMM_REFIT=tiles)
    bool tail;           // TAIL launches allowed (GMMHMM_REFIT_TAIL=0 switches them off)
    int n_items, kcap;   // kcap: the component count the buffers below are sized for (min(kmax, 8))
    rf_item* d_items;
    int32_t *d_iptr, *d_done, *d_tail_ids, *d_gen;
    double *d_P, *d_shift, *d_kscale, *d_rpart;
    std::vector<int32_t>* h_iptr;     // host copy of the item table's per-state ranges
    std::vector<int32_t>* h_tail_ids; // the items of the last tail launch (what d_tail_ids holds) ...
    std::vector<uint8_t>* h_tail_mask;   // ... and the active mask they were made from
    int tail_n;
    long n_tail_launches, n_tail_iters, n_plain_launches, n_tail_refused;   // (GMMHMM_REFIT_DEBUG=1 prints them when the session ends)
    uint8_t* h_act;      // pinned [S]: the active mask as of the last poll
};

namespace {

__global__ __launch_bounds__(64) void fit_tilecount_kernel(const ls_tile* __restrict__ tiles, int k, const int32_t* __restrict__ ids,
                                                           int32_t* __restrict__ tilecnt) {
    const ls_tile tl = tiles[blockIdx.x];
    const int lane = threadIdx.x;
    const bool act = lane < tl.count;
    const int id = act ? ids[tl.first + lane] : -1;
    for (int c = 0; c < k; ++c) {
        const int nc = __popcll(__ballot(act && id == c));
        if (lane == 0) tilecnt[(int64_t)blockIdx.x * k + c] = nc;
    }
}

__global__ void fit_u8_to_i32_kernel(const uint8_t* __restrict__ src, int64_t n, int32_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// mean of every state's frames in numpy's order (np.mean(seg, axis=0): the rows added one after the other)
__global__ __launch_bounds__(64) void fit_segsum_kernel(const double* __restrict__ X, int D, const int64_t* __restrict__ seg_off,
                                                        double* __restrict__ out /*[S, D+1]: sums | count*/) {
    const int s = blockIdx.x, d = blockIdx.y * 64 + threadIdx.x;
    const int64_t f0 = seg_off[s], n = seg_off[s + 1] - f0;
    if (d == 0) out[(int64_t)s * (D + 1) + D] = (double)n;
    if (d >= D) return;
    const double* col = X + f0 * D + d;
    double acc = 0.0;
    int64_t i = 0;
    for (; i + 32 <= n; i += 32) {
        double x[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) x[j] = col[(i + j) * D];
#pragma unroll
        for (int j = 0; j < 32; ++j) acc += x[j];
    }
    for (; i < n; ++i) acc += col[i * D];
    out[(int64_t)s * (D + 1) + d] = acc;
}

// The same sums with the rows FETCHED by eight waves and ADDED by one.  The sum of a state is one dependent chain per
// dimension (numpy's order), a few cycles per row once the row is there -- the kernel above waits for 32 rows, adds them, and
// only then asks for the next 32: ~1 us per 32 rows, 0.87 ms for the 28 000-frame states of continuous_train, with 50 waves on
// the whole chip.  Here a workgroup of 8 waves takes a state: in phase q wave w owns the rows [(8 q + w) 16, + 16) (lane =
// dimension, BUFFER loads: a row behind the state's end reads as 0 and is never added), keeps THREE phases in flight in
// registers (3 x 8 x 16 rows = 120 KB per state instead of 10), and passes a phase's rows through one of two LDS buffers to
// wave 0, which adds the 128 rows of phase q in order while the others already write phase q + 1 into the other buffer.
// One barrier per phase: a buffer is rewritten two phases later, behind the barrier wave 0 only reaches after its adds.
constexpr int SS_WAVES = 8, SS_ROWS = 16, SS_PHASE = SS_WAVES * SS_ROWS;
constexpr int SS_LIST_PAD = 2048;      // list entries a LIST chain may read behind its own (the id loads run six phases ahead)
// The chain itself, for all threads of the workgroup; the sum is wave 0's (lane = dimension).  xs: the state's first frame,
// state_bytes: its size (the descriptor's range).  LIST: row i of the chain is frame li[i] of the state (the frames of one
// cluster, kmeans_rowsum_kernel's lists; li readable SS_LIST_PAD entries behind the chain's n), else frame i.
template <bool LIST>
__device__ __forceinline__ double wide_row_sum(const double* __restrict__ xs, int D, int64_t state_bytes, const int32_t* __restrict__ li,
                                               int64_t n, double* __restrict__ ss_rows /*[2][SS_PHASE][64]*/) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (the descriptor in scalar registers for certain: a descriptor the compiler takes for per-lane data puts a
    //  readfirstlane loop around every load)
    auto uniform_ptr = [](const void* p) {
        const uint64_t a = (uint64_t)p;
        return ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a);
    };
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)uniform_ptr(xs), 0, __builtin_amdgcn_readfirstlane((int)state_bytes), 0x00020000);
    const unsigned row_b = (unsigned)D * 8u;
    const int64_t n_phase = (n + SS_PHASE - 1) / SS_PHASE;
    // LIST: the 16 frame numbers of a wave's share of a phase sit in lanes 0 .. 15 of one register per set, fetched SIX phases
    // ahead -- right behind the rows of three phases ahead in the (in-order) memory queue, so that waiting for them never
    // waits for younger rows -- and handed to the row loads as scalar offsets by v_readlane
    auto load_ids = [&](int& idv, int64_t q) { idv = li[(q * SS_WAVES + w) * SS_ROWS + (lane & 15)]; };
    // (the per-lane offset does the masking: lanes behind the last dimension -- and, without a list, phases behind the chain's
    //  end -- point out of the descriptor's range and read 0; with a list a row behind the end is some frame or 0, never added)
    auto issue = [&](double (&x)[SS_ROWS], int64_t q, int idv) {
        if (LIST) {
            const unsigned v0 = lane < D ? (unsigned)lane * 8u : 0x7ffffff0u;
#pragma unroll
            for (int j = 0; j < SS_ROWS; ++j)
                x[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, v0, (unsigned)__builtin_amdgcn_readlane(idv, j) * row_b, 0));
        } else {
            const bool in = lane < D && q < n_phase;
            const unsigned v0 = in ? (unsigned)((q * SS_WAVES + w) * SS_ROWS) * row_b + (unsigned)lane * 8u : 0x7ffffff0u;
            const unsigned step = in ? row_b : 0u;
#pragma unroll
            for (int j = 0; j < SS_ROWS; ++j)
                x[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, v0 + (unsigned)j * step, 0, 0));
        }
    };
    double acc = 0.0;
    // LDS image of a phase: [pair of rows][lane][2] -- a lane's values of two consecutive rows side by side, written and read
    // as 16 bytes (half the LDS instructions of the 8-byte form; the adds of wave 0 run 8 reads behind their data).
    // Measured: 0.87 -> 0.37 ms for continuous_train's states, ~24 cycles per row against the 8.3 of the dependent
    // v_add_f64 itself (tools/add_f64_latency.hip) -- what remains is the CU's load path, one 312-byte row per load
    using pair_t = double __attribute__((ext_vector_type(2)));
    pair_t* image = reinterpret_cast<pair_t*>(ss_rows);
    // wave 0 adds the rows of phase q from LDS buffer q & 1; a full phase with the LDS reads one batch ahead of the adds
    auto add_phase = [&](int64_t q) {
        const int64_t left = n - q * SS_PHASE;
        const pair_t* src = image + (size_t)(q & 1) * (SS_PHASE / 2) * 64 + lane;
        if (left >= SS_PHASE) {
            pair_t x0[8], x1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x0[j] = src[j * 64];
#pragma unroll
            for (int r = 0; r < SS_PHASE / 2; r += 16) {
#pragma unroll
                for (int j = 0; j < 8; ++j) x1[j] = src[(r + 8 + j) * 64];
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc += x0[j].x; acc += x0[j].y; }
                if (r + 16 < SS_PHASE / 2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) x0[j] = src[(r + 16 + j) * 64];
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc += x1[j].x; acc += x1[j].y; }
            }
        } else {
            for (int r = 0; r < (int)(left >> 1); ++r) { const pair_t v = src[r * 64]; acc += v.x; acc += v.y; }
            if (left > 0 && (left & 1)) acc += src[(left >> 1) * 64].x;      // (a phase behind the chain's end: left <= 0)
        }
    };
    // NO condition around a phase or a load (a `break` or an `if` around either costs an s_waitcnt vmcnt(0), i.e. the
    // read-ahead): the phases come in threes, one behind the chain's end adds nothing.  THREE sets of 16 rows in flight per
    // wave, not four: vmcnt counts to 63, and with 64 loads outstanding the compiler's bookkeeping gives up -- it then drains
    // the queue to 14 at the head of every round
    double a[SS_ROWS], b[SS_ROWS], c[SS_ROWS];
    pair_t* mine = image + (size_t)w * (SS_ROWS / 2) * 64 + lane;
    int ia = 0, ib = 0, ic = 0;
    auto phase = [&](double (&x)[SS_ROWS], int& idv, int64_t q) {
        pair_t* dst = mine + (size_t)(q & 1) * (SS_PHASE / 2) * 64;
#pragma unroll
        for (int j = 0; j < SS_ROWS; j += 2) { pair_t v; v.x = x[j]; v.y = x[j + 1]; dst[(j >> 1) * 64] = v; }
        issue(x, q + 3, idv);
        if (LIST) load_ids(idv, q + 6);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (w == 0) add_phase(q);
    };
    // (the sets' loads in program order, set after set: interleaved by the scheduler, the queue at the loop's head differs from
    //  the one at its back edge and the wait counts fall back to draining it)
    if (LIST) { load_ids(ia, 0); load_ids(ib, 1); load_ids(ic, 2); }
    __builtin_amdgcn_sched_barrier(0);
    issue(a, 0, ia);
    if (LIST) load_ids(ia, 3);
    __builtin_amdgcn_sched_barrier(0);
    issue(b, 1, ib);
    if (LIST) load_ids(ib, 4);
    __builtin_amdgcn_sched_barrier(0);
    issue(c, 2, ic);
    if (LIST) load_ids(ic, 5);
    __builtin_amdgcn_sched_barrier(0);
    for (int64_t p = 0; p < n_phase; p += 3) {
        phase(a, ia, p);
        phase(b, ib, p + 1);
        phase(c, ic, p + 2);
    }
    return acc;
}

__global__ __launch_bounds__(SS_WAVES * 64) void fit_segsum_wide_kernel(const double* __restrict__ X, int D, const int64_t* __restrict__ seg_off,
                                                                        double* __restrict__ out /*[S, D+1]: sums | count*/) {
    extern __shared__ double ss_rows[];                        // [2][SS_PHASE][64]
    const int s = blockIdx.x;
    const int64_t f0 = seg_off[s], n = seg_off[s + 1] - f0;
    if (threadIdx.x == 0) out[(int64_t)s * (D + 1) + D] = (double)n;
    const double acc = wide_row_sum<false>(X + f0 * D, D, n * D * 8, nullptr, n, ss_rows);
    if (threadIdx.x < D) out[(int64_t)s * (D + 1) + threadIdx.x] = acc;       // (threads 0 .. D-1: wave 0, D <= 64)
}

// kmeans_rowsum_kernel<0> over whole lists (chunk 0) in the same form: grid (S, k), the sum of every cluster's frames in
// frame order.  `lists` readable SS_LIST_PAD entries behind its last one (gh_fit_create pads it).
__global__ __launch_bounds__(SS_WAVES * 64) void kmeans_rowsum_wide_kernel(const double* __restrict__ X, int D, int k, const int64_t* __restrict__ seg_off,
                                                                           const uint8_t* __restrict__ active, const int32_t* __restrict__ lists,
                                                                           const int32_t* __restrict__ counts, const int32_t* __restrict__ cbase,
                                                                           double* __restrict__ sums, int sstride) {
    extern __shared__ double ss_rows[];
    const int s = blockIdx.x, c = blockIdx.y;
    if (active && !active[s]) return;
    const int64_t f0 = seg_off[s];
    const int n = counts[s * k + c];
    double* out = sums + (int64_t)s * sstride + (int64_t)c * (D + 1);
    if (threadIdx.x == 0) out[D] = (double)n;
    if (n <= 0) return;                                        // (as the one-wave kernel: the sums of an empty cluster stay)
    const double acc = wide_row_sum<true>(X + f0 * D, D, (seg_off[s + 1] - f0) * (int64_t)D * 8, lists + f0 + cbase[s * k + c], n, ss_rows);
    if (threadIdx.x < D) out[threadIdx.x] = acc;
}

// partition variances from (sum x | n) and sum (x - mean)^2: ddof = 1 (np.cov's default); n <= 1 -> NaN like numpy's
// 0 * (1 / 0).  one_pass: from global sums over all ranks, cluster 0 only, copied to every cluster (the sharded rule).
__global__ void fit_partvar_kernel(int S, int k, int D, int sstride, const double* __restrict__ sums /*[S][sstride]: [k][D+1]*/,
                                   const double* __restrict__ sq /*same layout*/, int one_pass, double* __restrict__ cov /*[S,k,D]*/) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * k * D) return;
    const int s = i / (k * D), c = (i / D) % k, d = i % D;
    const int cs = one_pass ? 0 : c;
    const double* su = sums + (int64_t)s * sstride + (int64_t)cs * (D + 1);
    const double n = su[D];
    const double q = sq[(int64_t)s * sstride + (int64_t)cs * (D + 1) + d];
    double v;
    if (one_pass) v = (q - su[d] * su[d] / n) / (n - 1.0);
    else v = n > 1.0 ? q * (1.0 / (n - 1.0)) : NAN;
    cov[i] = v;
}

// The partition variances of SMALL states once more, in two passes.  The streaming pass forms them from one-pass sums
// (count | sum y | sum y^2 around the state's shift point): accurate to ~1e-13 for groups of tens of frames, but a group of
// TWO frames that happen to lie close together in some dimension has a variance of 1e-7 .. 1e-9 of the spread, and the
// difference of squares keeps only 7-9 digits of it (1.4e-9 .. 5e-9 relative, tools/stress_refit.py).  States of up to
// `max_frames` frames -- the only ones with such groups -- get np.cov's two passes in frame order (what the tile path's
// kmeans_rowsum<0> / <1> + fit_partvar compute): grid (S, k), lane = dimension, D <= 64.
__global__ __launch_bounds__(64) void fit_partvar_small_kernel(const double* __restrict__ X, int D, int k, const int64_t* __restrict__ seg_off,
                                                               const int32_t* __restrict__ ids, int max_frames, double* __restrict__ cov /*[S,k,D]*/) {
    const int s = blockIdx.x, c = blockIdx.y, d = threadIdx.x;
    const int64_t f0 = seg_off[s];
    const int64_t n = seg_off[s + 1] - f0;
    if (n > max_frames || d >= D) return;
    double sum = 0.0;
    int cnt = 0;
    for (int64_t i = 0; i < n; ++i)
        if (ids[f0 + i] == c) { sum += X[(f0 + i) * D + d]; ++cnt; }
    double v = NAN;                                             // n <= 1: numpy's 0 * (1 / 0)
    if (cnt > 1) {
        const double ctr = sum / (double)cnt;
        double q = 0.0;
        for (int64_t i = 0; i < n; ++i)
            if (ids[f0 + i] == c) { const double t = X[(f0 + i) * D + d] - ctr; q += t * t; }
        v = q * (1.0 / (cnt - 1.0));
    }
    cov[((int64_t)s * k + c) * D + d] = v;
}

__global__ void fit_logdet_kernel(int S, int k, int D, const double* __restrict__ cov, double* __restrict__ logdet) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    double prod = 1.0;
    for (int d = 0; d < D; ++d) prod *= cov[(int64_t)s * k * D + d];
    logdet[s] = 0.5 * log(pow(2.0 * M_PI, (double)D) * prod);   // hmm_state.py:58
}

__global__ void fit_pack_changed_kernel(int S, int sstride, int at, const int32_t* __restrict__ changed, const uint8_t* __restrict__ active,
                                        double* __restrict__ sums) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < S) sums[(int64_t)s * sstride + at] = active[s] ? (double)changed[s] : 0.0;
}

// a stopped state contributes zeros to the all-reduced cluster sums (its frames are not re-summed)
__global__ void fit_zero_inactive_kernel(int S, int sstride, const uint8_t* __restrict__ active, double* __restrict__ buf) {
    const int s = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < sstride && !active[s]) buf[(int64_t)s * sstride + i] = 0.0;
}

// centroid update + stop rule of one state per block (kmeans.py:187-192; sharded: lockstep's "nobody's assignment changed")
__global__ __launch_bounds__(64) void fit_kmeans_update_kernel(int k, int D, int sstride, const double* __restrict__ sums,
                                                               int sharded /*1: all-reduced sums; 2: one rank, stop when no assignment changed*/,
                                                               double* __restrict__ cent, uint8_t* __restrict__ active,
                                                               int32_t* __restrict__ changed, int32_t* __restrict__ iters,
                                                               int* __restrict__ counter) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (!active[s]) return;
    const double* su = sums + (int64_t)s * sstride;
    double* ce = cent + (int64_t)s * k * D;
    bool same = true;
    for (int i = lane; i < k * D; i += 64) {
        const int c = i / D, d = i - c * D;
        const double nv = su[c * (D + 1) + d] / su[c * (D + 1) + D];    // an empty cluster: 0 / 0 = NaN, like np.mean of nothing
        if (!(nv == ce[i])) same = false;                              // np.array_equal: NaN never equals
        if (sharded) ce[i] = nv;
    }
    const bool all_same = __ballot(!same) == 0ull;
    const bool stop = sharded ? (su[k * (D + 1)] == 0.0) : all_same;
    if (!sharded && !stop)
        for (int i = lane; i < k * D; i += 64) { const int c = i / D, d = i - c * D; ce[i] = su[c * (D + 1) + d] / su[c * (D + 1) + D]; }
    if (lane == 0) {
        iters[s] += 1;
        changed[s] = 0;
        if (stop) active[s] = 0; else atomicAdd(counter, 1);
    }
}

// partial[s][c][z][D+1] (pieces of `chunk` list entries, in list order) -> sums[s][c][D+1] with the count in column D
__global__ void fit_chunk_reduce_kernel(int S, int k, int D, int nch, int chunk, const int32_t* __restrict__ counts,
                                        const double* __restrict__ partial, const uint8_t* __restrict__ active,
                                        double* __restrict__ sums, int sstride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * k * (D + 1)) return;
    const int s = i / (k * (D + 1)), c = (i / (D + 1)) % k, d = i % (D + 1);
    if (active && !active[s]) return;
    const int n = counts[s * k + c];
    double acc = 0.0;
    if (d == D) acc = (double)n;
    else {
        const int pieces = (n + chunk - 1) / chunk;
        const double* p = partial + ((int64_t)s * k + c) * nch * (D + 1) + d;
        for (int z = 0; z < pieces; ++z) acc += p[(int64_t)z * (D + 1)];
    }
    sums[(int64_t)s * sstride + (int64_t)c * (D + 1) + d] = acc;
}

// centroids = (sum of the cluster's frames in frame order) / count for every state: the values the reference returns
__global__ void fit_final_centroids_kernel(int S, int k, int D, int sstride, const double* __restrict__ sums, double* __restrict__ cent) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * k * D) return;
    const int s = i / (k * D), c = (i / D) % k, d = i % D;
    const double* su = sums + (int64_t)s * sstride + (int64_t)c * (D + 1);
    cent[i] = su[d] / su[D];
}

__global__ void fit_counts_kernel(int n, const int32_t* __restrict__ counts, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)counts[i];
}

__global__ void fit_em_prepare_kernel(int S, int k, int D, const double* __restrict__ var, const double* __restrict__ weight,
                                      const uint8_t* __restrict__ active, double* __restrict__ ivar, double* __restrict__ logc,
                                      int* __restrict__ counter) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= S * k || !active[g / k]) return;
    const double log2pi = 1.8378770664093454836;
    double sl = 0;
    for (int d = 0; d < D; ++d) {
        const double v = var[(int64_t)g * D + d];
        if (v == 0) atomicOr(counter + 1, 16);            // np.linalg.inv raises LinAlgError (hmm_state.py:17)
        ivar[(int64_t)g * D + d] = 1.0 / v;
        sl += log(v);
    }
    logc[g] = log(weight[g]) - 0.5 * (D * log2pi + sl);
}

__device__ __forceinline__ bool fit_close(double a, double b) {   // np.isclose(a, b), default tolerances
    if (a == b) return true;
    if (!(a - a == 0.0) || !(b - b == 0.0)) return false;
    return fabs(a - b) <= 1e-8 + 1e-5 * fabs(b);
}

// GMM.em_update (hmm_state.py:134-159) of one state per block: M-step from the centred statistics, parameters installed,
// then the allclose test against the previous iteration's; a state that passes stops, the others remember the new values
__global__ __launch_bounds__(256) void fit_em_update_kernel(int k, int D, int plen, const double* __restrict__ stats,
                                                            const double* __restrict__ nframes, double* __restrict__ mean,
                                                            double* __restrict__ var, double* __restrict__ weight,
                                                            double* __restrict__ old_mu, double* __restrict__ old_sigma,
                                                            double* __restrict__ old_w, uint8_t* __restrict__ active,
                                                            int32_t* __restrict__ conv_at, int it, int* __restrict__ counter) {
    const int s = blockIdx.x, tid = threadIdx.x;
    if (!active[s]) return;
    __shared__ int bad, moved;
    if (tid == 0) { bad = 0; moved = 0; }
    __syncthreads();
    const int Wd = 1 + 2 * D;
    const double* st = stats + (int64_t)s * plen;
    int mine = 0, diff = 0;
    unsigned onept = 0xffffffffu;                  // bit c: every entry of component c this thread holds says "one point"
    // `same`: identical to the previous iteration's value, NaN counting as equal to NaN
    auto same = [](double a, double b) { return a == b || (a != a && b != b); };
    for (int i = tid; i < k * D; i += blockDim.x) {
        const int c = i / D, d = i - c * D;
        const double s0 = st[c * Wd], S1 = st[c * Wd + 1 + d], S2 = st[c * Wd + 1 + D + d];
        // (a component with all its weight on ONE point -- see em_update_state, gh_refit_mfma.hip: the reference's variance is
        //  exactly 0 there and update_models raises; the centred sums leave rounding noise)
        {
            const double mq = S1 / s0, qq = S2 / s0;       // (weighted mean and mean square: products of the sums underflow for s0 ~ 1e-200)
            if (!(s0 >= 1e-290 && fabs(mq * mq - qq) <= 3.6e-15 * fabs(qq))) onept &= ~(1u << c);
        }
        const int64_t at = ((int64_t)s * k + c) * D + d;
        const double m0 = mean[at];
        const double occ = (s0 == 0) ? 1e-5 : s0;
        const double mu = (m0 * s0 + S1) / occ;
        const double dl = mu - m0;
        const double sg = (S2 - dl * (2.0 * S1 - dl * s0)) / occ;
        diff += !same(mu, m0) + !same(sg, var[at]);
        if (sg <= 0) atomicOr(counter + 1, 16);              // update_models raises before the convergence test (hmm_state.py:149); < 0: below the sums' rounding noise
        mean[at] = mu;
        var[at] = sg;
        mine += !fit_close(mu, old_mu[at]) + !fit_close(sg, old_sigma[at]);
    }
    if (tid < k) {
        const double w = st[tid * Wd] / nframes[s];
        diff += !same(w, weight[(int64_t)s * k + tid]);
        weight[(int64_t)s * k + tid] = w;
        mine += !fit_close(w, old_w[(int64_t)s * k + tid]);
    }
    if (mine) atomicAdd(&bad, mine);
    if (diff) atomicAdd(&moved, diff);
    for (int c = 0; c < k; ++c)
        if (__syncthreads_and((int)((onept >> c) & 1u)) && tid == 0) atomicOr(counter + 1, 16);
    __syncthreads();
    if (bad == 0) {
        if (tid == 0) { active[s] = 0; conv_at[s] = it; }
        return;
    }
    for (int i = tid; i < k * D; i += blockDim.x) {
        const int64_t at = (int64_t)s * k * D + i;
        old_mu[at] = mean[at];
        old_sigma[at] = var[at];
    }
    if (tid < k) old_w[(int64_t)s * k + tid] = weight[(int64_t)s * k + tid];
    // A state whose parameters came out bit for bit as they went in (NaN = NaN) sits at a fixed point the allclose
    // test cannot see -- an empty cluster's NaN mean poisons the whole mixture after one iteration, and the reference
    // then spins to max_iteration on NaN == NaN being False.  Every further iteration would reproduce the same bits:
    // the state stops here with the values the full loop ends with (converged_at stays -1).
    if (moved == 0) {
        if (tid == 0) active[s] = 0;
        return;
    }
    if (tid == 0) atomicAdd(counter, 1);
}

}  // namespace

extern "C" void gh_fit_destroy(gh_fit* f) {
    if (!f) return;
    hipSetDevice(f->ctx->device);
    hipStreamSynchronize(f->ctx->stream);
    // arena and page-locked blocks: kept in the context for the next session (the larger of the two arenas survives)
    gh_ctx* c = f->ctx;
    if (f->d_arena) {
        static const size_t keep = [] { const char* e = getenv("GMMHMM_FIT_KEEP_MB"); return (size_t)(e ? atoll(e) : 4096) << 20; }();
        if (f->arena_bytes <= keep && f->arena_bytes > c->fit_arena_bytes) {
            if (c->fit_arena) hipFree(c->fit_arena);
            c->fit_arena = f->d_arena;
            c->fit_arena_bytes = f->arena_bytes;
        } else {
            hipFree(f->d_arena);
        }
    }
    if (f->h_pin) { if (!c->fit_pin) c->fit_pin = f->h_pin; else hipHostFree(f->h_pin); }
    if (f->h_act) {
        if (f->act_bytes > c->fit_act_bytes) {
            if (c->fit_act) hipHostFree(c->fit_act);
            c->fit_act = f->h_act;
            c->fit_act_bytes = f->act_bytes;
        } else {
            hipHostFree(f->h_act);
        }
    }
    f->h_pin = nullptr; f->h_act = nullptr;
    if (getenv("GMMHMM_REFIT_DEBUG"))
        fprintf(stderr, "[gh_fit] %d states, %d items: %ld tail launches (%ld iterations), %ld refused, %ld ordinary launches\n", f->S, f->n_items,
                f->n_tail_launches, f->n_tail_iters, f->n_tail_refused, f->n_plain_launches);
    delete f->h_iptr;
    delete f->h_tail_ids;
    delete f->h_tail_mask;
    delete f;
}

extern "C" int gh_fit_create(gh_ctx* ctx, const gh_batch* b, int S, const int64_t* seg_off, int kmax, gh_fit** out) {
    GH_REQUIRE(ctx && b && out, "gh_fit_create: NULL argument");
    *out = nullptr;
    int rc = check_segments(b, S, seg_off, "gh_fit_create");
    if (rc) return rc;
    GH_REQUIRE(seg_off[0] == 0 && seg_off[S] == b->N, "gh_fit_create: the segments must cover the batch");
    if (kmax < 1 || kmax > 32 || b->D < 2) {
        gh_set_error("gh_fit_create: kmax=%d (1..32), D=%d (2..%d) outside the device-resident refit", kmax, b->D, LS_MAXD);
        return GH_ERR_UNSUPPORTED;
    }
    for (int s = 0; s < S; ++s)
        if ((seg_off[s + 1] - seg_off[s]) * (int64_t)b->D * 8 >= ((int64_t)1 << 31)) {
            gh_set_error("gh_fit_create: state %d holds %lld frames (a segment must stay below 2 GiB)", s, (long long)(seg_off[s + 1] - seg_off[s]));
            return GH_ERR_UNSUPPORTED;
        }
    GH_HIP(hipSetDevice(ctx->device));
    const int D = b->D, Wd = 1 + 2 * D;
    std::vector<ls_tile> tiles;
    std::vector<int32_t> tile_ptr;
    build_tiles(S, seg_off, nullptr, tiles, tile_ptr);
    gh_fit* f = new gh_fit();
    memset((void*)f, 0, sizeof *f);
    f->ctx = ctx; f->b = b; f->S = S; f->D = D; f->kmax = kmax; f->N = b->N; f->n_tiles = (int)tiles.size();
    const size_t nt = std::max<size_t>(1, tiles.size()), N1 = std::max<int64_t>(1, b->N);
    const size_t skd = (size_t)S * kmax * D, sstride = (size_t)kmax * (D + 1) + 1, plen = (size_t)kmax * Wd + 1;
    // LDS of the two tile kernels at kmax
    const size_t lds_km = ((size_t)kmax * D + D + 64 * (size_t)(D + 1)) * 8 + 64 * 4 + 16;
    const size_t lds_em = ((size_t)2 * kmax * D + kmax + (size_t)kmax * 64 + 64 * (size_t)(D + 1)) * 8 + 16;
    if (lds_km > 150 * 1024 || lds_em > 150 * 1024) {
        delete f;
        gh_set_error("gh_fit_create: k=%d x D=%d does not fit LDS", kmax, D);
        return GH_ERR_UNSUPPORTED;
    }
    UploadLayout lay;
    lay.add((void**)&f->d_tiles, nt * sizeof(ls_tile), tiles.data(), tiles.size() * sizeof(ls_tile));
    lay.add((void**)&f->d_tptr, (size_t)(S + 1) * 4, tile_ptr.data(), (size_t)(S + 1) * 4);
    lay.add((void**)&f->d_segoff, (size_t)(S + 1) * 8, seg_off, (size_t)(S + 1) * 8);
    lay.add((void**)&f->d_active, (size_t)S, nullptr);
    lay.add((void**)&f->d_ids, N1 * 4, nullptr);
    lay.add((void**)&f->d_tilecnt, nt * kmax * 4, nullptr);
    lay.add((void**)&f->d_counts, (size_t)S * kmax * 4, nullptr);
    lay.add((void**)&f->d_cbase, (size_t)S * kmax * 4, nullptr);
    lay.add((void**)&f->d_lists, (N1 + SS_LIST_PAD) * 4, nullptr);      // (kmeans_rowsum_wide_kernel reads ids ahead of a list's end)
    lay.add((void**)&f->d_changed, (size_t)S * 4, nullptr);
    lay.add((void**)&f->d_iters, (size_t)S * 4, nullptr);
    lay.add((void**)&f->d_cent, skd * 8, nullptr);
    lay.add((void**)&f->d_cov, skd * 8, nullptr);
    lay.add((void**)&f->d_logdet, (size_t)S * 8, nullptr);
    lay.add((void**)&f->d_sums, (size_t)S * sstride * 8, nullptr);
    lay.add((void**)&f->d_sq, (size_t)S * sstride * 8, nullptr);
    {
        int64_t longest = 1;
        for (int s = 0; s < S; ++s) longest = std::max(longest, seg_off[s + 1] - seg_off[s]);
        f->nch = (int)((longest + FIT_CHUNK - 1) / FIT_CHUNK);
    }
    lay.add((void**)&f->d_psum, (size_t)S * kmax * f->nch * (D + 1) * 8, nullptr);
    lay.add((void**)&f->d_counter, 64, nullptr);
    lay.add((void**)&f->d_mean, skd * 8, nullptr);
    lay.add((void**)&f->d_var, skd * 8, nullptr);
    lay.add((void**)&f->d_weight, (size_t)S * kmax * 8, nullptr);
    lay.add((void**)&f->d_ivar, skd * 8, nullptr);
    lay.add((void**)&f->d_logc, (size_t)S * kmax * 8, nullptr);
    lay.add((void**)&f->d_old_mu, skd * 8, nullptr);
    lay.add((void**)&f->d_old_sigma, skd * 8, nullptr);
    lay.add((void**)&f->d_old_w, (size_t)S * kmax * 8, nullptr);
    lay.add((void**)&f->d_nframes, (size_t)S * 8, nullptr);
    lay.add((void**)&f->d_part, nt * plen * 8, nullptr);
    lay.add((void**)&f->d_stats, (size_t)S * plen * 8, nullptr);
    // items of the streaming kernels: a state's frames cut into pieces of item_frames (a state without frames: one empty
    // item, whose tail still runs the state's update)
    std::vector<rf_item> items;
    std::vector<int32_t> iptr(S + 1, 0);
    {
        const char* e = getenv("GMMHMM_REFIT");
        f->mfma = !(e && !strcmp(e, "tiles"));
        int item_frames = 512;
        if (const char* ei = getenv("GMMHMM_REFIT_ITEM")) item_frames = std::max(64, atoi(ei) & ~15);
        f->kcap = std::min(kmax, 8);
        for (int s = 0; s < S; ++s) {
            iptr[s] = (int32_t)items.size();
            const int64_t n = seg_off[s + 1] - seg_off[s];
            if (n == 0) items.push_back(rf_item{seg_off[s], s, 0});
            for (int64_t o = 0; o < n; o += item_frames)
                items.push_back(rf_item{seg_off[s] + o, s, (int32_t)std::min<int64_t>(item_frames, n - o)});
        }
        iptr[S] = (int32_t)items.size();
        f->n_items = (int)items.size();
        f->h_iptr = new std::vector<int32_t>(iptr);
        f->h_tail_ids = new std::vector<int32_t>();
        f->h_tail_mask = new std::vector<uint8_t>();
        f->tail_n = 0;
        {
            const char* et = getenv("GMMHMM_REFIT_TAIL");
            f->tail = !(et && *et == '0');
        }
        if (f->mfma) {
            const size_t pst = std::max(rf_em_pstride(f->kcap, D), rf_km_pstride(f->kcap, D));
            lay.add((void**)&f->d_items, items.size() * sizeof(rf_item), items.data(), items.size() * sizeof(rf_item));
            lay.add((void**)&f->d_iptr, (size_t)(S + 1) * 4, iptr.data(), (size_t)(S + 1) * 4);
            lay.add((void**)&f->d_done, (size_t)S * 4, nullptr);
            lay.add((void**)&f->d_gen, (size_t)S * 4, nullptr);
            lay.add((void**)&f->d_tail_ids, items.size() * 4, nullptr);
            lay.add((void**)&f->d_P, (size_t)S * pst * 8, nullptr);
            lay.add((void**)&f->d_shift, (size_t)S * D * 8, nullptr);
            lay.add((void**)&f->d_kscale, (size_t)S * 8, nullptr);
            lay.add((void**)&f->d_rpart, items.size() * ((size_t)f->kcap * Wd + 1) * 8, nullptr);
        }
    }
    // the arena and the page-locked blocks of the session closed last on this context, when they are large enough (a
    // recycled arena is cleared: what a session finds is what a fresh allocation holds)
    hipError_t he = hipSuccess;
    f->arena_bytes = lay.total;
    f->act_bytes = (size_t)std::max(S, 1);
    if (ctx->fit_arena && ctx->fit_arena_bytes >= lay.total) {
        f->d_arena = ctx->fit_arena;
        f->arena_bytes = ctx->fit_arena_bytes;
        ctx->fit_arena = nullptr;
        ctx->fit_arena_bytes = 0;
        he = hipMemsetAsync(f->d_arena, 0, lay.total, ctx->stream);
    } else {
        he = hipMalloc(&f->d_arena, lay.total);
    }
    if (he == hipSuccess) {
        if (ctx->fit_pin) { f->h_pin = (decltype(f->h_pin))ctx->fit_pin; ctx->fit_pin = nullptr; }
        else he = hipHostMalloc((void**)&f->h_pin, 64, hipHostMallocDefault);
    }
    if (he == hipSuccess) {
        if (ctx->fit_act && ctx->fit_act_bytes >= f->act_bytes) {
            f->h_act = (uint8_t*)ctx->fit_act;
            f->act_bytes = ctx->fit_act_bytes;
            ctx->fit_act = nullptr;
            ctx->fit_act_bytes = 0;
        } else {
            he = hipHostMalloc((void**)&f->h_act, f->act_bytes, hipHostMallocDefault);
        }
    }
    if (he != hipSuccess) {
        gh_set_error("gh_fit_create: %s", hipGetErrorString(he));
        gh_fit_destroy(f);
        return he == hipErrorOutOfMemory ? GH_ERR_NOMEM : GH_ERR_HIP;
    }
    rc = lay.commit(f->d_arena, ctx->stream, true);
    if (!rc && hipMemsetAsync(f->d_counter, 0, 64, ctx->stream) != hipSuccess) rc = GH_ERR_HIP;
    if (!rc && hipMemsetAsync(f->d_changed, 0, (size_t)S * 4, ctx->stream) != hipSuccess) rc = GH_ERR_HIP;
    if (!rc && f->mfma && hipMemsetAsync(f->d_done, 0, (size_t)S * 4, ctx->stream) != hipSuccess) rc = GH_ERR_HIP;
    if (rc) { gh_fit_destroy(f); return rc; }
    *out = f;
    return GH_OK;
}

extern "C" int gh_fit_segment_means(gh_ctx* ctx, gh_fit* f, double* out /*[S, D+1]: sums | count*/) {
    GH_REQUIRE(ctx && f && out && f->ctx == ctx, "gh_fit_segment_means: NULL argument / foreign context");
    GH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    static const bool narrow = [] { const char* e = getenv("GMMHMM_SEGSUM"); return e && !strcmp(e, "narrow"); }();
    if (f->D <= 64 && !narrow)      // (a state stays below 2 GiB: gh_fit_create)
        hipLaunchKernelGGL(fit_segsum_wide_kernel, dim3((unsigned)f->S), dim3(SS_WAVES * 64), (size_t)2 * SS_PHASE * 64 * 8, st,
                           (const double*)f->b->feats, f->D, f->d_segoff, f->d_sums);
    else
        hipLaunchKernelGGL(fit_segsum_kernel, dim3((unsigned)f->S, (unsigned)((f->D + 63) / 64)), dim3(64), 0, st, (const double*)f->b->feats,
                           f->D, f->d_segoff, f->d_sums);
    GH_HIP(hipGetLastError());
    GH_HIP(hipMemcpyAsync(out, f->d_sums, (size_t)f->S * (f->D + 1) * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}

// lists of every (state, cluster)'s frames from f->d_ids, for the states that are active (all: active == nullptr)
static int fit_build_lists(gh_fit* f, int k, bool from_assign_counts, const uint8_t* active) {
    hipStream_t st = f->ctx->stream;
    if (f->n_tiles == 0) return GH_OK;
    const dim3 grid((unsigned)f->n_tiles), blk(64);
    if (!from_assign_counts) hipLaunchKernelGGL(fit_tilecount_kernel, grid, blk, 0, st, f->d_tiles, k, f->d_ids, f->d_tilecnt);
    hipLaunchKernelGGL(kmeans_scan_kernel, dim3((unsigned)f->S), dim3(256), 0, st, f->d_tptr, k, f->d_tilecnt, f->d_counts, f->d_cbase, active);
    hipLaunchKernelGGL(kmeans_scatter_kernel, grid, blk, 0, st, f->d_tiles, k, f->d_ids, f->d_segoff, f->d_tilecnt, f->d_cbase, f->d_lists, active);
    GH_HIP(hipGetLastError());
    return GH_OK;
}

// slot: where the last iteration left the number of states still active (0: the tile kernels' counter; 2 + (it & 7):
// the streaming kernels', which count per iteration so that nothing has to be cleared between launches)
static int fit_poll(gh_fit* f, gh_comm* comm, int* n_active, int* flags, int slot = 0, bool want_mask = false) {
    hipStream_t st = f->ctx->stream;
    GH_HIP(hipMemcpyAsync(f->h_pin, f->d_counter, 64, hipMemcpyDeviceToHost, st));
    if (want_mask) GH_HIP(hipMemcpyAsync(f->h_act, f->d_active, (size_t)f->S, hipMemcpyDeviceToHost, st));
    const int rc_ = gh_stream_wait(f->ctx, comm, "gh_fit");
    if (rc_) return rc_;
    *n_active = f->h_pin[slot];
    if (want_mask) {                   // (a tail launch runs more iterations than there are slots: count the mask itself)
        int n = 0;
        for (int s = 0; s < f->S; ++s) n += f->h_act[s] != 0;
        *n_active = n;
    }
    *flags = f->h_pin[1];
    return GH_OK;
}

// TAIL launches: the items of the states that are still active (as of the last poll), when they are few enough for all of
// their workgroups to be resident at once -- the workgroups then run the next iterations inside ONE launch, handing over
// through the state's generation word (gh_refit_mfma.hip).  Returns the number of items written to d_tail_ids (0: none).
static int fit_tail_items(gh_fit* f, int limit) {
    const std::vector<int32_t>& ip = *f->h_iptr;
    std::vector<uint8_t>& had = *f->h_tail_mask;
    bool same = f->tail_n > 0 && (int)had.size() == f->S;
    for (int s = 0; same && s < f->S; ++s) same = (had[s] != 0) == (f->h_act[s] != 0);
    if (!same) {                       // (the same states as for the last tail launch: the list on the device still holds)
        std::vector<int32_t>& ids = *f->h_tail_ids;
        ids.clear();
        for (int s = 0; s < f->S; ++s)
            if (f->h_act[s]) {
                if ((int)ids.size() + (ip[s + 1] - ip[s]) > limit) { f->tail_n = 0; return 0; }
                for (int t = ip[s]; t < ip[s + 1]; ++t) ids.push_back(t);
            }
        f->tail_n = 0;
        if (ids.empty()) return 0;
        // (`ids` lives in the session: the copy needs no wait; a later rebuild only happens behind a poll's synchronisation)
        if (hipMemcpyAsync(f->d_tail_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, f->ctx->stream) != hipSuccess) return 0;
        had.assign(f->h_act, f->h_act + f->S);
        f->tail_n = (int)ids.size();
    }
    if (hipMemsetAsync(f->d_gen, 0, (size_t)f->S * 4, f->ctx->stream) != hipSuccess) return 0;
    return f->tail_n;
}

// the point a state's frames and parameters are taken relative to by the streaming kernels: the average of the state's
// (finite) component means -- any point near the data serves; it only has to be the same on every rank
static void fit_shift_points(int S, int k, int D, const double* mean, std::vector<double>& shift) {
    shift.assign((size_t)S * D, 0.0);
    for (int s = 0; s < S; ++s)
        for (int d = 0; d < D; ++d) {
            double acc = 0;
            int n = 0;
            for (int c = 0; c < k; ++c) {
                const double v = mean[((size_t)s * k + c) * D + d];
                if (v - v == 0.0) { acc += v; ++n; }
            }
            shift[(size_t)s * D + d] = n ? acc / n : 0.0;
        }
}

extern "C" int gh_fit_kmeans(gh_ctx* ctx, gh_fit* f, gh_comm* comm, int k, const double* centroids_in, const uint8_t* part,
                             int max_iteration, int check_every, double* out_centroids, double* out_cov, double* out_counts,
                             int32_t* out_iters) {
    GH_REQUIRE(ctx && f && centroids_in && part && f->ctx == ctx, "gh_fit_kmeans: NULL argument / foreign context");
    GH_REQUIRE(k >= 1 && k <= f->kmax, "gh_fit_kmeans: k=%d (1..%d)", k, f->kmax);
    GH_REQUIRE(!comm || gh_comm_context(comm) == ctx, "gh_fit_kmeans: the communicator belongs to another context");
    GH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int S = f->S, D = f->D;
    const int sstride = k * (D + 1) + 1;
    const double* X = (const double*)f->b->feats;
    const dim3 skz((unsigned)S, (unsigned)k, (unsigned)((D + 63) / 64));
    int rc;
    const bool exact_env = [] { const char* e = getenv("GMMHMM_KMEANS_EXACT"); return e && *e && *e != '0'; }();
    // Streaming matrix-core form (gh_refit_mfma.hip): see the loop below
    const bool streaming = f->mfma && !exact_env && k <= f->kcap && rf_supported(k, D);
    // ---- the random partition and its variances (kmeans.py:171-177) ----
    {
        void* stage;
        rc = gh_scratch(ctx, (size_t)std::max<int64_t>(1, f->N), &stage);
        if (rc) return rc;
        if (f->N > 0) {
            GH_HIP(hipMemcpyAsync(stage, part, (size_t)f->N, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL(fit_u8_to_i32_kernel, dim3((unsigned)((f->N + 255) / 256)), dim3(256), 0, st, (const uint8_t*)stage, f->N, f->d_ids);
        }
        if (streaming) {
            // ONE streaming pass: count | sum (x - shift) | sum (x - shift)^2 of every (state, group) on the matrix cores, the
            // variances from those (the shift sits inside the data, so the one-pass form loses nothing; round 3-4: cluster
            // lists + two frame-order passes, 0.6 ms per call on 1.4 M frames).  Sharded: cluster 0's sums over all ranks.
            std::vector<double> shift;
            fit_shift_points(S, k, D, centroids_in, shift);
            GH_HIP(hipMemcpyAsync(f->d_shift, shift.data(), shift.size() * 8, hipMemcpyHostToDevice, st));
            GH_HIP(hipMemsetAsync(f->d_active, 1, (size_t)S, st));
            GH_HIP(hipMemsetAsync(f->d_counter, 0, 64, st));
            GH_HIP(hipMemsetAsync(f->d_done, 0, (size_t)S * 4, st));
            rf_em_args h;
            memset(&h, 0, sizeof h);
            h.c.X = X; h.c.D = D; h.c.k = k; h.c.S = S; h.c.items = f->d_items; h.c.item_ptr = f->d_iptr; h.c.shift = f->d_shift;
            h.c.active = f->d_active; h.c.done = f->d_done; h.c.partial = f->d_rpart; h.c.counter = f->d_counter;
            h.stats = f->d_stats; h.hard_ids = f->d_ids;
            rc = rf_launch_partition_sums(ctx, h, f->n_items);
            if (!rc && comm) rc = gh_comm_allreduce_enqueue(comm, f->d_stats, (int64_t)S * (k * (1 + 2 * D) + 1));
            if (!rc) rc = rf_launch_partvar(ctx, S, k, D, f->d_stats, comm ? 1 : 0, f->d_cov);
            if (rc) return rc;
            if (!comm && D <= 64) {     // (sharded: the variances come from sums over all ranks -- one pass by construction)
                hipLaunchKernelGGL(fit_partvar_small_kernel, dim3((unsigned)S, (unsigned)k), dim3(64), 0, st, X, D, k, f->d_segoff, (const int32_t*)f->d_ids,
                                   64, f->d_cov);
                GH_HIP(hipGetLastError());
            }
        } else {
        GH_HIP(hipMemsetAsync(f->d_counts, 0, (size_t)S * k * 4, st));
        GH_HIP(hipMemsetAsync(f->d_cbase, 0, (size_t)S * k * 4, st));
        rc = fit_build_lists(f, k, false, nullptr);
        if (rc) return rc;
        GH_HIP(hipMemsetAsync(f->d_sums, 0, (size_t)S * sstride * 8, st));
        GH_HIP(hipMemsetAsync(f->d_sq, 0, (size_t)S * sstride * 8, st));
        hipLaunchKernelGGL(kmeans_rowsum_kernel<0>, skz, dim3(64), 0, st, X, D, k, f->d_segoff, (const uint8_t*)nullptr, f->d_lists, f->d_counts,
                           f->d_cbase, f->d_sums, sstride, (const double*)nullptr, 0);
        if (!comm) {
            hipLaunchKernelGGL(kmeans_rowsum_kernel<1>, skz, dim3(64), 0, st, X, D, k, f->d_segoff, (const uint8_t*)nullptr, f->d_lists,
                               f->d_counts, f->d_cbase, f->d_sq, sstride, (const double*)f->d_sums, sstride);
        } else {   // sharded: (n, sum x, sum x^2) of cluster 0 over all ranks, one-pass variance (lockstep.py)
            hipLaunchKernelGGL(kmeans_rowsum_kernel<2>, skz, dim3(64), 0, st, X, D, k, f->d_segoff, (const uint8_t*)nullptr, f->d_lists,
                               f->d_counts, f->d_cbase, f->d_sq, sstride, (const double*)nullptr, 0);
            rc = gh_comm_allreduce_enqueue(comm, f->d_sums, (int64_t)S * sstride);
            if (!rc) rc = gh_comm_allreduce_enqueue(comm, f->d_sq, (int64_t)S * sstride);
            if (rc) return rc;
        }
        GH_HIP(hipGetLastError());
        hipLaunchKernelGGL(fit_partvar_kernel, dim3((unsigned)((S * k * D + 255) / 256)), dim3(256), 0, st, S, k, D, sstride,
                           (const double*)f->d_sums, (const double*)f->d_sq, comm ? 1 : 0, f->d_cov);
        }
        hipLaunchKernelGGL(fit_logdet_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, S, k, D, (const double*)f->d_cov, f->d_logdet);
        GH_HIP(hipGetLastError());
    }
    // ---- lock-step k-means ----
    // The reference recomputes every centroid as np.mean(data[clusters == c]) -- the cluster's frames added in frame
    // order, a serial chain per (state, cluster) that no amount of hardware shortens (207 us per iteration for 14 000
    // frame clusters: a wave has at most 64 loads in flight) -- and stops when the centroids repeat bit for bit.
    // Default here: the iterations use sums over 512-frame pieces of the cluster lists (a fixed order, 15 us) and stop
    // when no assignment of the state changed -- the same iteration, since equal assignments give equal sums -- and ONE
    // frame-order pass over the final assignment produces the centroids, which are then numpy's bit for bit.  What can
    // differ: a frame whose two nearest centroids tie to within an ulp might be assigned differently in an intermediate
    // iteration.  GMMHMM_KMEANS_EXACT=1 keeps the frame-order sums and the np.array_equal rule in every iteration.
    const bool exact_order = exact_env && !comm;
    // the tile kernels (k > 8, GMMHMM_REFIT=tiles) take the reference's division in the distance; its reciprocal form
    // (25 fewer instructions per term, assignments that can differ on ties at the last bit) is opt-in: GMMHMM_KMEANS_RECIP=1
    const bool exact_div = exact_env || ![] { const char* e = getenv("GMMHMM_KMEANS_RECIP"); return e && *e && *e != '0'; }();
    GH_HIP(hipMemcpyAsync(f->d_cent, centroids_in, (size_t)S * k * D * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemsetAsync(f->d_active, 1, (size_t)S, st));
    GH_HIP(hipMemsetAsync(f->d_iters, 0, (size_t)S * 4, st));
    GH_HIP(hipMemsetAsync(f->d_changed, 0, (size_t)S * 4, st));
    if (f->N > 0) GH_HIP(hipMemsetAsync(f->d_ids, 0xFF, (size_t)f->N * 4, st));
    const size_t lds = ((size_t)k * D + D + 64 * (size_t)(D + 1)) * 8 + 64 * 4 + 16;
    const int n_it = std::max(max_iteration, 1);
    check_every = std::max(1, check_every);
    // Streaming matrix-core form (gh_refit_mfma.hip): assignment, cluster sums, centroid update, stop rule and the next
    // iteration's operands in ONE launch per iteration (sums, collective, update with a communicator).  The assignments are
    // the reference's: near-ties are re-tested with its own division (so GMMHMM_KMEANS_EXACT only adds the frame-order
    // sums in every iteration, which the tile kernels below still provide).
    if (streaming) {
        // (the shift points are already in place: the partition pass above uses the same ones)
        GH_HIP(hipMemsetAsync(f->d_counter, 0, 64, st));
        GH_HIP(hipMemsetAsync(f->d_done, 0, (size_t)S * 4, st));
        rf_km_args a;
        memset(&a, 0, sizeof a);
        a.c.X = X; a.c.D = D; a.c.k = k; a.c.S = S; a.c.items = f->d_items; a.c.item_ptr = f->d_iptr; a.c.shift = f->d_shift;
        a.c.active = f->d_active; a.c.done = f->d_done; a.c.partial = f->d_rpart; a.c.counter = f->d_counter; a.c.fused = comm ? 0 : 1;
        a.P = f->d_P; a.kscale = f->d_kscale; a.cent = f->d_cent; a.var = f->d_cov; a.logdet = f->d_logdet; a.ids = f->d_ids;
        a.sums = f->d_sums; a.iters = f->d_iters;
        rc = rf_launch_km_update(ctx, a, 1);
        if (rc) return rc;
        const bool tail_ok = f->tail && !comm;
        int tail_n = 0;                     // > 0: the next block of iterations is ONE launch over these items
        f->tail_n = 0;
        for (int it = 0; it < n_it;) {
            int blkn = std::min(check_every, n_it - it);
            GH_HIP(hipMemsetAsync(f->d_counter + 2, 0, 32, st));      // the per-iteration slots of this block
            bool tailed = false;
            if (tail_n > 0) {
                const int blk_tail = std::min(FIT_TAIL_ITERS, n_it - it);
                blkn = blk_tail;
                a.c.it = it; a.c.item_ids = f->d_tail_ids; a.c.gen = f->d_gen; a.c.n_iter = blkn;
                rc = rf_launch_km(ctx, a, tail_n);
                if (rc < 0) return rc;
                tailed = rc == 0;
                if (tailed) { ++f->n_tail_launches; f->n_tail_iters += blkn; } else { ++f->n_tail_refused; blkn = std::min(check_every, n_it - it); }
                a.c.item_ids = nullptr; a.c.gen = nullptr; a.c.n_iter = 1;
            }
            for (int j = 0; !tailed && j < blkn; ++j) {
                a.c.it = it + j;
                rc = rf_launch_km(ctx, a, f->n_items);
                if (rc) return rc;
                ++f->n_plain_launches;
                if (comm) {
                    hipLaunchKernelGGL(fit_zero_inactive_kernel, dim3((unsigned)((sstride + 127) / 128), (unsigned)S), dim3(128), 0, st, S, sstride,
                                       (const uint8_t*)f->d_active, f->d_sums);
                    rc = gh_comm_allreduce_enqueue(comm, f->d_sums, (int64_t)S * sstride);
                    if (!rc) rc = rf_launch_km_update(ctx, a, 0);
                    if (rc) return rc;
                }
            }
            it += blkn;
            int n_active = 0, flags = 0;
            rc = fit_poll(f, comm, &n_active, &flags, 2 + ((it - 1) & 7), tail_ok);
            if (rc) return rc;
            if (flags & 32) { gh_set_error("gh_fit_kmeans: a workgroup of a tail launch waited for its state in vain"); return GH_ERR_HIP; }
            if (n_active == 0) break;
            tail_n = tail_ok ? fit_tail_items(f, 4096) : 0;
        }
        GH_HIP(hipMemsetAsync(f->d_counts, 0, (size_t)S * k * 4, st));
        GH_HIP(hipMemsetAsync(f->d_cbase, 0, (size_t)S * k * 4, st));
        rc = fit_build_lists(f, k, false, nullptr);       // the final assignment's cluster lists and sizes
        if (rc) return rc;
    } else
    for (int it = 0; it < n_it;) {
        const int blkn = std::min(check_every, n_it - it);
        for (int j = 0; j < blkn; ++j) {
            if (f->n_tiles > 0) {
                const dim3 grid((unsigned)f->n_tiles), blk(128);
#define GH_KM(DR, RC) hipLaunchKernelGGL((kmeans_multi_kernel<DR, RC>), grid, blk, lds, st, X, D, k, f->d_tiles, (const double*)f->d_cent, \
                                     (const double*)f->d_cov, (const double*)f->d_logdet, f->d_ids, f->d_changed, f->d_tilecnt, (const uint8_t*)f->d_active, k * D)
                if (exact_div) { if (D <= 16) GH_KM(16, false); else if (D <= 40) GH_KM(40, false); else GH_KM(64, false); }
                else { if (D <= 16) GH_KM(16, true); else if (D <= 40) GH_KM(40, true); else GH_KM(64, true); }
#undef GH_KM
            }
            rc = fit_build_lists(f, k, true, f->d_active);
            if (rc) return rc;
            if (exact_order) {
                hipLaunchKernelGGL(kmeans_rowsum_kernel<0>, skz, dim3(64), 0, st, X, D, k, f->d_segoff, (const uint8_t*)f->d_active, f->d_lists,
                                   f->d_counts, f->d_cbase, f->d_sums, sstride, (const double*)nullptr, 0, 0, 0);
            } else {
                hipLaunchKernelGGL(kmeans_rowsum_kernel<0>, dim3((unsigned)S, (unsigned)k, (unsigned)f->nch), dim3(64), 0, st, X, D, k, f->d_segoff,
                                   (const uint8_t*)f->d_active, f->d_lists, f->d_counts, f->d_cbase, f->d_psum, 0, (const double*)nullptr, 0,
                                   FIT_CHUNK, f->nch);
                hipLaunchKernelGGL(fit_chunk_reduce_kernel, dim3((unsigned)((S * k * (D + 1) + 255) / 256)), dim3(256), 0, st, S, k, D, f->nch, FIT_CHUNK,
                                   (const int32_t*)f->d_counts, (const double*)f->d_psum, (const uint8_t*)f->d_active, f->d_sums, sstride);
            }
            hipLaunchKernelGGL(fit_pack_changed_kernel, dim3((unsigned)((S + 63) / 64)), dim3(64), 0, st, S, sstride, k * (D + 1),
                               (const int32_t*)f->d_changed, (const uint8_t*)f->d_active, f->d_sums);
            if (comm) {
                hipLaunchKernelGGL(fit_zero_inactive_kernel, dim3((unsigned)((sstride + 127) / 128), (unsigned)S), dim3(128), 0, st, S, sstride,
                                   (const uint8_t*)f->d_active, f->d_sums);
                rc = gh_comm_allreduce_enqueue(comm, f->d_sums, (int64_t)S * sstride);
                if (rc) return rc;
            }
            GH_HIP(hipMemsetAsync(f->d_counter, 0, 4, st));
            hipLaunchKernelGGL(fit_kmeans_update_kernel, dim3((unsigned)S), dim3(64), 0, st, k, D, sstride, (const double*)f->d_sums, exact_order ? 0 : 1,
                               f->d_cent, f->d_active, f->d_changed, f->d_iters, f->d_counter);
            GH_HIP(hipGetLastError());
        }
        it += blkn;
        int n_active = 0, flags = 0;
        rc = fit_poll(f, comm, &n_active, &flags);
        if (rc) return rc;
        if (n_active == 0) break;
    }
    if (!exact_order && !comm) {
        // the centroids the reference returns: the mean of every cluster's frames summed IN FRAME ORDER (one sequential
        // pass per k-means call instead of one per iteration)
        static const bool narrow = [] { const char* e = getenv("GMMHMM_SEGSUM"); return e && !strcmp(e, "narrow"); }();
        if (D <= 64 && !narrow)
            hipLaunchKernelGGL(kmeans_rowsum_wide_kernel, dim3((unsigned)S, (unsigned)k), dim3(SS_WAVES * 64), (size_t)2 * SS_PHASE * 64 * 8, st, X, D, k,
                               f->d_segoff, (const uint8_t*)nullptr, f->d_lists, f->d_counts, f->d_cbase, f->d_sums, sstride);
        else
            hipLaunchKernelGGL(kmeans_rowsum_kernel<0>, skz, dim3(64), 0, st, X, D, k, f->d_segoff, (const uint8_t*)nullptr, f->d_lists, f->d_counts,
                               f->d_cbase, f->d_sums, sstride, (const double*)nullptr, 0, 0, 0);
        hipLaunchKernelGGL(fit_final_centroids_kernel, dim3((unsigned)((S * k * D + 255) / 256)), dim3(256), 0, st, S, k, D, sstride,
                           (const double*)f->d_sums, f->d_cent);
        GH_HIP(hipGetLastError());
    }
    // ---- results: centroids, partition variances, cluster sizes (np.unique counts; over all ranks when sharded) ----
    hipLaunchKernelGGL(fit_counts_kernel, dim3((unsigned)((S * k + 255) / 256)), dim3(256), 0, st, S * k, (const int32_t*)f->d_counts, f->d_sq);
    GH_HIP(hipGetLastError());
    if (comm) { rc = gh_comm_allreduce_enqueue(comm, f->d_sq, (int64_t)S * k); if (rc) return rc; }
    rc = gh_stream_wait(ctx, comm, "gh_fit_kmeans");      // (before the copies into pageable memory, which block on the stream)
    if (rc) return rc;
    if (out_centroids) GH_HIP(hipMemcpyAsync(out_centroids, f->d_cent, (size_t)S * k * D * 8, hipMemcpyDeviceToHost, st));
    if (out_cov) GH_HIP(hipMemcpyAsync(out_cov, f->d_cov, (size_t)S * k * D * 8, hipMemcpyDeviceToHost, st));
    if (out_counts) GH_HIP(hipMemcpyAsync(out_counts, f->d_sq, (size_t)S * k * 8, hipMemcpyDeviceToHost, st));
    if (out_iters) GH_HIP(hipMemcpyAsync(out_iters, f->d_iters, (size_t)S * 4, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}

extern "C" int gh_fit_clusters(gh_ctx* ctx, gh_fit* f, int32_t* out /*[N]*/) {
    GH_REQUIRE(ctx && f && out && f->ctx == ctx, "gh_fit_clusters: NULL argument / foreign context");
    if (f->N == 0) return GH_OK;
    GH_HIP(hipSetDevice(ctx->device));
    GH_HIP(hipMemcpyAsync(out, f->d_ids, (size_t)f->N * 4, hipMemcpyDeviceToHost, ctx->stream));
    GH_HIP(hipStreamSynchronize(ctx->stream));
    return GH_OK;
}

extern "C" int gh_fit_em(gh_ctx* ctx, gh_fit* f, gh_comm* comm, int k, double* mean_io, double* var_io, double* weight_io,
                         double* mu_old_io, double* sigma_old_io, double* w_old_io, const double* n_frames, int max_iteration,
                         int check_every, int32_t* out_converged_at) {
    GH_REQUIRE(ctx && f && mean_io && var_io && weight_io && mu_old_io && sigma_old_io && w_old_io && n_frames && f->ctx == ctx,
               "gh_fit_em: NULL argument / foreign context");
    GH_REQUIRE(k >= 1 && k <= f->kmax, "gh_fit_em: k=%d (1..%d)", k, f->kmax);
    GH_REQUIRE(!comm || gh_comm_context(comm) == ctx, "gh_fit_em: the communicator belongs to another context");
    GH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int S = f->S, D = f->D, Wd = 1 + 2 * D, plen = k * Wd + 1;
    const size_t skd = (size_t)S * k * D * 8, sk = (size_t)S * k * 8;
    const double* X = (const double*)f->b->feats;
    GH_HIP(hipMemcpyAsync(f->d_mean, mean_io, skd, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(f->d_var, var_io, skd, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(f->d_weight, weight_io, sk, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(f->d_old_mu, mu_old_io, skd, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(f->d_old_sigma, sigma_old_io, skd, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(f->d_old_w, w_old_io, sk, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(f->d_nframes, n_frames, (size_t)S * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemsetAsync(f->d_active, 1, (size_t)S, st));
    GH_HIP(hipMemsetAsync(f->d_iters, 0xFF, (size_t)S * 4, st));        // -1: did not converge within max_iteration
    GH_HIP(hipMemsetAsync(f->d_counter, 0, 8, st));
    GH_HIP(hipMemsetAsync(f->d_stats, 0, (size_t)S * plen * 8, st));
    const size_t lds = ((size_t)2 * k * D + k + (size_t)k * 64 + 64 * (size_t)(D + 1)) * 8 + 16;
    check_every = std::max(1, check_every);
    int rc;
    // Streaming matrix-core form (gh_refit_mfma.hip): E-step sums, their reduction, GMM.em_update and the next iteration's
    // operands in ONE launch per iteration (sums, collective, update with a communicator)
    if (f->mfma && k <= f->kcap && rf_supported(k, D)) {
        std::vector<double> shift;
        fit_shift_points(S, k, D, mean_io, shift);
        GH_HIP(hipMemcpyAsync(f->d_shift, shift.data(), shift.size() * 8, hipMemcpyHostToDevice, st));
        GH_HIP(hipMemsetAsync(f->d_counter, 0, 64, st));
        GH_HIP(hipMemsetAsync(f->d_done, 0, (size_t)S * 4, st));
        rf_em_args a;
        memset(&a, 0, sizeof a);
        a.c.X = X; a.c.D = D; a.c.k = k; a.c.S = S; a.c.items = f->d_items; a.c.item_ptr = f->d_iptr; a.c.shift = f->d_shift;
        a.c.active = f->d_active; a.c.done = f->d_done; a.c.partial = f->d_rpart; a.c.counter = f->d_counter; a.c.fused = comm ? 0 : 1;
        a.P = f->d_P; a.exp_tab = ctx->d_fp64_tables; a.stats = f->d_stats; a.nframes = f->d_nframes; a.mean = f->d_mean; a.var = f->d_var;
        a.weight = f->d_weight; a.old_mu = f->d_old_mu; a.old_sigma = f->d_old_sigma; a.old_w = f->d_old_w; a.conv_at = f->d_iters;
        rc = rf_launch_em_update(ctx, a, 1);
        if (rc) return rc;
        const bool tail_ok = f->tail && !comm;
        int tail_n = 0;                     // > 0: the next block of iterations is ONE launch over these items
        f->tail_n = 0;
        for (int it = 0; it < max_iteration;) {
            int blkn = std::min(check_every, max_iteration - it);
            GH_HIP(hipMemsetAsync(f->d_counter + 2, 0, 32, st));      // the per-iteration slots of this block
            bool tailed = false;
            if (tail_n > 0) {
                const int blk_tail = std::min(FIT_TAIL_ITERS, max_iteration - it);
                blkn = blk_tail;
                a.c.it = it; a.c.item_ids = f->d_tail_ids; a.c.gen = f->d_gen; a.c.n_iter = blkn;
                rc = rf_launch_em(ctx, a, tail_n);
                if (rc < 0) return rc;
                tailed = rc == 0;
                if (tailed) { ++f->n_tail_launches; f->n_tail_iters += blkn; } else { ++f->n_tail_refused; blkn = std::min(check_every, max_iteration - it); }
                a.c.item_ids = nullptr; a.c.gen = nullptr; a.c.n_iter = 1;
            }
            for (int j = 0; !tailed && j < blkn; ++j) {
                a.c.it = it + j;
                rc = rf_launch_em(ctx, a, f->n_items);
                if (rc) return rc;
                ++f->n_plain_launches;
                if (comm) {
                    hipLaunchKernelGGL(fit_zero_inactive_kernel, dim3((unsigned)((plen + 127) / 128), (unsigned)S), dim3(128), 0, st, S, plen,
                                       (const uint8_t*)f->d_active, f->d_stats);
                    rc = gh_comm_allreduce_enqueue(comm, f->d_stats, (int64_t)S * plen);
                    if (!rc) rc = rf_launch_em_update(ctx, a, 0);
                    if (rc) return rc;
                }
            }
            it += blkn;
            int n_active = 0, flags = 0;
            rc = fit_poll(f, comm, &n_active, &flags, 2 + ((it - 1) & 7), tail_ok);
            if (rc) return rc;
            if (flags & 16) {
                gh_set_error("gh_fit_em: a variance is 0 (singular covariance)");
                return GH_ERR_INVALID;
            }
            if (flags & 32) { gh_set_error("gh_fit_em: a workgroup of a tail launch waited for its state in vain"); return GH_ERR_HIP; }
            if (n_active == 0) break;
            tail_n = tail_ok ? fit_tail_items(f, 4096) : 0;
        }
    } else
    for (int it = 0; it < max_iteration;) {
        const int blkn = std::min(check_every, max_iteration - it);
        for (int j = 0; j < blkn; ++j) {
            hipLaunchKernelGGL(fit_em_prepare_kernel, dim3((unsigned)((S * k + 63) / 64)), dim3(64), 0, st, S, k, D, (const double*)f->d_var,
                               (const double*)f->d_weight, (const uint8_t*)f->d_active, f->d_ivar, f->d_logc, f->d_counter);
            if (f->n_tiles > 0) {
                const dim3 grid((unsigned)f->n_tiles), blk(64 * GH_EM_WAVES);
#define GH_EM(DR) hipLaunchKernelGGL((em_multi_kernel<DR>), grid, blk, lds, st, X, D, k, f->d_tiles, (const double*)f->d_mean, \
                                     (const double*)f->d_ivar, (const double*)f->d_logc, f->d_part, (const uint8_t*)f->d_active)
                if (D <= 16) GH_EM(16); else if (D <= 40) GH_EM(40); else GH_EM(64);
#undef GH_EM
            }
            hipLaunchKernelGGL(tiles_reduce_kernel, dim3((unsigned)((plen + 63) / 64), (unsigned)S), dim3(256), 0, st, (const double*)f->d_part,
                               (const int32_t*)f->d_tptr, plen, f->d_stats, (const uint8_t*)f->d_active);
            if (comm) {
                hipLaunchKernelGGL(fit_zero_inactive_kernel, dim3((unsigned)((plen + 127) / 128), (unsigned)S), dim3(128), 0, st, S, plen,
                                   (const uint8_t*)f->d_active, f->d_stats);
                rc = gh_comm_allreduce_enqueue(comm, f->d_stats, (int64_t)S * plen);
                if (rc) return rc;
            }
            GH_HIP(hipMemsetAsync(f->d_counter, 0, 4, st));
            hipLaunchKernelGGL(fit_em_update_kernel, dim3((unsigned)S), dim3(256), 0, st, k, D, plen, (const double*)f->d_stats,
                               (const double*)f->d_nframes, f->d_mean, f->d_var, f->d_weight, f->d_old_mu, f->d_old_sigma, f->d_old_w,
                               f->d_active, f->d_iters, it + j, f->d_counter);
            GH_HIP(hipGetLastError());
        }
        it += blkn;
        int n_active = 0, flags = 0;
        rc = fit_poll(f, comm, &n_active, &flags);
        if (rc) return rc;
        if (flags & 16) {
            gh_set_error("gh_fit_em: a variance is 0 (singular covariance)");
            return GH_ERR_INVALID;
        }
        if (n_active == 0) break;
    }
    rc = gh_stream_wait(ctx, comm, "gh_fit_em");          // (before the copies into pageable memory, which block on the stream)
    if (rc) return rc;
    GH_HIP(hipMemcpyAsync(mean_io, f->d_mean, skd, hipMemcpyDeviceToHost, st));
    GH_HIP(hipMemcpyAsync(var_io, f->d_var, skd, hipMemcpyDeviceToHost, st));
    GH_HIP(hipMemcpyAsync(weight_io, f->d_weight, sk, hipMemcpyDeviceToHost, st));
    GH_HIP(hipMemcpyAsync(mu_old_io, f->d_old_mu, skd, hipMemcpyDeviceToHost, st));
    GH_HIP(hipMemcpyAsync(sigma_old_io, f->d_old_sigma, skd, hipMemcpyDeviceToHost, st));
    GH_HIP(hipMemcpyAsync(w_old_io, f->d_old_w, sk, hipMemcpyDeviceToHost, st));
    if (out_converged_at) GH_HIP(hipMemcpyAsync(out_converged_at, f->d_iters, (size_t)S * 4, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Segmental k-means of MANY word models in lock-step (kmeans.py:111-155 for every word at once; sr/core.py:57-60 trains
// word after word).  The session's "states" are the words (a word's templates back to back in the batch), the
// "clusters" the n segments of the word: gh_fit_dtw aligns every template against ITS word's segment means in one launch
// and leaves the segment of every frame on the device; gh_fit_group_stats turns those into the segment means (the
// frames of a segment summed in template / frame order: np.mean of segment_data's concatenation) and variances
// (two-pass, ddof 1: np.cov(...).diagonal()) of combine_templates (kmeans.py:15-30).
#include "gh_dtw.h"

extern "C" int gh_fit_set_ids(gh_ctx* ctx, gh_fit* f, const int32_t* ids) {
    GH_REQUIRE(ctx && f && (ids || f->N == 0) && f->ctx == ctx, "gh_fit_set_ids: NULL argument / foreign context");
    if (f->N == 0) return GH_OK;
    GH_HIP(hipSetDevice(ctx->device));
    GH_HIP(hipMemcpyAsync(f->d_ids, ids, (size_t)f->N * 4, hipMemcpyHostToDevice, ctx->stream));
    GH_HIP(hipStreamSynchronize(ctx->stream));
    return GH_OK;
}

extern "C" int gh_fit_dtw(gh_ctx* ctx, gh_fit* f, int n, const double* y, const double* trans, const int32_t* utt_model,
                          const uint8_t* active) {
    GH_REQUIRE(ctx && f && y && trans && f->ctx == ctx, "gh_fit_dtw: NULL argument / foreign context");
    GH_REQUIRE(n > 1 && n <= 255 && n <= f->kmax, "gh_fit_dtw: n=%d (2..min(255, kmax=%d))", n, f->kmax);
    const gh_batch* b = f->b;
    const int64_t U = b->U, N = b->N;
    const int W = f->S, D = f->D;
    GH_REQUIRE(utt_model || U == 0, "gh_fit_dtw: utt_model is NULL");
    if (U == 0) return GH_OK;
    for (int64_t u = 0; u < U; ++u) {
        GH_REQUIRE(utt_model[u] >= 0 && utt_model[u] < W, "gh_fit_dtw: utt_model[%lld]=%d", (long long)u, utt_model[u]);
        GH_REQUIRE(b->offsets[u + 1] - b->offsets[u] > 1, "gh_fit_dtw: utterance %lld has fewer than 2 frames (decode.py:22)", (long long)u);
    }
    GH_HIP(hipSetDevice(ctx->device));
    std::vector<int64_t> moff(U + 1);
    for (int64_t u = 0; u <= U; ++u) moff[u] = (int64_t)n * b->offsets[u];
    double *d_y, *d_tr;
    int64_t* d_moff;
    int32_t* d_um;
    uint8_t *d_bp, *d_act = nullptr;
    Carver cv;
    cv.add(&d_y, (size_t)W * n * D); cv.add(&d_tr, (size_t)W * n * n); cv.add(&d_moff, U + 1); cv.add(&d_um, U);
    cv.add(&d_bp, (size_t)n * N);
    if (active) cv.add(&d_act, W);
    int rc = cv.commit(ctx);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    GH_HIP(hipMemsetAsync(ctx->d_flag, 0, sizeof(int), st));
    GH_HIP(hipMemcpyAsync(d_y, y, (size_t)W * n * D * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(d_tr, trans, (size_t)W * n * n * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(d_moff, moff.data(), (U + 1) * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(d_um, utt_model, U * 4, hipMemcpyHostToDevice, st));
    if (active) GH_HIP(hipMemcpyAsync(d_act, active, W, hipMemcpyHostToDevice, st));
    gh_dtw_args a;
    memset(&a, 0, sizeof a);
    a.x = (const double*)b->feats; a.utt_off = b->d_offsets; a.n = n; a.D = D; a.beam = 0;
    a.y = d_y; a.trans = d_tr; a.bp = d_bp; a.bp_off = d_moff; a.flag = ctx->d_flag;
    a.utt_model = d_um; a.model_active = d_act; a.frame_row = f->d_ids;
    rc = gh_launch_dtw(ctx, a, U);
    if (rc) return rc;
    int flag = 0;
    GH_HIP(hipMemcpyAsync(&flag, ctx->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    if (flag & 2) {
        gh_set_error("gh_fit_dtw: back-trace left the matrix: cell (0,0) is not reachable from the end cell");
        return GH_ERR_INVALID;
    }
    return GH_OK;
}

extern "C" int gh_fit_group_stats(gh_ctx* ctx, gh_fit* f, int k, const uint8_t* active, double* out_mean, double* out_var,
                                  double* out_count) {
    GH_REQUIRE(ctx && f && f->ctx == ctx, "gh_fit_group_stats: NULL argument / foreign context");
    GH_REQUIRE(k >= 1 && k <= f->kmax, "gh_fit_group_stats: k=%d (1..%d)", k, f->kmax);
    GH_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int S = f->S, D = f->D, sstride = k * (D + 1) + 1;
    const double* X = (const double*)f->b->feats;
    const uint8_t* d_act = nullptr;
    if (active) {
        GH_HIP(hipMemcpyAsync(f->d_active, active, (size_t)S, hipMemcpyHostToDevice, st));
        d_act = f->d_active;
    }
    int rc = fit_build_lists(f, k, false, d_act);
    if (rc) return rc;
    const dim3 skz((unsigned)S, (unsigned)k, 1);
    hipLaunchKernelGGL(kmeans_rowsum_kernel<0>, skz, dim3(64), 0, st, X, D, k, f->d_segoff, d_act, f->d_lists, f->d_counts, f->d_cbase,
                       f->d_sums, sstride, (const double*)nullptr, 0, 0, 0);
    hipLaunchKernelGGL(kmeans_rowsum_kernel<1>, skz, dim3(64), 0, st, X, D, k, f->d_segoff, d_act, f->d_lists, f->d_counts, f->d_cbase,
                       f->d_sq, sstride, (const double*)f->d_sums, sstride, 0, 0);
    hipLaunchKernelGGL(fit_partvar_kernel, dim3((unsigned)((S * k * D + 255) / 256)), dim3(256), 0, st, S, k, D, sstride,
                       (const double*)f->d_sums, (const double*)f->d_sq, 0, f->d_cov);
    hipLaunchKernelGGL(fit_final_centroids_kernel, dim3((unsigned)((S * k * D + 255) / 256)), dim3(256), 0, st, S, k, D, sstride,
                       (const double*)f->d_sums, f->d_cent);
    hipLaunchKernelGGL(fit_counts_kernel, dim3((unsigned)((S * k + 255) / 256)), dim3(256), 0, st, S * k, (const int32_t*)f->d_counts, f->d_stats);   // (d_sq keeps the squared sums of the words that sit this round out)
    GH_HIP(hipGetLastError());
    if (out_mean) GH_HIP(hipMemcpyAsync(out_mean, f->d_cent, (size_t)S * k * D * 8, hipMemcpyDeviceToHost, st));
    if (out_var) GH_HIP(hipMemcpyAsync(out_var, f->d_cov, (size_t)S * k * D * 8, hipMemcpyDeviceToHost, st));
    if (out_count) GH_HIP(hipMemcpyAsync(out_count, f->d_stats, (size_t)S * k * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}
