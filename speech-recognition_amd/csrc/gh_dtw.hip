// Template-matching dynamic program with optional beam (reference: dtw,
// sr/recognition/decode.py:7-77).
//
// One utterance per workgroup: thread i owns template row i (one wave for n <= 64 -- the reference's 5-state word
// templates -- up to 16 waves for n <= 1024; back-pointers are bytes up to 255 rows, 16-bit above).  Per column the
// lanes compute their distance to the frame (Euclidean norm or the diagonal-
// Gaussian negative log-likelihood `mahalanobis`, hmm_state.py:48-58) -- or read
// a caller-supplied distance matrix -- then minimise over ALL origins of the
// previous column in ascending order with a strict '<' (np.argmin over the full
// candidate list, decode.py:44-53: +inf arcs are candidates too, so an
// unreachable cell points at the first candidate).  Beam (decode.py:62-68): the
// cells ranked >= beam in the finished column are marked; a marked origin is
// dropped from the candidate list of row 0 only (row 0 is the first reader and
// turns the mark back into +inf, decode.py:46-48) and reads as +inf for the other
// rows.  The previous column lives in LDS; back-pointers (origin row, uint8) go to
// HBM scratch and lane 0 walks them until cell (0,0) (decode.py:70-76).
#include "gh_internal.h"
#include "gh_dtw.h"

namespace {

// workgroup barrier that orders LDS traffic only: the frame prefetch and the back-pointer / cost stores stay in flight
// (a full __syncthreads waits for their round trips: three per column)
__device__ __forceinline__ void dtw_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// STAGED: template rows, variances, arc costs and the current frame are read from LDS (address space known at compile
// time: with one pointer that may be LDS or global the compiler falls back to flat loads and spills the selects)
template <typename BPT, bool STAGED>
__global__ __launch_bounds__(1024) void dtw_kernel(gh_dtw_args a) {
    extern __shared__ __attribute__((aligned(16))) double dtw_lds[];
    const int np = blockDim.x;                                // n rounded up to whole waves
    double* col0 = dtw_lds;                                   // [2][np] columns, then [2][np] beam marks
    unsigned char* mark0 = reinterpret_cast<unsigned char*>(dtw_lds + 2 * np);
    auto col = [&](int b) { return col0 + b * np; };
    auto mark = [&](int b) { return mark0 + b * np; };
    const int i = threadIdx.x;
    const int64_t u = blockIdx.x;
    const int n = a.n, D = a.D;
    const int64_t f0 = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - f0);
    const double INF = INFINITY;
    double* costs = a.costs ? a.costs + a.costs_off[u] : nullptr;
    BPT* bp = reinterpret_cast<BPT*>(a.bp) + a.bp_off[u];
    const bool act = i < n;
    const int model = a.utt_model ? a.utt_model[u] : 0;
    if (a.model_active && !a.model_active[model]) return;     // (uniform for the workgroup)
    if (T <= 0) {
        if (i == 0 && a.path_len) a.path_len[u] = 0;
        return;
    }
    // this utterance's template set
    const double* m_y = a.y ? a.y + (int64_t)model * n * D : nullptr;
    const double* m_var = a.var ? a.var + (int64_t)model * n * D : nullptr;
    const double* m_logdet = a.logdet ? a.logdet + (int64_t)model * n : nullptr;
    const double* m_trans = a.trans + (int64_t)model * n * n;
    // built-in distances: the template rows (and their variances) are staged in LDS once when they fit, and the frame
    // of column j + 1 travels from HBM into registers while column j is computed -- the distance loop itself only
    // reads LDS.  (Reading template row and frame element by element from global memory inside the loop cost one
    // memory round trip per dimension: 16 k cycles per column for D = 39.)
    double* xs0 = reinterpret_cast<double*>(mark0 + 2 * np + ((8 - (2 * np) % 8) % 8));   // [2][D] frames, 8-byte aligned
    double* ys = xs0 + 2 * D;                                 // [n][D] template rows (a.tpl_lds)
    double* vs = ys + (STAGED ? n * D : 0);                   // [n][D] variances
    double* ts = vs + ((STAGED && a.var) ? n * D : 0);        // [n][n] arc costs
    const bool own_dist = a.E == nullptr;
    if (STAGED) {
        for (int k = i; k < n * n; k += np) ts[k] = m_trans[k];
        if (own_dist) for (int k = i; k < n * D; k += np) { ys[k] = m_y[k]; if (a.var) vs[k] = m_var[k]; }
    }
    const double* g_trow = m_trans + (int64_t)(act ? i : 0) * n;
    const double* g_yrow = own_dist ? m_y + (int64_t)(act ? i : 0) * D : nullptr;
    const double* g_vrow = (own_dist && a.var) ? m_var + (int64_t)(act ? i : 0) * D : nullptr;
    const double* l_trow = ts + (act ? i : 0) * n;
    const double* l_yrow = ys + (act ? i : 0) * D;
    const double* l_vrow = vs + (act ? i : 0) * D;
    double logdet = 0;
    if (a.var && act) logdet = m_logdet[i];
    col(0)[i] = INF; col(1)[i] = INF;
    mark(0)[i] = 0; mark(1)[i] = 0;
    constexpr int XR = 4;                                     // frame elements per thread (D <= 4 * threads; else global reads)
    const bool x_lds = STAGED && own_dist;                    // (the host only picks STAGED when D <= XR * threads)
    double xn[XR];
    auto x_fetch = [&](int j) {
#pragma unroll
        for (int q = 0; q < XR; ++q) {
            const int d = i + q * np;
            xn[q] = (x_lds && j < T && d < D) ? a.x[(f0 + j) * D + d] : 0.0;
        }
    };
    auto x_park = [&](int buf) {
#pragma unroll
        for (int q = 0; q < XR; ++q) {
            const int d = i + q * np;
            if (x_lds && d < D) xs0[buf * D + d] = xn[q];
        }
    };
    x_fetch(0);
    x_park(0);
    __syncthreads();
    int pb = 0;  // buffer holding column j-1
    for (int j = 0; j < T; ++j) {
        const int cb = pb ^ 1;
        x_fetch(j + 1);                                       // in flight during this column
        double dist = 0;
        if (act) {
            if (a.E) {
                dist = a.E[a.e_off[u] + (int64_t)i * T + j];
            } else {
                const double* xl = xs0 + (j & 1) * D;
                const double* xg = a.x + (f0 + j) * D;
                double q = 0;
                // (unrolled: the reads of 8 dimensions are in flight together; the sum keeps its order)
                if (a.var) {
#pragma unroll 8
                    for (int d = 0; d < D; ++d) {
                        const double t = STAGED ? xl[d] - l_yrow[d] : xg[d] - g_yrow[d];
                        q += t / (STAGED ? l_vrow[d] : g_vrow[d]) * t;
                    }
                    dist = logdet + 0.5 * q;
                } else {
#pragma unroll 8
                    for (int d = 0; d < D; ++d) { const double t = STAGED ? xl[d] - l_yrow[d] : xg[d] - g_yrow[d]; q = fma(t, t, q); }
                    dist = sqrt(q);
                }
            }
        }
        double c = INF;
        if (act) {
            if (i == 0 && j == 0) {
                c = dist;  // decode.py:34-40
                bp[0] = (BPT)~BPT(0);
            } else {
                double best = 0;
                int bo = -1;
                for (int o = 0; o < n; ++o) {
                    const bool pruned = mark(pb)[o] != 0;
                    if (pruned && i == 0) continue;  // first reader drops the marked cell
                    const double v = (STAGED ? l_trow[o] : g_trow[o]) + (pruned ? INF : col(pb)[o]);
                    // np.argmin: the first NaN wins over everything, else the first minimum
                    if (bo < 0 || v < best || (v != v && best == best)) { best = v; bo = o; }
                }
                c = best + dist;
                if (c != c) c = INF;  // min(inf, nan) keeps inf (decode.py:60)
                if (bo < 0) { atomicOr(a.flag, 8); bo = 0; c = INF; }  // np.argmin([]) -> ValueError
                bp[(int64_t)j * n + i] = (BPT)bo;
            }
        }
        col(cb)[i] = c;
        mark(cb)[i] = 0;
        dtw_lds_barrier();
        unsigned char mk = 0;
        if (a.beam > 0 && act) {
            // rank in ascending (value, row) order == position in np.argsort of the column
            int rank = 0;
            for (int o = 0; o < n; ++o) {
                const double v = col(cb)[o];
                rank += (v < c) || (v == c && o < i);
            }
            if (rank >= a.beam && !isinf(c)) mk = 1;
        }
        if (costs && act) {
            // what the reference returns: a marked cell reads -1 in the last column (never reset)
            // and +inf elsewhere (reset by the next column's row 0)
            costs[(int64_t)i * T + j] = mk ? ((j == T - 1) ? -1.0 : INF) : c;
        }
        dtw_lds_barrier();
        mark(cb)[i] = mk;
        x_park((j + 1) & 1);
        dtw_lds_barrier();
        pb = cb;
    }
    __syncthreads();   // (full barrier: the back-pointers written above are read back below)
    if (a.path || a.frame_row) {
        // the walk is a chain of dependent reads (one per column): blocks of back-pointer columns are staged into LDS
        // by all threads (coalesced) and thread 0 walks them there
        BPT* win = reinterpret_cast<BPT*>(dtw_lds);            // the column buffers are free now
        const int cap_cols = (int)(((size_t)2 * np * 8 + (size_t)2 * np) / sizeof(BPT) / n);   // columns that fit (>= 8)
        __shared__ int s_state[3];                            // r, j, done
        int32_t* path = a.path ? a.path + 2 * a.path_off[u] : nullptr;
        const int64_t cap = a.path ? a.path_off[u + 1] - a.path_off[u] : (int64_t)T;
        int32_t* frow = a.frame_row ? a.frame_row + f0 : nullptr;
        int len = 0;
        if (i == 0) { s_state[0] = n - 1; s_state[1] = T - 1; s_state[2] = 0; if (frow) frow[T - 1] = n - 1; }
        __syncthreads();
        while (!s_state[2]) {
            const int jhi = s_state[1];                        // highest column still to be read
            const int jlo = max(0, jhi - cap_cols + 1);
            __syncthreads();
            for (int k = i; k < (jhi - jlo + 1) * n; k += np) win[k] = bp[(int64_t)jlo * n + k];
            __syncthreads();
            if (i == 0) {
                int r = s_state[0], j = jhi;
                bool done = false;
                while (true) {
                    if (r == 0 && j == 0) { done = true; break; }                      // decode.py:74
                    if (j <= 0 || len >= cap) { atomicOr(a.flag, 2); done = true; break; }  // would wrap to column -1
                    if (j < jlo) break;                                                // next block
                    r = win[(j - jlo) * n + r];
                    --j;
                    if (path) { path[2 * len] = r; path[2 * len + 1] = j; }
                    if (frow) frow[j] = r;
                    ++len;
                }
                s_state[0] = r; s_state[1] = j; s_state[2] = done ? 1 : 0;
            }
            __syncthreads();
        }
        if (i == 0 && a.path_len) a.path_len[u] = len;
    }
}

}  // namespace

int gh_launch_dtw(gh_ctx* ctx, const gh_dtw_args& a, int64_t U) {
    if (U <= 0) return GH_OK;
    const int np = (a.n + 63) & ~63;
    const size_t base = (size_t)2 * np * 8 + (size_t)2 * np + 8 + (size_t)2 * a.D * 8;
    const size_t extra = (size_t)a.n * a.n * 8 + (a.E ? 0 : (size_t)a.n * a.D * 8 * (a.var ? 2 : 1));
    const bool staged = extra <= 96 * 1024 && a.D <= 4 * np;      // templates, variances and arc costs fit LDS
    const size_t lds = base + (staged ? extra : 0);
    if (a.n <= 255) {
        if (staged) hipLaunchKernelGGL((dtw_kernel<uint8_t, true>), dim3((unsigned)U), dim3((unsigned)np), lds, ctx->stream, a);
        else hipLaunchKernelGGL((dtw_kernel<uint8_t, false>), dim3((unsigned)U), dim3((unsigned)np), lds, ctx->stream, a);
    } else {
        if (staged) hipLaunchKernelGGL((dtw_kernel<uint16_t, true>), dim3((unsigned)U), dim3((unsigned)np), lds, ctx->stream, a);
        else hipLaunchKernelGGL((dtw_kernel<uint16_t, false>), dim3((unsigned)U), dim3((unsigned)np), lds, ctx->stream, a);
    }
    GH_HIP(hipGetLastError());
    return GH_OK;
}
