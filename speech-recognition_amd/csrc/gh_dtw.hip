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

template <typename BPT>
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
    if (T <= 0) {
        if (i == 0 && a.path_len) a.path_len[u] = 0;
        return;
    }
    // per-lane template row kept in registers when it fits, else re-read from memory
    const double* yrow = a.y ? a.y + (int64_t)(act ? i : 0) * D : nullptr;
    const double* vrow = a.var ? a.var + (int64_t)(act ? i : 0) * D : nullptr;
    double logdet = 0;
    if (a.var && act) logdet = a.logdet[i];
    col(0)[i] = INF; col(1)[i] = INF;
    mark(0)[i] = 0; mark(1)[i] = 0;
    __syncthreads();
    int pb = 0;  // buffer holding column j-1
    for (int j = 0; j < T; ++j) {
        const int cb = pb ^ 1;
        double dist = 0;
        if (act) {
            if (a.E) {
                dist = a.E[a.e_off[u] + (int64_t)i * T + j];
            } else {
                const double* x = a.x + (f0 + j) * D;
                double q = 0;
                if (a.var) {
                    for (int d = 0; d < D; ++d) { const double t = x[d] - yrow[d]; q += t / vrow[d] * t; }
                    dist = logdet + 0.5 * q;
                } else {
                    for (int d = 0; d < D; ++d) { const double t = x[d] - yrow[d]; q = fma(t, t, q); }
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
                    const double v = a.trans[i * n + o] + (pruned ? INF : col(pb)[o]);
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
        __syncthreads();
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
        __syncthreads();
        mark(cb)[i] = mk;
        __syncthreads();
        pb = cb;
    }
    if (i == 0 && a.path) {
        int32_t* path = a.path + 2 * a.path_off[u];
        const int64_t cap = a.path_off[u + 1] - a.path_off[u];
        int r = n - 1, j = T - 1, len = 0;
        while (r != 0 || j != 0) {  // decode.py:74
            if (j <= 0 || len >= cap) { atomicOr(a.flag, 2); break; }  // would wrap to column -1
            r = bp[(int64_t)j * n + r];
            --j;
            path[2 * len] = r;
            path[2 * len + 1] = j;
            ++len;
        }
        a.path_len[u] = len;
    }
}

}  // namespace

int gh_launch_dtw(gh_ctx* ctx, const gh_dtw_args& a, int64_t U) {
    if (U <= 0) return GH_OK;
    const int np = (a.n + 63) & ~63;
    const size_t lds = (size_t)2 * np * 8 + (size_t)2 * np;
    if (a.n <= 255) hipLaunchKernelGGL(dtw_kernel<uint8_t>, dim3((unsigned)U), dim3((unsigned)np), lds, ctx->stream, a);
    else hipLaunchKernelGGL(dtw_kernel<uint16_t>, dim3((unsigned)U), dim3((unsigned)np), lds, ctx->stream, a);
    GH_HIP(hipGetLastError());
    return GH_OK;
}
