// Viterbi over an HMM state lattice with non-emitting rows (reference:
// decode_hmm_states, sr/recognition/decode.py:80-146).
//
// One utterance per workgroup.  The cost column lives in LDS (prev / cur, fp64);
// the rows of a column are processed in LEVELS: level 0 = rows whose arcs only
// read the previous column, level k = rows with a same-column arc (an arc that
// touches a non-emitting row, decode.py:109-111) from a level k-1 row.  Inside a
// level rows are independent and spread across lanes; one barrier per level
// reproduces the reference's ascending-row sweep exactly.  Ties: candidates are
// scanned in ascending origin order with a strict '<' (np.argmin, decode.py:118).
// Emission costs come from the resident [N,S] likelihood matrix, one coalesced
// S-vector per column staged in LDS (non-emitting rows add 0).
// Back-pointers: uint16 per cell in HBM scratch (origin row | same-column bit),
// walked by lane 0 once the last column is done.
#include "gh_internal.h"
#include "gh_viterbi.h"

namespace {

constexpr uint16_t BP_NONE = 0xFFFFu;

// Generic kernel: any graph (any number of levels / rows / arcs, NaN arc costs, same-column self
// arcs).  The common graphs take the lean kernel in gh_viterbi_lean.hip instead.
template <typename ET, bool WANT_PATH>
__global__ void viterbi_kernel(gh_vit_args a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    __shared__ int s_bi;
    const int tid = threadIdx.x, bd = blockDim.x;
    const int64_t slot = a.u_begin + blockIdx.x;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int l = a.utt_lat ? a.utt_lat[u] : 0;
    const gh_lattices::desc dsc = a.descs[l];
    const int R = dsc.R, nlev = dsc.nlev, n_end = dsc.n_end;
    const int32_t* row_state = a.row_state + dsc.row_base;
    const uint8_t* row_start = a.row_start + dsc.row_base;
    const int32_t* pred_ptr = a.pred_ptr + dsc.ptr_base;
    const uint32_t* pred_row = a.pred_row + dsc.arc_base;
    const double* pred_cost = a.pred_cost + dsc.arc_base;
    const int32_t* order = a.order + dsc.row_base;
    const int32_t* level_ptr = a.level_ptr + dsc.lev_base;
    const int32_t* end_rows = a.end_rows + dsc.end_base;
    const int64_t f0 = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - f0);
    const int S = a.S;
    const ET* nll = static_cast<const ET*>(a.nll) + f0 * S;

    double* prev = lds;
    double* cur = lds + a.r_pad;
    double* em = lds + 2 * a.r_pad;
    double* costs = a.costs ? a.costs + a.costs_off[u] : nullptr;
    uint16_t* bp = WANT_PATH ? a.bp + a.bp_off[slot] : nullptr;
    const double INF = INFINITY;

    if (T <= 0) {
        if (tid == 0) {
            if (a.best_end) a.best_end[u] = -1;
            if (a.path_len) a.path_len[u] = 0;
        }
        return;
    }
    for (int r = tid; r < R; r += bd) {
        prev[r] = INF;
        cur[r] = INF;
    }
    __syncthreads();

    if (T == 1) {
        // decode.py:113 with c == 0: column c-1 == -1 wraps onto column 0 ITSELF, so every
        // arc reads the partially filled current column; serial ascending-row sweep.
        for (int s = tid; s < S; s += bd) em[s] = (double)nll[s];
        __syncthreads();
        if (tid == 0) {
            for (int r = 0; r < R; ++r) {
                const int st = row_state[r];
                const double e = st >= 0 ? em[st] : 0.0;
                double c = INF;
                if (row_start[r] & 1) {
                    c = e;
                } else {
                    const int p0 = pred_ptr[r], p1 = pred_ptr[r + 1];
                    if (p0 < p1) {
                        double best = 0;
                        for (int p = p0; p < p1; ++p) {
                            const double v = pred_cost[p] + cur[pred_row[p] & GH_ARC_ROW];
                            if (p == p0 || v < best || (v != v && best == best)) best = v;
                        }
                        c = best + e;
                        if (c != c) c = INF;
                    }
                }
                cur[r] = c;
                if (costs) costs[r] = c;
            }
        }
        __syncthreads();
        double* t_ = prev; prev = cur; cur = t_;
    } else {
        for (int t = 0; t < T; ++t) {
            for (int s = tid; s < S; s += bd) em[s] = (double)nll[(int64_t)t * S + s];
            __syncthreads();
            for (int lev = 0; lev < nlev; ++lev) {
                const int i1 = level_ptr[lev + 1];
                for (int i = level_ptr[lev] + tid; i < i1; i += bd) {
                    const int r = order[i];
                    const int st = row_state[r];
                    const double e = st >= 0 ? em[st] : 0.0;
                    double c = INF;
                    uint16_t b = BP_NONE;
                    if (t == 0 && (row_start[r] & 1)) {
                        c = e;  // decode.py:99-101
                    } else {
                        const int p0 = pred_ptr[r], p1 = pred_ptr[r + 1];
                        if (p0 < p1) {  // rows without finite arcs are skipped (decode.py:116-117)
                            double best = 0;
                            uint32_t bw = 0;
                            for (int p = p0; p < p1; ++p) {
                                const uint32_t w = pred_row[p];
                                const int o = (int)(w & GH_ARC_ROW);
                                double v;
                                if (w & GH_ARC_DEAD) v = INF;
                                else v = pred_cost[p] + ((w & GH_ARC_SAME) ? cur[o] : prev[o]);
                                // np.argmin: the first NaN wins over everything, else the first minimum
                                if (p == p0 || v < best || (v != v && best == best)) { best = v; bw = w; }
                            }
                            c = best + e;
                            if (c != c) c = INF;  // min(inf, nan) keeps inf (decode.py:124)
                            const int o = (int)(bw & GH_ARC_ROW);
                            if ((bw & GH_ARC_SAME) && o == r) atomicOr(a.flag, 1);  // decode.py:120-121
                            b = (uint16_t)(o | ((bw & GH_ARC_SAME) ? 0x8000u : 0u));
                        }
                    }
                    cur[r] = c;
                    if (WANT_PATH) bp[(int64_t)t * R + r] = b;
                    if (costs) costs[(int64_t)r * T + t] = c;
                }
                __syncthreads();
            }
            if (a.beam > 0 && t < T - 1) {
                // Rank beam (the one of dtw, decode.py:62-68, carried over to lattices; gh_lattices_set_beam): cells of the
                // FINISHED column ranked >= beam in ascending (cost, row) order are pruned -- they read +inf as origins of
                // the next column.  Counting pairs is O(R^2) per column; a pruned decode of a few hundred rows is a
                // feature for parity and experiments, not a speed-up (see profiles/: the active-row histogram of C5).
                int* prn = reinterpret_cast<int*>(em + S);
                for (int r = tid; r < R; r += bd) {
                    const double c = cur[r];
                    int rank = 0;
                    if (c < INF) {
                        for (int o = 0; o < R; ++o) {
                            const double v = cur[o];
                            rank += (v < c) || (v == c && o < r);
                        }
                    }
                    prn[r] = (c < INF) && rank >= a.beam;
                }
                __syncthreads();
                for (int r = tid; r < R; r += bd)
                    if (prn[r]) {
                        cur[r] = INF;
                        if (costs) costs[(int64_t)r * T + t] = INF;
                    }
                __syncthreads();
            }
            double* t_ = prev; prev = cur; cur = t_;
        }
    }
    // `prev` now holds the last column.
    if (tid == 0) {
        double best = INF;
        int bi = -1;
        double* ec = a.end_cost ? a.end_cost + (a.end_off ? a.end_off[u] : u * n_end) : nullptr;
        for (int k = 0; k < n_end; ++k) {
            const double c = prev[end_rows[k]];
            if (ec) ec[k] = c;
            if (best >= c) { best = c; bi = k; }  // '>=': last minimum wins (decode.py:131)
        }
        if (a.best_end) a.best_end[u] = bi;
        s_bi = bi;
    }
    __syncthreads();
    // ---- back-trace (decode.py:143-145).  Lane 0 walks the back-pointers; the (row, col) pairs are
    // parked in LDS and flushed by the whole workgroup, so the dependent bp loads of the walk are not
    // serialised behind path stores (vmcnt retires in order).
    if (WANT_PATH) {
        __shared__ int s_state[4];  // i, j, len, done
        int32_t* pbuf = reinterpret_cast<int32_t*>(lds);  // the cost columns are dead now
        const int PB = a.r_pad + S / 2 - 1;                // pairs per flush (cost columns + emission vector)
        int32_t* path = a.path + 2 * a.path_off[u];
        const int64_t cap = a.path_off[u + 1] - a.path_off[u];
        if (tid == 0) {
            const int bi = s_bi;
            s_state[0] = (bi >= 0) ? end_rows[bi] : 0;
            s_state[1] = T - 1;
            s_state[2] = 0;
            s_state[3] = !(T > 1 && bi >= 0);
        }
        __syncthreads();
        while (!s_state[3]) {
            int n_new = 0;
            if (tid == 0) {
                int i = s_state[0], j = s_state[1], len = s_state[2];
                while (j != 0 && n_new < PB) {
                    const uint16_t b = bp[(int64_t)j * R + i];
                    if (b == BP_NONE) { atomicOr(a.flag, 2); j = 0; break; }
                    // all-inf cells may point along dead same-column arcs and cycle (the reference
                    // would spin forever); a live path has at most one cell per (column, level)
                    if (len + n_new >= cap) { atomicOr(a.flag, 4); j = 0; break; }
                    i = b & 0x7FFF;
                    if (!(b & 0x8000u)) --j;
                    pbuf[2 * n_new] = i;
                    pbuf[2 * n_new + 1] = j;
                    ++n_new;
                }
                s_state[0] = i; s_state[1] = j;
                s_state[3] = (j == 0);
                pbuf[2 * PB] = n_new;
            }
            __syncthreads();
            n_new = pbuf[2 * PB];
            const int len = s_state[2];
            for (int k = tid; k < 2 * n_new; k += bd) path[2 * (int64_t)len + k] = pbuf[k];
            __syncthreads();
            if (tid == 0) s_state[2] = len + n_new;
            __syncthreads();
        }
        if (tid == 0) a.path_len[u] = s_state[2];
    }
}

}  // namespace

int gh_launch_viterbi(gh_ctx* ctx, const gh_vit_args& a, int64_t n_utts, int block, size_t lds_bytes,
                      bool f64, bool want_path) {
    if (n_utts <= 0) return GH_OK;
    dim3 grid((unsigned)n_utts), blk((unsigned)block);
    if (f64) {
        if (want_path) hipLaunchKernelGGL((viterbi_kernel<double, true>), grid, blk, lds_bytes, ctx->stream, a);
        else hipLaunchKernelGGL((viterbi_kernel<double, false>), grid, blk, lds_bytes, ctx->stream, a);
    } else {
        if (want_path) hipLaunchKernelGGL((viterbi_kernel<float, true>), grid, blk, lds_bytes, ctx->stream, a);
        else hipLaunchKernelGGL((viterbi_kernel<float, false>), grid, blk, lds_bytes, ctx->stream, a);
    }
    GH_HIP(hipGetLastError());
    return GH_OK;
}
