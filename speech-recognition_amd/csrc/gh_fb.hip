// Forward-backward over an HMM state lattice with non-emitting rows -- the sum-product twin of
// the Viterbi kernel (gh_viterbi.hip).  NOT IN THE REFERENCE (the
// reference trains by Viterbi alignment only); defined as: same arcs, same same-column rule for
// arcs touching a non-emitting row (decode.py:109-111), start only at the start rows in column 0,
// end set = the graph's end rows in the last column, min -> -logsumexp.  Pinned by brute-force
// path enumeration (tests), not by reference outputs.
//
// One utterance per workgroup, log domain, fp64.  Forward: alpha column double-buffered in LDS,
// rows swept level by level (ascending) exactly like the Viterbi kernel, every column stored to
// HBM scratch [T,R].  Backward: beta column double-buffered in LDS, levels descending, successor
// arcs from the transposed CSR; gamma = exp(alpha + beta - logP) is produced on the fly and
// (optionally) folded into per-frame state occupancies occ[n,s] with LDS fp64 atomics.
#include "gh_internal.h"
#include "gh_fb.h"

namespace {

// streaming log-sum-exp: (m, s) with value m + log s
__device__ __forceinline__ void lse_add(double v, double& m, double& s) {
    if (v == -INFINITY) return;
    if (v <= m) {
        s += exp(v - m);
    } else {
        s = s * exp(m - v) + 1.0;
        m = v;
    }
}
__device__ __forceinline__ double lse_val(double m, double s) { return (s > 0.0) ? m + log(s) : -INFINITY; }

template <typename ET>
__global__ void fb_kernel(gh_fb_args a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, bd = blockDim.x;
    const int64_t slot = a.u_begin + blockIdx.x;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int l = a.utt_lat ? a.utt_lat[u] : 0;
    const gh_lattices::desc dsc = a.descs[l];
    const int R = dsc.R, nlev = dsc.nlev, n_end = dsc.n_end;
    const int32_t* row_state = a.row_state + dsc.row_base;
    const uint8_t* row_flag = a.row_flag + dsc.row_base;
    const int32_t* pred_ptr = a.pred_ptr + dsc.ptr_base;
    const uint32_t* pred_row = a.pred_row + dsc.arc_base;
    const double* pred_cost = a.pred_cost + dsc.arc_base;
    const int32_t* succ_ptr = a.succ_ptr + dsc.ptr_base;
    const uint32_t* succ_row = a.succ_row + dsc.arc_base;
    const double* succ_cost = a.succ_cost + dsc.arc_base;
    const int32_t* order = a.order + dsc.row_base;
    const int32_t* level_ptr = a.level_ptr + dsc.lev_base;
    const int32_t* end_rows = a.end_rows + dsc.end_base;
    const int64_t f0 = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - f0);
    const int S = a.S;
    const ET* nll = static_cast<const ET*>(a.nll) + f0 * S;
    const double NEG = -INFINITY;

    double* colA = lds;                  // alpha prev / beta next
    double* colB = lds + a.r_pad;        // alpha cur  / beta cur
    double* em0 = lds + 2 * a.r_pad;     // emissions of the current column
    double* em1 = em0 + S;               // emissions of column c+1 (backward)
    double* occ = em1 + S;               // [S] occupancy of the current column
    __shared__ double s_logp;
    double* alpha = a.alpha_scratch + a.scratch_off[slot];  // [T,R]
    double* o_alpha = a.out_alpha ? a.out_alpha + a.mat_off[u] : nullptr;  // [R,T]
    double* o_beta = a.out_beta ? a.out_beta + a.mat_off[u] : nullptr;
    double* o_gamma = a.out_gamma ? a.out_gamma + a.mat_off[u] : nullptr;

    if (T <= 0) {
        if (tid == 0 && a.logp) a.logp[u] = NEG;
        return;
    }
    for (int r = tid; r < R; r += bd) { colA[r] = NEG; colB[r] = NEG; }
    __syncthreads();
    double* prev = colA;
    double* cur = colB;
    // ------------------------------------------------------------------ forward
    for (int t = 0; t < T; ++t) {
        for (int s = tid; s < S; s += bd) em0[s] = (double)nll[(int64_t)t * S + s];
        __syncthreads();
        for (int lev = 0; lev < nlev; ++lev) {
            const int i1 = level_ptr[lev + 1];
            for (int i = level_ptr[lev] + tid; i < i1; i += bd) {
                const int r = order[i];
                const int st = row_state[r];
                const double e = st >= 0 ? em0[st] : 0.0;
                double v;
                if (t == 0 && (row_flag[r] & 1)) {
                    v = -e;
                } else {
                    double m = NEG, sm = 0.0;
                    for (int p = pred_ptr[r]; p < pred_ptr[r + 1]; ++p) {
                        const uint32_t w = pred_row[p];
                        if (w & GH_ARC_DEAD) continue;  // same-column origin not yet computed: +inf cost
                        const int o = (int)(w & GH_ARC_ROW);
                        lse_add(((w & GH_ARC_SAME) ? cur[o] : prev[o]) - pred_cost[p], m, sm);
                    }
                    v = lse_val(m, sm) - e;
                }
                cur[r] = v;
                alpha[(int64_t)t * R + r] = v;
                if (o_alpha) o_alpha[(int64_t)r * T + t] = v;
            }
            __syncthreads();
        }
        double* t_ = prev; prev = cur; cur = t_;
    }
    if (tid == 0) {  // `prev` holds the last alpha column
        double m = NEG, sm = 0.0;
        for (int k = 0; k < n_end; ++k) lse_add(prev[end_rows[k]], m, sm);
        s_logp = lse_val(m, sm);
        if (a.logp) a.logp[u] = s_logp;
    }
    __syncthreads();
    const double logp = s_logp;
    // ----------------------------------------------------------------- backward
    double* nxt = colA;  // beta of column c+1
    cur = colB;
    for (int r = tid; r < R; r += bd) { nxt[r] = NEG; cur[r] = NEG; }
    __syncthreads();
    for (int t = T - 1; t >= 0; --t) {
        for (int s = tid; s < S; s += bd) {
            em0[s] = (double)nll[(int64_t)t * S + s];
            em1[s] = (t + 1 < T) ? (double)nll[(int64_t)(t + 1) * S + s] : 0.0;
            occ[s] = 0.0;
        }
        __syncthreads();
        for (int lev = nlev - 1; lev >= 0; --lev) {
            const int i1 = level_ptr[lev + 1];
            for (int i = level_ptr[lev] + tid; i < i1; i += bd) {
                const int r = order[i];
                double m = NEG, sm = 0.0;
                if (t == T - 1 && (row_flag[r] & 2)) lse_add(0.0, m, sm);
                for (int p = succ_ptr[r]; p < succ_ptr[r + 1]; ++p) {
                    const uint32_t w = succ_row[p];
                    if (w & GH_ARC_DEAD) continue;
                    const int s2 = (int)(w & GH_ARC_ROW);
                    const int st2 = row_state[s2];
                    if (w & GH_ARC_SAME) {
                        lse_add(cur[s2] - succ_cost[p] - (st2 >= 0 ? em0[st2] : 0.0), m, sm);
                    } else if (t + 1 < T) {
                        lse_add(nxt[s2] - succ_cost[p] - (st2 >= 0 ? em1[st2] : 0.0), m, sm);
                    }
                }
                const double bv = lse_val(m, sm);
                cur[r] = bv;
                const double av = alpha[(int64_t)t * R + r];
                double g = exp(av + bv - logp);
                if (!(g == g)) g = 0.0;  // -inf - -inf, or logP == -inf
                if (o_beta) o_beta[(int64_t)r * T + t] = bv;
                if (o_gamma) o_gamma[(int64_t)r * T + t] = g;
                const int st = row_state[r];
                if (a.occ && st >= 0 && g != 0.0) atomicAdd(&occ[st], g);
            }
            __syncthreads();
        }
        if (a.occ) {
            double* orow = a.occ + (f0 + t) * S;
            for (int s = tid; s < S; s += bd) orow[s] = occ[s];
        }
        __syncthreads();
        double* t_ = nxt; nxt = cur; cur = t_;
        for (int r = tid; r < R; r += bd) cur[r] = NEG;  // same-column reads see only rows of this column
        __syncthreads();
    }
}

}  // namespace

int gh_launch_fb(gh_ctx* ctx, const gh_fb_args& a, int64_t n_utts, int block, size_t lds_bytes, bool f64) {
    if (n_utts <= 0) return GH_OK;
    if (f64) hipLaunchKernelGGL((fb_kernel<double>), dim3((unsigned)n_utts), dim3((unsigned)block), lds_bytes, ctx->stream, a);
    else hipLaunchKernelGGL((fb_kernel<float>), dim3((unsigned)n_utts), dim3((unsigned)block), lds_bytes, ctx->stream, a);
    GH_HIP(hipGetLastError());
    return GH_OK;
}
