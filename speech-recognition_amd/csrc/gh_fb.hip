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
#include "gh_xnum.h"

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

// workgroup barrier that orders LDS traffic only (the alpha / occupancy stores and the prefetch loads stay in flight)
__device__ __forceinline__ void fb_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The graph (both CSRs, row tables, level offsets) is staged into LDS once per workgroup and the next column's
// emissions / alpha column travel from HBM while the current column is processed: walking the CSR and fetching the
// emissions from global memory inside the column loop cost 2-3 memory round trips per level (2.6 us per column).
template <typename ET>
__global__ void fb_kernel(gh_fb_args a) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, bd = blockDim.x;
    const int64_t slot = a.u_begin + blockIdx.x;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int l = a.utt_lat ? a.utt_lat[u] : 0;
    const gh_lattices::desc dsc = a.descs[l];
    const int R = dsc.R, nlev = dsc.nlev, n_end = dsc.n_end;
    const int32_t* end_rows = a.end_rows + dsc.end_base;
    const int64_t f0 = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - f0);
    const int S = a.S;
    const ET* nll = static_cast<const ET*>(a.nll) + f0 * S;
    const double NEG = -INFINITY;

    double* colA = lds;                  // alpha prev / beta next
    double* colB = lds + a.r_pad;        // alpha cur  / beta cur
    double* emA = lds + 2 * a.r_pad;     // emission vectors (rotating)
    double* emB = emA + S;
    double* emC = emB + S;
    double* occ = emC + S;               // [S] occupancy of the current column
    double* xis = occ + S;               // [S] expected self transitions of this utterance
    double* g_pcost = xis + S;           // [arc_cap]
    double* g_scost = g_pcost + a.arc_cap;
    int32_t* g_prow = reinterpret_cast<int32_t*>(g_scost + a.arc_cap);   // [arc_cap]
    int32_t* g_srow = g_prow + a.arc_cap;
    int32_t* g_pptr = g_srow + a.arc_cap;     // [R + 1]
    int32_t* g_sptr = g_pptr + a.r_pad + 2;
    int32_t* g_order = g_sptr + a.r_pad + 2;  // [R]
    int32_t* g_state = g_order + a.r_pad;
    int32_t* g_flag = g_state + a.r_pad;
    int32_t* g_lev = g_flag + a.r_pad;        // [nlev + 1]
    __shared__ double s_logp;
    double* alpha = a.alpha_scratch + a.scratch_off[slot];  // [T,R]
    double* o_alpha = a.out_alpha ? a.out_alpha + a.mat_off[u] : nullptr;  // [R,T]
    double* o_beta = a.out_beta ? a.out_beta + a.mat_off[u] : nullptr;
    double* o_gamma = a.out_gamma ? a.out_gamma + a.mat_off[u] : nullptr;

    if (T <= 0) {
        if (tid == 0 && a.logp) a.logp[u] = NEG;
        return;
    }
    {   // ---- graph -> LDS
        const int32_t* pred_ptr = a.pred_ptr + dsc.ptr_base;
        const int32_t* succ_ptr = a.succ_ptr + dsc.ptr_base;
        const int n_arcs = pred_ptr[R];
        for (int p = tid; p < n_arcs; p += bd) {
            g_pcost[p] = a.pred_cost[dsc.arc_base + p]; g_prow[p] = (int32_t)a.pred_row[dsc.arc_base + p];
            g_scost[p] = a.succ_cost[dsc.arc_base + p]; g_srow[p] = (int32_t)a.succ_row[dsc.arc_base + p];
        }
        for (int r = tid; r <= R; r += bd) { g_pptr[r] = pred_ptr[r]; g_sptr[r] = succ_ptr[r]; }
        for (int r = tid; r < R; r += bd) {
            g_order[r] = a.order[dsc.row_base + r];
            g_state[r] = a.row_state[dsc.row_base + r];
            g_flag[r] = a.row_flag[dsc.row_base + r];
            colA[r] = NEG; colB[r] = NEG;
        }
        for (int k = tid; k <= nlev; k += bd) g_lev[k] = a.level_ptr[dsc.lev_base + k];
    }
    // emissions: every lane owns the states tid, tid + bd, ... (at most FB_EM per lane; larger S falls back to a loop)
    constexpr int FB_EM = 4;
    const bool em_regs = S <= FB_EM * bd;
    ET epre[FB_EM];
    auto em_prefetch = [&](int t) {   // column t -> registers (issued early, parked later)
        if (!em_regs) return;
#pragma unroll
        for (int k = 0; k < FB_EM; ++k) {
            const int s = tid + k * bd;
            epre[k] = (t >= 0 && t < T && s < S) ? nll[(int64_t)t * S + s] : ET(0);
        }
    };
    auto em_park = [&](double* dst, int t) {
        if (em_regs) {
#pragma unroll
            for (int k = 0; k < FB_EM; ++k) { const int s = tid + k * bd; if (s < S) dst[s] = (double)epre[k]; }
        } else {
            for (int s = tid; s < S; s += bd) dst[s] = (t >= 0 && t < T) ? (double)nll[(int64_t)t * S + s] : 0.0;
        }
    };
    em_prefetch(0);
    em_park(emA, 0);
    __syncthreads();
    double* prev = colA;
    double* cur = colB;
    double* em0 = emA;     // emissions of the current column
    double* emn = emB;     // being filled for the next one
    // ------------------------------------------------------------------ forward
    for (int t = 0; t < T; ++t) {
        em_prefetch(t + 1);
        for (int lev = 0; lev < nlev; ++lev) {
            const int i1 = g_lev[lev + 1];
            for (int i = g_lev[lev] + tid; i < i1; i += bd) {
                const int r = g_order[i];
                const int st = g_state[r];
                const double e = st >= 0 ? em0[st] : 0.0;
                double v;
                if (t == 0 && (g_flag[r] & 1)) {
                    v = -e;
                } else {
                    double m = NEG, sm = 0.0;
                    for (int p = g_pptr[r]; p < g_pptr[r + 1]; ++p) {
                        const uint32_t w = (uint32_t)g_prow[p];
                        if (w & GH_ARC_DEAD) continue;  // same-column origin not yet computed: +inf cost
                        const int o = (int)(w & GH_ARC_ROW);
                        lse_add(((w & GH_ARC_SAME) ? cur[o] : prev[o]) - g_pcost[p], m, sm);
                    }
                    v = lse_val(m, sm) - e;
                }
                cur[r] = v;
                alpha[(int64_t)t * R + r] = v;
                if (o_alpha) o_alpha[(int64_t)r * T + t] = v;
            }
            fb_lds_barrier();
        }
        em_park(emn, t + 1);
        fb_lds_barrier();
        { double* t_ = prev; prev = cur; cur = t_; }
        { double* t_ = em0; em0 = emn; emn = t_; }
    }
    if (tid == 0) {  // `prev` holds the last alpha column
        double m = NEG, sm = 0.0;
        for (int k = 0; k < n_end; ++k) lse_add(prev[end_rows[k]], m, sm);
        s_logp = lse_val(m, sm);
        if (a.logp) a.logp[u] = s_logp;
    }
    __syncthreads();   // (full barrier: the alpha columns written above are read back below)
    const double logp = s_logp;
    // ----------------------------------------------------------------- backward
    double* nxt = colA;  // beta of column c+1
    cur = colB;
    for (int r = tid; r < R; r += bd) { nxt[r] = NEG; cur[r] = NEG; }
    // emission window: em0 = column t, em1 = column t + 1, emn = being filled with column t - 1
    double* em1 = emC;
    em0 = emA; emn = emB;
    em_prefetch(T - 1);
    em_park(em0, T - 1);
    for (int s = tid; s < S; s += bd) { em1[s] = 0.0; xis[s] = 0.0; }
    // alpha column of every row this lane owns (rows i = tid, tid + bd, ... of the level order), one column ahead
    constexpr int FB_AL = 4;
    const bool al_regs = R <= FB_AL * bd;
    double apre[FB_AL];
    auto alpha_prefetch = [&](int t) {
        if (!al_regs) return;
#pragma unroll
        for (int k = 0; k < FB_AL; ++k) {
            const int i = tid + k * bd;
            apre[k] = (t >= 0 && i < R) ? alpha[(int64_t)t * R + g_order[i]] : 0.0;
        }
    };
    __syncthreads();
    alpha_prefetch(T - 1);
    for (int t = T - 1; t >= 0; --t) {
        double acur[FB_AL];
#pragma unroll
        for (int k = 0; k < FB_AL; ++k) acur[k] = apre[k];
        alpha_prefetch(t - 1);
        em_prefetch(t - 1);
        for (int s = tid; s < S; s += bd) occ[s] = 0.0;
        fb_lds_barrier();
        auto beta_row = [&](int i, double av) {   // row i of the level order; av = its alpha in this column
            const int r = g_order[i];
            double m = NEG, sm = 0.0;
            if (t == T - 1 && (g_flag[r] & 2)) lse_add(0.0, m, sm);
            for (int p = g_sptr[r]; p < g_sptr[r + 1]; ++p) {
                const uint32_t w = (uint32_t)g_srow[p];
                if (w & GH_ARC_DEAD) continue;
                const int s2 = (int)(w & GH_ARC_ROW);
                const int st2 = g_state[s2];
                if (w & GH_ARC_SAME) {
                    lse_add(cur[s2] - g_scost[p] - (st2 >= 0 ? em0[st2] : 0.0), m, sm);
                } else if (t + 1 < T) {
                    const double term = nxt[s2] - g_scost[p] - (st2 >= 0 ? em1[st2] : 0.0);
                    lse_add(term, m, sm);
                    if (a.self_xi && s2 == r && st2 >= 0) {      // xi_t(r -> r): alpha_t(r) a_rr b_r(x_{t+1}) beta_{t+1}(r) / P
                        const double x = exp(av + term - logp);
                        if (x == x && x != 0.0) atomicAdd(&xis[st2], x);
                    }
                }
            }
            const double bv = lse_val(m, sm);
            cur[r] = bv;
            double g = exp(av + bv - logp);
            if (!(g == g)) g = 0.0;  // -inf - -inf, or logP == -inf
            if (o_beta) o_beta[(int64_t)r * T + t] = bv;
            if (o_gamma) o_gamma[(int64_t)r * T + t] = g;
            const int st = g_state[r];
            if (a.occ && st >= 0 && g != 0.0) atomicAdd(&occ[st], g);
        };
        for (int lev = nlev - 1; lev >= 0; --lev) {
            const int i0 = g_lev[lev], i1 = g_lev[lev + 1];
            if (al_regs) {
#pragma unroll
                for (int k = 0; k < FB_AL; ++k) {
                    const int i = tid + k * bd;
                    if (i >= i0 && i < i1) beta_row(i, acur[k]);
                }
            } else {
                for (int i = i0 + tid; i < i1; i += bd) beta_row(i, alpha[(int64_t)t * R + g_order[i]]);
            }
            fb_lds_barrier();
        }
        if (a.occ) {
            double* orow = a.occ + (f0 + t) * S;
            for (int s = tid; s < S; s += bd) orow[s] = occ[s];
        }
        em_park(emn, t - 1);
        { double* t_ = nxt; nxt = cur; cur = t_; }
        { double* t_ = em1; em1 = em0; em0 = emn; emn = t_; }   // window slides down: (t, t+1) -> (t-1, t)
        fb_lds_barrier();
        for (int r = tid; r < R; r += bd) cur[r] = NEG;  // same-column reads see only rows of this column
    }
    if (a.self_xi) {
        __syncthreads();
        for (int s = tid; s < S; s += bd) if (xis[s] != 0.0) atomicAdd(a.self_xi + s, xis[s]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// One-word chain graphs (gh_fbchain): ONE LANE PER UTTERANCE.  The alpha / beta vectors of the <= 8 states live in
// registers, the recursion needs no LDS, no shuffles and no barriers; a wave advances 64 utterances (sorted by
// length) one column per iteration.  Against the generic kernel (one workgroup per utterance, 5 of 64 lanes busy,
// three barriers per column) this cuts the instruction stream per utterance and column by ~two orders of
// magnitude.  Same definition as fb_kernel: log domain, alpha_0 = -c0 - e, end = last row.
__device__ __forceinline__ double lse3(double x, double y, double z) {
    const double m = fmax(fmax(x, y), z);
    if (m == -INFINITY) return -INFINITY;
    return m + log(exp(x - m) + exp(y - m) + exp(z - m));
}

__device__ __forceinline__ double lse2(double x, double y) {
    const double m = fmax(x, y);
    if (m == -INFINITY) return -INFINITY;
    return m + log1p(exp(fmin(x, y) - m));   // one exp + one log1p (three exps + a log in lse3)
}

// Lane layout: NL = 8 (or 16, for chains of 9 .. 16 rows) lanes per utterance (lane j of the group = chain row j, idle
// when j >= n), 64 / NL utterances per wave.
// A first version with one lane per utterance and all rows in registers had U/64 waves of ~800 dependent
// instructions per column (0.85 ms for 12 500 utterances); one row per lane gives 8x the waves and 1/5 of the
// serial work per lane.  Neighbouring rows are one DPP row shift away (the 8-lane groups sit inside 16-lane DPP rows;
// what a shift drags in from the neighbouring group is masked by the j >= 1 / j >= 2 tests).
// The recursion runs on probabilities with an extended exponent (gh_xnum.h), like fb_seq_kernel: a term is a multiply
// and an integer add, the one transcendental per cell is the split exponential of the emission cost (the log-domain
// version -- exp per term, log per sum -- took 0.34 ms per 12 500 utterances, compute bound at 8 % of HBM).
template <int CTRL> __device__ __forceinline__ xnum fbc_dpp(xnum v) {
    xnum o;
    o.f = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v.f), CTRL, 0xF, 0xF, true),
                           __builtin_amdgcn_update_dpp(0, __double2loint(v.f), CTRL, 0xF, 0xF, true));
    o.e = __builtin_amdgcn_update_dpp(XN_ZERO_E, v.e, CTRL, 0xF, 0xF, false);
    return o;
}

template <typename ET, int NL>
__global__ __launch_bounds__(64) void fb_chain_kernel(gh_fbchain_args a) {
    constexpr int NMAX = GH_FBCHAIN_MAX;   // capacity of the chain arrays
    constexpr int PD = 4;                  // columns of loads in flight per lane (8: no change)
    const int j = threadIdx.x & (NL - 1);
    const int64_t slot = (int64_t)blockIdx.x * (64 / NL) + (threadIdx.x / NL);
    const bool has_utt = slot < a.U;
    const int64_t u = has_utt ? (a.perm ? a.perm[slot] : slot) : 0;
    const gh_fbchain* chp = a.chains + ((has_utt && a.utt_lat) ? a.utt_lat[u] : 0);
    const int n = chp->n;
    const bool act = has_utt && j < n;     // this lane owns a row
    const bool skip = chp->pad != 0;
    const int st = act ? chp->state[j] : 0;
    const double INF = INFINITY;
    // arc probabilities exp(-cost): into this row from j, j-1, j-2, and out of it towards j+1 / j+2 (stored at their
    // destination); absent arcs are 0
    const xnum p_self = xn_exp_neg(act ? chp->self_c[j] : INF);
    const xnum p_next = xn_exp_neg((act && j >= 1) ? chp->next_c[j] : INF);
    const xnum p_skip = xn_exp_neg((act && j >= 2 && skip) ? chp->skip_c[j] : INF);
    const xnum p_out1 = xn_exp_neg((act && j + 1 < n) ? chp->next_c[(j + 1 < NMAX) ? j + 1 : j] : INF);
    const xnum p_out2 = xn_exp_neg((act && j + 2 < n && skip) ? chp->skip_c[(j + 2 < NMAX) ? j + 2 : j] : INF);
    const int64_t f0 = has_utt ? a.utt_off[u] : 0;
    const int T = has_utt ? (int)(a.utt_off[u + 1] - f0) : 0;
    // every lane of the wave runs to the longest utterance of the wave (cross-lane operations need converged lanes)
    int Tmax = T;
#pragma unroll
    for (int o = 32; o >= NL; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o));
    if (has_utt && j == 0 && T <= 0 && a.logp) a.logp[u] = -INF;
    const ET* nll = static_cast<const ET*>(a.nll) + f0 * a.S;
    const int64_t cells = (int64_t)T * n;
    double* alf = a.alpha_scratch + (has_utt ? a.scratch_off[slot] : 0);   // [T, n] mantissas, then [T, n] exponents
    int* ale = reinterpret_cast<int*>(alf + cells);
    auto load_e = [&](int t) -> ET { return (act && t >= 0 && t < T) ? nll[(int64_t)t * a.S + st] : ET(0); };
    ET ering[PD];
#pragma unroll
    for (int k = 0; k < PD; ++k) ering[k] = load_e(1 + k);
    // ---- forward ----
    xnum al = (act && j == 0 && T > 0) ? xn_norm(xn_mul(xn_exp_neg(chp->c0), xn_exp_neg((double)load_e(0)))) : xn_zero();
    xnum al_last = xn_zero();   // alpha of this row in column T - 1
    for (int t0 = 0; t0 < Tmax; t0 += PD) {
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            const int t = t0 + k;
            if (t >= Tmax) break;
            if (act && t < T) { alf[(int64_t)t * n + j] = al.f; ale[(int64_t)t * n + j] = al.e; }
            if (t == T - 1) al_last = al;
            xnum up1 = fbc_dpp<0x111>(al), up2 = fbc_dpp<0x112>(al);            // rows j - 1, j - 2 (row_shr)
            if (j < 1) up1 = xn_zero();                                          // (what came from the neighbouring group)
            if (j < 2) up2 = xn_zero();
            const xnum bn = xn_exp_neg((double)ering[k]);                        // emission probability of column t + 1
            const xnum sum = skip ? xn_add3(xn_mul(al, p_self), xn_mul(up1, p_next), xn_mul(up2, p_skip))
                                  : xn_add(xn_mul(al, p_self), xn_mul(up1, p_next));
            const xnum nx = xn_norm(xn_mul(sum, bn));
            if (act && t + 1 < T) al = nx;
            ering[k] = load_e(t + 1 + PD);   // slot k next serves column t + 1 + PD
        }
    }
    xnum P;
    P.f = __shfl(al_last.f, (n > 0 ? n - 1 : 0), NL);
    P.e = __shfl(al_last.e, (n > 0 ? n - 1 : 0), NL);
    const double logp = xn_log(P);
    if (has_utt && j == 0 && T > 0 && a.logp) a.logp[u] = logp;
    if (!a.occ && !a.gam && !a.self_xi_utt) return;
    const bool reach = P.f > 0.0;              // log P = -inf: every gamma is 0
    const double inv_pf = 1.0 / P.f;
    // ---- backward: beta in a register, gamma straight into the occupancy rows ----
    auto load_af = [&](int t) -> double { return (act && t >= 0 && t < T) ? alf[(int64_t)t * n + j] : 0.0; };
    auto load_ae = [&](int t) -> int { return (act && t >= 0 && t < T) ? ale[(int64_t)t * n + j] : XN_ZERO_E; };
    xnum be = (act && j == n - 1) ? xn_one() : xn_zero();
    double xi_acc = 0.0;                      // expected self transitions of this row (when asked for)
    int r_lo = T, r_hi = -1;                  // frames of this row with gamma above the floor (when asked for)
    const bool want_rng = a.occ_rng != nullptr;
    const bool want_xi = a.self_xi_utt != nullptr;
    xnum ap = xnum{load_af(T - 1), load_ae(T - 1)};
    double e = (double)load_e(T - 1);
    double arf[PD];
    int are[PD];
#pragma unroll
    for (int k = 0; k < PD; ++k) { arf[k] = load_af(T - 2 - k); are[k] = load_ae(T - 2 - k); ering[k] = load_e(T - 2 - k); }
    // lanes of shorter utterances idle (be = 0, no stores) until the wave's column index reaches their T - 1
    for (int t0 = Tmax - 1; t0 >= 0; t0 -= PD) {
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            const int tw = t0 - k;              // column index of the wave's longest utterance
            if (tw < 0) break;
            const int t = tw - (Tmax - T);      // this utterance's column (negative: it has not started yet)
            if ((a.occ || a.gam) && has_utt && t >= 0) {
                const double g = (act && reach) ? xn_ratio(ap, be, inv_pf, P.e) : 0.0;
                if (a.gam) a.gam[(f0 + t) * NL + j] = g;              // all NL columns (rows >= n: 0): one 64- / 128-byte line per frame
                else if (act) a.occ[(f0 + t) * a.S + st] = g;
                if (want_rng && ((g > a.rng_floor) | (g != g))) { r_lo = t; if (r_hi < 0) r_hi = t; }   // (t runs downwards)
            }
            // beta_{t-1}(j) = sum over successors s = j, j+1, j+2 of beta_t(s) b_s(x_t) a_{j -> s}
            const xnum w = (act && t >= 0) ? xn_mul(be, xn_exp_neg(e)) : xn_zero();
            xnum d1 = fbc_dpp<0x101>(w), d2 = fbc_dpp<0x102>(w);                // rows j + 1, j + 2 (row_shl)
            if (j + 1 >= n) d1 = xn_zero();
            if (j + 2 >= n) d2 = xn_zero();
            const xnum ws = xn_mul(w, p_self);
            const xnum nb = xn_norm(skip ? xn_add3(ws, xn_mul(d1, p_out1), xn_mul(d2, p_out2)) : xn_add(ws, xn_mul(d1, p_out1)));
            const xnum aprev = xnum{arf[k], are[k]};
            if (want_xi && act && reach && t > 0) xi_acc += xn_ratio(aprev, ws, inv_pf, P.e);   // xi_t(j -> j) = alpha_{t-1}(j) a_jj b_j(x_t) beta_t(j) / P
            if (act && t > 0) { be = nb; ap = aprev; e = (double)ering[k]; }
            if (t >= 0) { arf[k] = load_af(t - 1 - PD); are[k] = load_ae(t - 1 - PD); ering[k] = load_e(t - 1 - PD); }
        }
    }
    if (want_xi && has_utt) a.self_xi_utt[u * NMAX + j] = act ? xi_acc : 0.0;
    if (want_rng && has_utt) { a.occ_rng[(u * NMAX + j) * 2] = r_lo; a.occ_rng[(u * NMAX + j) * 2 + 1] = r_hi; }
}

// ---------------------------------------------------------------------------------------------------------
// Two-way form.  fb_chain_kernel is latency bound on small batches: a wave walks its utterances' columns twice (forward,
// then backward: ~300 dependent steps of ~2 000 cycles for a 150-frame utterance), and 12 500 utterances are only 1.5 waves
// per SIMD.  The backward recursion on w_t(j) = beta_t(j) b_j(x_t),
//     w_{t-1}(j) = b_j(x_{t-1}) [ w_t(j) a_jj + w_t(j+1) a_{j,j+1} + w_t(j+2) a_{j,j+2} ],
// is the forward recursion with time and chain reversed, and the bracket is beta_{t-1}(j).  So an utterance gets 2 NL
// lanes: the first NL run alpha over rows 0 .. n-1 from the first frame, the other NL run w over rows n-1 .. 0 from the
// last frame -- the SAME instructions, half the dependent steps, twice the waves.  Both store their column (alpha_t, resp.
// the bracket beta_t) to scratch; once both are through, the group's lanes sweep the (t, row) cells in parallel:
// gamma = alpha beta / P (compact, all NL columns), xi_t(j -> j) = alpha_{t-1}(j) a_jj b_j(x_t) beta_t(j) / P summed per
// row, the rows' occupancy ranges, log P.  Same definitions as fb_chain_kernel; results agree to rounding (beta_t is kept
// as the unnormalised bracket, gamma is formed in a different order).
template <typename ET, int NL>
__global__ __launch_bounds__(64) void fb_chain2_kernel(gh_fbchain_args a) {
    constexpr int G = 2 * NL;              // lanes per utterance
    constexpr int NMAX = GH_FBCHAIN_MAX;
    constexpr int PD = 4;                  // columns of emission loads in flight per lane
    const int lane = threadIdx.x;
    const int k = lane & (NL - 1);         // position inside the direction's lane group
    const int dir = (lane / NL) & 1;       // 0: forward (alpha), 1: backward (w)
    const int64_t slot = (int64_t)blockIdx.x * (64 / G) + lane / G;
    const bool has_utt = slot < a.U;
    const int64_t u = has_utt ? (a.perm ? a.perm[slot] : slot) : 0;
    const gh_fbchain* chp = a.chains + ((has_utt && a.utt_lat) ? a.utt_lat[u] : 0);
    const int n = chp->n;
    const bool act = has_utt && k < n;
    const bool skip = chp->pad != 0;
    const int j = act ? (dir ? n - 1 - k : k) : 0;            // chain row of this lane in the recursion
    const int st = act ? chp->state[j] : 0;
    const double INF = INFINITY;
    // arc probabilities into this lane's cell from lane k (same row), k - 1, k - 2 of its direction:
    //   forward: rows j, j-1, j-2 -> j;  backward: successors j, j+1, j+2 of row j (arcs are stored at their destination)
    const xnum p_a = xn_exp_neg(act ? chp->self_c[j] : INF);
    const xnum p_b = xn_exp_neg((act && k >= 1) ? (dir ? chp->next_c[j + 1] : chp->next_c[j]) : INF);
    const xnum p_c = xn_exp_neg((act && k >= 2 && skip) ? (dir ? chp->skip_c[j + 2] : chp->skip_c[j]) : INF);
    const int64_t f0 = has_utt ? a.utt_off[u] : 0;
    const int T = has_utt ? (int)(a.utt_off[u + 1] - f0) : 0;
    int Tmax = T;
#pragma unroll
    for (int o = 32; o >= G; o >>= 1) Tmax = max(Tmax, __shfl_xor(Tmax, o));
    if (has_utt && lane % G == 0 && T <= 0 && a.logp) a.logp[u] = -INF;
    const ET* nll = static_cast<const ET*>(a.nll) + f0 * a.S;
    const int64_t cells = (int64_t)T * n;
    double* base = a.alpha_scratch + (has_utt ? a.scratch_off[slot] : 0);
    double* af = base;                                         // alpha mantissas [T, n]
    double* bf = base + cells;                                 // beta mantissas
    int* ae = reinterpret_cast<int*>(base + 2 * cells);        // alpha exponents, then beta exponents
    int* be = ae + cells;
    double* myf = dir ? bf : af;
    int* mye = dir ? be : ae;
    auto col = [&](int i) -> int { return dir ? T - 1 - i : i; };   // frame of this direction's step i
    auto load_e = [&](int i) -> ET { return (act && i >= 0 && i < T) ? nll[(int64_t)col(i) * a.S + st] : ET(0); };
    ET ering[PD];
#pragma unroll
    for (int q = 0; q < PD; ++q) ering[q] = load_e(1 + q);
    // step 0: alpha_0 = [row 0] exp(-c0) b(x_0);   w_{T-1} = [row n-1] b(x_{T-1}), beta_{T-1} = [row n-1]
    xnum v = xn_zero(), sv = xn_zero();
    if (act && k == 0 && T > 0) {
        v = xn_norm(xn_mul(dir ? xn_one() : xn_exp_neg(chp->c0), xn_exp_neg((double)load_e(0))));
        sv = dir ? xn_one() : v;
    }
    xnum v_last = xn_zero();
    for (int i0 = 0; i0 < Tmax; i0 += PD) {
#pragma unroll
        for (int q = 0; q < PD; ++q) {
            const int i = i0 + q;
            if (i >= Tmax) break;
            if (act && i < T) { const int64_t at = (int64_t)col(i) * n + j; myf[at] = sv.f; mye[at] = sv.e; }
            if (i == T - 1) v_last = v;
            xnum up1 = fbc_dpp<0x111>(v), up2 = fbc_dpp<0x112>(v);              // lanes k - 1, k - 2 (row_shr)
            if (k < 1) up1 = xn_zero();                                          // (what came from the neighbouring group)
            if (k < 2) up2 = xn_zero();
            const xnum bn = xn_exp_neg((double)ering[q]);                        // emission probability of step i + 1
            const xnum s = skip ? xn_add3(xn_mul(v, p_a), xn_mul(up1, p_b), xn_mul(up2, p_c))
                                : xn_add(xn_mul(v, p_a), xn_mul(up1, p_b));
            const xnum nx = xn_norm(xn_mul(s, bn));
            if (act && i + 1 < T) { v = nx; sv = dir ? s : nx; }                 // backward: the bracket is beta of step i + 1
            ering[q] = load_e(i + 1 + PD);
        }
    }
    // P = alpha_{T-1}(n - 1): forward lane n - 1 of the group; left behind the columns for the cell kernel
    xnum P;
    P.f = __shfl(v_last.f, (n > 0 ? n - 1 : 0), G);
    P.e = __shfl(v_last.e, (n > 0 ? n - 1 : 0), G);
    if (has_utt && lane % G == 0 && T > 0) {
        base[3 * cells] = P.f;
        base[3 * cells + 1] = (double)P.e;
        if (a.logp) a.logp[u] = xn_log(P);
    }
}

// The cells of the two-way form, one wave per utterance, lane = (frame mod 64 / NL, row): gamma = alpha beta / P written
// as whole lines of the compact matrix, xi_t(j -> j) = alpha_{t-1}(j) a_jj b_j(x_t) beta_t(j) / P summed per row, the rows'
// occupancy ranges.  (Inside the recursion kernel -- the group's lanes sweeping their utterance after the last column --
// this cost as much as the recursion had saved: a latency-bound loop of T / 2 dependent round trips per lane.)
template <typename ET, int NL>
__global__ __launch_bounds__(64) void fb_chain2_cells_kernel(gh_fbchain_args a) {
    constexpr int NMAX = GH_FBCHAIN_MAX;
    constexpr int FPI = 64 / NL;           // frames per sweep of the wave
    constexpr int UN = 4;
    const int64_t slot = blockIdx.x;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const gh_fbchain* chp = a.chains + (a.utt_lat ? a.utt_lat[u] : 0);
    const int n = chp->n;
    const int64_t f0 = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - f0);
    const int k = threadIdx.x & (NL - 1), tl = threadIdx.x / NL;
    const bool want_xi = a.self_xi_utt != nullptr;
    const bool want_rng = a.occ_rng != nullptr;
    if (T <= 0) {
        if (tl == 0) {
            if (want_xi) a.self_xi_utt[u * NMAX + k] = 0.0;
            if (want_rng) { a.occ_rng[(u * NMAX + k) * 2] = 0; a.occ_rng[(u * NMAX + k) * 2 + 1] = -1; }
        }
        return;
    }
    const ET* nll = static_cast<const ET*>(a.nll) + f0 * a.S;
    const int64_t cells = (int64_t)T * n;
    const double* base = a.alpha_scratch + a.scratch_off[slot];
    const double* af = base;
    const double* bf = base + cells;
    const int* ae = reinterpret_cast<const int*>(base + 2 * cells);
    const int* be = ae + cells;
    xnum P;
    P.f = base[3 * cells];
    P.e = (int)base[3 * cells + 1];
    const bool reach = P.f > 0.0;
    const double inv_pf = 1.0 / P.f;
    const bool row = k < n;
    const int st = row ? chp->state[k] : 0;
    const xnum p_self = xn_exp_neg(row ? chp->self_c[k] : INFINITY);
    double xi_acc = 0.0;
    int r_lo = T, r_hi = -1;
    for (int t0 = tl; t0 < T; t0 += FPI * UN) {
        double xaf[UN], xbf[UN], xpf[UN];
        int xae[UN], xbe[UN], xpe[UN];
        ET xe[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int t = t0 + FPI * q;
            const bool in = row && t < T;
            const int64_t at = (int64_t)t * n + k;
            xaf[q] = in ? af[at] : 0.0; xae[q] = in ? ae[at] : XN_ZERO_E;
            xbf[q] = in ? bf[at] : 0.0; xbe[q] = in ? be[at] : XN_ZERO_E;
            const bool inp = in && t >= 1 && want_xi;
            xpf[q] = inp ? af[at - n] : 0.0; xpe[q] = inp ? ae[at - n] : XN_ZERO_E;
            xe[q] = inp ? nll[(int64_t)t * a.S + st] : ET(0);
        }
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int t = t0 + FPI * q;
            if (t >= T) break;
            const xnum A = xnum{xaf[q], xae[q]}, B = xnum{xbf[q], xbe[q]};
            const double g = (row && reach) ? xn_ratio(A, B, inv_pf, P.e) : 0.0;
            a.gam[(f0 + t) * NL + k] = g;                     // all NL columns (rows >= n: 0): whole lines per sweep
            if (want_rng && ((g > a.rng_floor) | (g != g))) { r_hi = t; if (r_lo > t) r_lo = t; }   // (t runs upwards)
            if (want_xi && row && reach && t >= 1) {
                const xnum wt = xn_mul(B, xn_exp_neg((double)xe[q]));
                xi_acc += xn_ratio(xnum{xpf[q], xpe[q]}, xn_mul(wt, p_self), inv_pf, P.e);
            }
        }
    }
    // a row's lanes: k, k + NL, k + 2 NL, ...
#pragma unroll
    for (int o = NL; o < 64; o <<= 1) {
        xi_acc += __shfl_xor(xi_acc, o);
        r_lo = min(r_lo, __shfl_xor(r_lo, o));
        r_hi = max(r_hi, __shfl_xor(r_hi, o));
    }
    if (tl == 0) {
        if (want_xi) a.self_xi_utt[u * NMAX + k] = row ? xi_acc : 0.0;
        if (want_rng) { a.occ_rng[(u * NMAX + k) * 2] = r_lo; a.occ_rng[(u * NMAX + k) * 2 + 1] = r_hi; }
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

// compact gamma and nothing per state: the two-way form (GMMHMM_FBCHAIN=1 keeps the one-way kernel)
// The two-way form halves the dependent steps of a wave and doubles the waves, at the price of a second pass over the
// columns (beta is stored too; gamma / xi come from a separate kernel): it pays while the one-way waves are too few to hide
// their own latency -- 12 500 five-state utterances = 1.5 waves per SIMD: 0.257 -> 0.129 + 0.072 ms -- and costs a little once
// the one-way kernel is throughput bound (50 000 sixteen-state utterances = 12 waves per SIMD: 0.87 -> 1.2 ms).
// GMMHMM_FBCHAIN=1 / 2 forces the one-way / two-way form.
bool gh_fbchain_two_way(const gh_ctx* ctx, bool compact_gamma, bool occupancy_matrix, int64_t U, int lanes) {
    if (!compact_gamma || occupancy_matrix) return false;
    if (const char* e = getenv("GMMHMM_FBCHAIN")) { if (atoi(e) == 1) return false; if (atoi(e) == 2) return true; }
    const int64_t one_way_waves = (U * lanes + 63) / 64;
    return one_way_waves < (int64_t)4 * 4 * ctx->n_cu;        // fewer than 4 waves per SIMD
}

int gh_launch_fb_chain(gh_ctx* ctx, const gh_fbchain_args& a, bool f64) {
    if (a.U <= 0) return GH_OK;
    if (a.lanes != 8 && a.lanes != 16) { gh_set_error("gh_launch_fb_chain: internal: %d lanes per utterance", a.lanes); return GH_ERR_INVALID; }
    if (gh_fbchain_two_way(ctx, a.gam != nullptr, a.occ != nullptr, a.U, a.lanes)) {
        const int per_wave2 = 32 / a.lanes;                // 4 (or 2) utterances per wave, 2 x 8 (or 2 x 16) lanes each
        const dim3 grid2((unsigned)((a.U + per_wave2 - 1) / per_wave2)), blk2(64);
        if (a.lanes == 8) {
            if (f64) hipLaunchKernelGGL((fb_chain2_kernel<double, 8>), grid2, blk2, 0, ctx->stream, a);
            else hipLaunchKernelGGL((fb_chain2_kernel<float, 8>), grid2, blk2, 0, ctx->stream, a);
        } else {
            if (f64) hipLaunchKernelGGL((fb_chain2_kernel<double, 16>), grid2, blk2, 0, ctx->stream, a);
            else hipLaunchKernelGGL((fb_chain2_kernel<float, 16>), grid2, blk2, 0, ctx->stream, a);
        }
        const dim3 grid3((unsigned)a.U);
        if (a.lanes == 8) {
            if (f64) hipLaunchKernelGGL((fb_chain2_cells_kernel<double, 8>), grid3, blk2, 0, ctx->stream, a);
            else hipLaunchKernelGGL((fb_chain2_cells_kernel<float, 8>), grid3, blk2, 0, ctx->stream, a);
        } else {
            if (f64) hipLaunchKernelGGL((fb_chain2_cells_kernel<double, 16>), grid3, blk2, 0, ctx->stream, a);
            else hipLaunchKernelGGL((fb_chain2_cells_kernel<float, 16>), grid3, blk2, 0, ctx->stream, a);
        }
        GH_HIP(hipGetLastError());
        return GH_OK;
    }
    const int per_wave = 64 / a.lanes;                     // 8 (or 4) utterances per wave, 8 (or 16) lanes each
    const dim3 grid((unsigned)((a.U + per_wave - 1) / per_wave)), blk(64);
    if (a.lanes == 8) {
        if (f64) hipLaunchKernelGGL((fb_chain_kernel<double, 8>), grid, blk, 0, ctx->stream, a);
        else hipLaunchKernelGGL((fb_chain_kernel<float, 8>), grid, blk, 0, ctx->stream, a);
    } else {
        if (f64) hipLaunchKernelGGL((fb_chain_kernel<double, 16>), grid, blk, 0, ctx->stream, a);
        else hipLaunchKernelGGL((fb_chain_kernel<float, 16>), grid, blk, 0, ctx->stream, a);
    }
    GH_HIP(hipGetLastError());
    return GH_OK;
}
