// Forced-alignment lattices in SEQUENCE form (gh_seqgraph, gh_internal.h): the graphs `continuous_train` decodes
// every utterance through (continuous_speech.py:80-89: one word per layer, the words of the transcript) -- Viterbi
// (decode_hmm_states, decode.py:80-146) and its sum-product twin (forward-backward).
//
// gfx950 mapping: FOUR UTTERANCES PER WAVE -- DPP row = utterance, lane = layer (= word of the transcript, <= 16), the N
// states of the lane's word in registers.  Inside a word the neighbours are registers; the non-emitting row behind a
// layer has ONE origin, the last state of the layer, so it is handed to the next lane by a single row_shr:1 -- no
// reduction at all; a word's first state takes it in the SAME column (decode.py:109-111).  Rows whose utterance has
// ended are switched off by EXEC.  No LDS, no barrier.  Back-pointers are decision bits (N per column and lane), the
// back-trace is lattice_backtrace_kernel (gh_viterbi_layers.hip), one lane per utterance.
// The row-per-lane lean kernel ran these graphs with three LDS phases and barriers per column and one workgroup per
// utterance; forward-backward used the generic fb_kernel, 5 of 64 lanes busy.
#include "gh_internal.h"
#include "gh_viterbi.h"
#include "gh_fb.h"
#include "gh_xnum.h"
#include <cstring>

namespace {

__device__ __forceinline__ double sq_vmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// lane i <- lane i-1 inside the 16-lane row; lane 0 of a row keeps `fill`
__device__ __forceinline__ double row_shr1(double v, double fill) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), 0x111, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), 0x111, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// lane i <- lane i+1 inside the row; lane 15 keeps `fill`
__device__ __forceinline__ double row_shl1(double v, double fill) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), 0x101, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), 0x101, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void sq_push_bit(uint32_t& word, unsigned long long mask) {
    unsigned long long carry_out;
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(word), "=s"(carry_out) : "v"(word), "s"(mask));
}

template <typename ET, int N, bool SKIP, bool WANT_BP>
__global__ __launch_bounds__(64) void viterbi_seq_kernel(gh_layers_args a, int64_t slot_end) {
    constexpr int HB = N + (SKIP ? N - 2 : 0);                // decision bits per column and lane (<= 30: N = 16 with skips)
    constexpr int CPW = 32 / HB;
    constexpr int PF = N > 8 ? 2 : 4;                          // columns of emissions in flight (12 / 16 states: the ring is registers)
    static_assert(HB < 32, "the decision bits of a column and lane live in one word");
    const int lane = threadIdx.x, kk = lane >> 4, k = lane & 15;
    const int64_t slot = a.slot0 + (int64_t)blockIdx.x * 4 + kk;
    const bool has_utt = slot < slot_end;
    const int64_t u = has_utt ? (a.perm ? a.perm[slot] : slot) : 0;
    const gh_seqgraph* g = a.seqgraphs + ((has_utt && a.utt_lat) ? a.utt_lat[u] : 0);
    const int K = g->K;
    const int64_t f0 = has_utt ? a.utt_off[u] : 0;
    const int T = has_utt ? (int)(a.utt_off[u + 1] - f0) : 0;
    const double INF = INFINITY;
    int Tmax = T;
    Tmax = max(Tmax, __shfl_xor(Tmax, 16));
    Tmax = max(Tmax, __shfl_xor(Tmax, 32));
    const bool lact = has_utt && k < K;
    const gh_seqword* wd = a.seqwords + (lact ? g->word[k] : 0);
    double c0[N], c1[N], c2[N];
    unsigned sto[N];
#pragma unroll
    for (int s = 0; s < N; ++s) {
        c0[s] = lact ? wd->c0[s] : INF;
        c1[s] = lact ? wd->c1[s] : INF;
        c2[s] = (SKIP && lact) ? wd->c2[s] : INF;
        sto[s] = (unsigned)(lact ? wd->state[s] : 0) * (unsigned)sizeof(ET);
    }
    const double cin = lact ? wd->cin : INF, cout = lact ? wd->cout : INF;
    const char* nllb = static_cast<const char*>(a.nll) + (T > 0 ? f0 : 0) * a.S * (int64_t)sizeof(ET);   // (no frames: frame 0)
    const int64_t rowb = (int64_t)a.S * (int64_t)sizeof(ET);
    ET ring[PF][N];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int s = 0; s < N; ++s) ring[p][s] = (p < T) ? *reinterpret_cast<const ET*>(nllb + p * rowb + sto[s]) : ET(0);
    double prev[N];
#pragma unroll
    for (int s = 0; s < N; ++s) prev[s] = INF;
    uint32_t word = 0;
    uint32_t* bp = (WANT_BP && has_utt) ? reinterpret_cast<uint32_t*>(a.bp + a.bp_off[slot]) + k : nullptr;
    for (int t0 = 0; t0 < Tmax; t0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int t = t0 + p;
            double e[N];
#pragma unroll
            for (int s = 0; s < N; ++s) e[s] = (double)ring[p][s];
            if (t < T) {
                const double base0 = c0[0] + prev[0];
#pragma unroll
                for (int s = N - 1; s >= 1; --s) {
                    const double v0 = c0[s] + prev[s];
                    const double v1 = c1[s] + prev[s - 1];
                    double best;
                    if (SKIP && s >= 2) {
                        const double v2 = c2[s] + prev[s - 2];
                        const bool b_a = v1 < v2;
                        const double m = sq_vmin(v1, v2);
                        const bool b_b = v0 < m;
                        best = sq_vmin(v0, m);
                        if (WANT_BP) { sq_push_bit(word, __ballot(b_a)); sq_push_bit(word, __ballot(b_b)); }
                    } else {
                        const bool b = v0 < v1;
                        best = sq_vmin(v0, v1);
                        if (WANT_BP) sq_push_bit(word, __ballot(b));
                    }
                    prev[s] = sq_vmin(best + e[s], INF);
                }
                // the non-emitting row behind this layer has one origin: hand it to the next lane (same column)
                const double nin = row_shr1(prev[N - 1] + cout, (t == 0) ? 0.0 : INF);   // lane 0: the start row
                const double cn = nin + cin;
                const bool b0 = base0 < cn;                   // the non-emitting row (lower row index) wins ties
                if (WANT_BP) sq_push_bit(word, __ballot(b0));
                prev[0] = sq_vmin(sq_vmin(base0, cn) + e[0], INF);
                if (WANT_BP) {
                    const int ci = t % CPW;
                    if (ci == CPW - 1 || t == T - 1) {
                        if (CPW > 1 && ci < CPW - 1) word <<= HB * (CPW - 1 - ci);
                        bp[(int64_t)(t / CPW) * 16] = word;
                        word = 0;
                    }
                }
            }
            {   // the slot's refill: unconditional, from a clamped column, OUTSIDE the divergent region and behind the last
                // use of the old value.  (A load under `if (t + PF < T)`, a `break` out of the unrolled group, or a refill
                // issued while the old value is live each put an s_waitcnt vmcnt(0) into every column or group: the ring
                // never ran ahead.)  Rows without frames read frame 0 (some row has frames, or this loop would not run).
                const int tn = (t + PF < T) ? t + PF : (T > 0 ? T - 1 : 0);
                const char* colp = nllb + (int64_t)tn * rowb;
#pragma unroll
                for (int s = 0; s < N; ++s) ring[p][s] = *reinterpret_cast<const ET*>(colp + sto[s]);
            }
        }
    }
    if (!has_utt) return;
    // ---- end costs + end selection inside the row ('>=': the last of equal minima) ----
    double best_v = INF;
    int best_slot = -1;
    double* ec = a.end_cost ? a.end_cost + (a.end_off ? a.end_off[u] : u * (int64_t)g->n_end) : nullptr;
#pragma unroll
    for (int s = 0; s < N; ++s) {
        if (lact) {
            const int r = k * (N + 1) + 1 + s;
            const int es = a.end_slot[g->row_base + r];
            if (es >= 0) {
                const double v = T > 0 ? prev[s] : INF;
                if (ec) ec[es] = v;
                if (v < best_v || (v == best_v && es > best_slot)) { best_v = v; best_slot = es; }
            }
        }
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(best_v, o);
        const int os = __shfl_xor(best_slot, o);
        if (ov < best_v || (ov == best_v && os > best_slot)) { best_v = ov; best_slot = os; }
    }
    if (k == 0 && a.best_end) a.best_end[u] = T > 0 ? best_slot : -1;
}

// Back-trace (decode.py:143-145), one lane per utterance (see lattice_backtrace_kernel for why not one wave): the cell
// (layer bk, state bs) in column j; a lane keeps the decision word of its current (word index, layer) in a register.
template <int N, bool SKIP>
__global__ __launch_bounds__(64) void seq_backtrace_kernel(gh_layers_args a, int64_t slot_end) {
    constexpr int HB = N + (SKIP ? N - 2 : 0);
    constexpr int CPW = 32 / HB;
    const int64_t slot = a.slot0 + (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (slot >= slot_end) return;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const gh_seqgraph* g = a.seqgraphs + (a.utt_lat ? a.utt_lat[u] : 0);
    const int T = (int)(a.utt_off[u + 1] - a.utt_off[u]);
    const int be = a.best_end[u];
    if (T <= 1 || be < 0) { a.path_len[u] = 0; return; }
    int bk, bs;
    {
        const int r = a.end_rows[g->end_base + be];
        bk = (r - 1) / (N + 1);
        bs = (r - 1) % (N + 1);
    }
    const uint32_t* bpu = reinterpret_cast<const uint32_t*>(a.bp + a.bp_off[slot]);
    int2* path = reinterpret_cast<int2*>(a.path + 2 * a.path_off[u]);
    const int64_t cap = a.path_off[u + 1] - a.path_off[u];
    int64_t len = 0;
    int flag = 0;
    int j = T - 1;
    int64_t key = -1;
    uint32_t cw = 0;
    // the arcs of the current word's states in registers (16 bytes, fetched when the layer changes): a load per CELL, in front
    // of the decision it validates, was a round trip in the chain of every column
    // (a second decision word fetched ahead -- 16 words below, the same layer CPW columns earlier -- made the kernel SLOWER,
    //  634 -> 769 us on continuous_train's 2 000 utterances: the lanes' words lie on 64 different lines, twice the loads)
    uint4 aw = *reinterpret_cast<const uint4*>(a.seqwords[g->word[bk]].arcs);
    auto arcs_of = [&](int st) -> int {
        const uint32_t w4 = st < 8 ? (st < 4 ? aw.x : aw.y) : (st < 12 ? aw.z : aw.w);
        return (int)((w4 >> ((st & 3) * 8)) & 0xffu);
    };
    while (j != 0) {
        const int64_t want = (int64_t)(j / CPW) * 16 + bk;
        if (want != key) { cw = bpu[want]; key = want; }
        const uint32_t hb = (cw >> ((CPW - 1 - j % CPW) * HB)) & ((1u << HB) - 1u);
        const int arcs = arcs_of(bs);
        if (len + 2 > cap) { flag |= 4; break; }
        if (bs >= 1) {
            int before = 0;
            for (int s2 = N - 1; s2 > bs; --s2) before += (SKIP && s2 >= 2) ? 2 : 1;
            int code;
            if (SKIP && bs >= 2) {
                const int b_a = (hb >> (HB - 1 - before)) & 1, b_b = (hb >> (HB - 2 - before)) & 1;
                code = b_b ? 0 : (b_a ? 1 : 2);
            } else {
                code = ((hb >> (HB - 1 - before)) & 1) ? 0 : 1;
            }
            // every candidate was +inf: the first existing arc (lowest origin) -- or none at all
            if (!((arcs >> code) & 1)) code = (arcs & 4) ? 2 : (arcs & 2) ? 1 : (arcs & 1) ? 0 : -1;
            if (code < 0) { flag |= 2; break; }
            bs -= code;
            --j;
            path[len++] = make_int2(bk * (N + 1) + 1 + bs, j);
        } else if ((hb & 1) && (arcs & 1)) {                  // self loop of the word's first state
            --j;
            path[len++] = make_int2(bk * (N + 1) + 1, j);
        } else {
            path[len++] = make_int2(bk * (N + 1), j);         // the non-emitting row in front of the layer ...
            if (bk == 0) { flag |= 2; break; }                // ... the start row in a column > 0: no origin
            --bk;                                             // ... whose only origin is the last state of the layer before
            bs = N - 1;
            aw = *reinterpret_cast<const uint4*>(a.seqwords[g->word[bk]].arcs);
            path[len++] = make_int2(bk * (N + 1) + 1 + bs, j);
        }
    }
    if (flag) atomicOr(a.flag, flag);
    a.path_len[u] = (int32_t)len;
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward-backward over the same graphs (definition: fb_kernel, gh_fb.hip -- same arcs, same same-column rule, start row
// in column 0, end rows in the last column, min -> -logsumexp).  The recursion runs on probabilities with an extended
// exponent (gh_xnum.h): a term is a multiply and an integer add, the one transcendental per state and column is the
// split exponential of the emission cost.  (A first version in the log domain -- exp per term, log per sum -- spent
// 3 700 instructions per column and lane and was issue bound at 6.7 ms for 12 500 seven-word transcripts.)
// alpha columns go to HBM scratch ([T, K, N] mantissas, then [T, K, N] exponents) and come back in the backward sweep,
// beta lives in registers.  Inside a column the backward order is: states 0..N-2 of every layer (they only read the
// next column), the non-emitting rows (one row_shl:1), then the last states.  gamma is added into occ[frame, state]
// with double atomics (a word may stand in several layers of one transcript); the expected self transitions are kept
// per lane and added into one of GH_FBSEQ_XI_PARTS partial vectors.
__device__ __forceinline__ xnum xn_row_shr1(xnum v, xnum fill) {
    xnum o;
    o.f = row_shr1(v.f, fill.f);
    o.e = __builtin_amdgcn_update_dpp(fill.e, v.e, 0x111, 0xF, 0xF, false);
    return o;
}
__device__ __forceinline__ xnum xn_row_shl1(xnum v, xnum fill) {
    xnum o;
    o.f = row_shl1(v.f, fill.f);
    o.e = __builtin_amdgcn_update_dpp(fill.e, v.e, 0x101, 0xF, 0xF, false);
    return o;
}

template <typename ET, int N, bool SKIP, bool OCC_LDS>
__global__ __launch_bounds__(64) void fb_seq_kernel(gh_fbseq_args a, int64_t slot_end) {
#ifndef GH_FBSEQ_PF
#define GH_FBSEQ_PF 2
#endif
    constexpr int PF = GH_FBSEQ_PF;                           // columns whose loads are in flight ahead of the recursion
    extern __shared__ __attribute__((aligned(16))) double seq_lds[];   // OCC_LDS: [4 utterances][S] occupancies of a frame
    const int lane = threadIdx.x, kk = lane >> 4, k = lane & 15;
    double* occ_row = seq_lds + (OCC_LDS ? kk * a.S : 0);
    if (OCC_LDS) {
        for (int s = k; s < a.S; s += 16) occ_row[s] = 0.0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    const int64_t slot = a.slot0 + (int64_t)blockIdx.x * 4 + kk;
    const bool has_utt = slot < slot_end;
    const int64_t u = has_utt ? (a.perm ? a.perm[slot] : slot) : 0;
    const gh_seqgraph* g = a.graphs + ((has_utt && a.utt_lat) ? a.utt_lat[u] : 0);
    const int K = g->K;
    const int64_t f0 = has_utt ? a.utt_off[u] : 0;
    const int T = has_utt ? (int)(a.utt_off[u + 1] - f0) : 0;
    const double INF = INFINITY;
    int Tmax = T;
    Tmax = max(Tmax, __shfl_xor(Tmax, 16));
    Tmax = max(Tmax, __shfl_xor(Tmax, 32));
    const bool lact = has_utt && k < K;
    const gh_seqword* wd = a.words + (lact ? g->word[k] : 0);
    xnum p0[N], p1[N], p2[N];                                 // arc probabilities exp(-cost); absent arcs: 0
    int st[N];
    unsigned endm = 0;
#pragma unroll
    for (int s = 0; s < N; ++s) {
        p0[s] = xn_exp_neg(lact ? wd->c0[s] : INF);
        p1[s] = xn_exp_neg(lact ? wd->c1[s] : INF);
        p2[s] = xn_exp_neg((SKIP && lact) ? wd->c2[s] : INF);
        st[s] = lact ? wd->state[s] : 0;
        if (lact && a.end_slot[g->row_base + k * (N + 1) + 1 + s] >= 0) endm |= 1u << s;
    }
    const xnum pin = xn_exp_neg(lact ? wd->cin : INF), pout = xn_exp_neg(lact ? wd->cout : INF);
    const ET* nll = static_cast<const ET*>(a.nll) + f0 * a.S;
    const int64_t astride = (int64_t)K * N;
    double* alf = a.alpha_scratch + (has_utt ? a.scratch_off[slot] : 0) + (int64_t)k * N;             // [T, K, N]
    int* ale = reinterpret_cast<int*>(a.alpha_scratch + (has_utt ? a.scratch_off[slot] : 0) + (int64_t)T * astride) + k * N;
    if (T <= 0) {                                             // (a whole row: no shuffle below involves it)
        if (has_utt && k == 0 && a.logp) a.logp[u] = -INF;
    }
    // ---- forward ----
    ET ring[PF][N];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int s = 0; s < N; ++s) ring[p][s] = (lact && p < T) ? nll[(int64_t)p * a.S + st[s]] : ET(0);
    xnum al[N];
#pragma unroll
    for (int s = 0; s < N; ++s) al[s] = xn_zero();
    for (int t0 = 0; t0 < Tmax; t0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int t = t0 + p;
            if (t >= Tmax) break;
            if (t < T) {
                xnum b[N];
#pragma unroll
                for (int s = 0; s < N; ++s) b[s] = xn_exp_neg((double)ring[p][s]);
                if (lact && t + PF < T) {
#pragma unroll
                    for (int s = 0; s < N; ++s) ring[p][s] = nll[(int64_t)(t + PF) * a.S + st[s]];
                }
                const xnum self0 = xn_mul(al[0], p0[0]);
#pragma unroll
                for (int s = N - 1; s >= 1; --s) {
                    const xnum a0 = xn_mul(al[s], p0[s]), a1 = xn_mul(al[s - 1], p1[s]);
                    const xnum sum = (SKIP && s >= 2) ? xn_add3(a0, a1, xn_mul(al[s - 2], p2[s])) : xn_add(a0, a1);
                    al[s] = xn_norm(xn_mul(sum, b[s]));
                }
                const xnum nin = xn_row_shr1(xn_mul(al[N - 1], pout), (t == 0) ? xn_one() : xn_zero());   // lane 0: the start row
                al[0] = xn_norm(xn_mul(xn_add(self0, xn_mul(nin, pin)), b[0]));
                if (lact) {
#pragma unroll
                    for (int s = 0; s < N; ++s) { alf[(int64_t)t * astride + s] = al[s].f; ale[(int64_t)t * astride + s] = al[s].e; }
                }
            }
        }
    }
    xnum P = xn_zero();
#pragma unroll
    for (int s = 0; s < N; ++s) if (endm >> s & 1) P = xn_add(P, al[s]);
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) {
        xnum q;
        q.f = __shfl_xor(P.f, o);
        q.e = __shfl_xor(P.e, o);
        P = xn_add(P, q);
    }
    P = xn_norm(P);
    if (has_utt && k == 0 && T > 0 && a.logp) a.logp[u] = xn_log(P);
    if (!a.occ && !a.self_xi_parts) return;
    const bool reach = P.f > 0.0;                             // log P = -inf: every gamma is 0 (the generic kernel's NaN -> 0)
    const double inv_pf = 1.0 / P.f;
    // ---- backward (column t = T - 1 - i of this row's utterance) ----
    xnum be[N], bn[N];                                        // beta and emission probability of column t + 1
    double xi[N];
#pragma unroll
    for (int s = 0; s < N; ++s) { be[s] = xn_zero(); bn[s] = xn_one(); xi[s] = 0.0; }
    int seg_first = 0x7fffffff, seg_last = -1;             // frames of this layer with occupancy above the floor
    ET ering[PF][N];
    double arf[PF][N];
    int are[PF][N];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int s = 0; s < N; ++s) {
            const int t = T - 1 - p;
            ering[p][s] = (lact && t >= 0) ? nll[(int64_t)t * a.S + st[s]] : ET(0);
            arf[p][s] = (lact && t >= 0) ? alf[(int64_t)t * astride + s] : 0.0;
            are[p][s] = (lact && t >= 0) ? ale[(int64_t)t * astride + s] : XN_ZERO_E;
        }
    for (int i0 = 0; i0 < Tmax; i0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int i = i0 + p;
            if (i >= Tmax) break;
            if (i < T) {
                const int t = T - 1 - i;
                xnum b[N], av[N];
#pragma unroll
                for (int s = 0; s < N; ++s) { b[s] = xn_exp_neg((double)ering[p][s]); av[s] = xnum{arf[p][s], are[p][s]}; }
                if (lact && t - PF >= 0) {
#pragma unroll
                    for (int s = 0; s < N; ++s) {
                        ering[p][s] = nll[(int64_t)(t - PF) * a.S + st[s]];
                        arf[p][s] = alf[(int64_t)(t - PF) * astride + s];
                        are[p][s] = ale[(int64_t)(t - PF) * astride + s];
                    }
                }
                xnum w[N], ws[N], nb[N];
#pragma unroll
                for (int s = 0; s < N; ++s) {
                    w[s] = xn_mul(be[s], bn[s]);                  // column t + 1 seen from column t, before the arc
                    ws[s] = xn_mul(w[s], p0[s]);                  // ... through the self arc
                }
                const xnum end1 = (i == 0) ? xn_one() : xn_zero();
#pragma unroll
                for (int s = 0; s < N - 1; ++s) {
                    const xnum b1 = xn_mul(w[s + 1], p1[s + 1]);
                    xnum sum = (SKIP && s + 2 < N) ? xn_add3(ws[s], b1, xn_mul(w[s + 2], p2[s + 2])) : xn_add(ws[s], b1);
                    if (endm >> s & 1) sum = xn_add(sum, end1);
                    nb[s] = xn_norm(sum);
                }
                // the non-emitting row in front of layer k + 1 (its only successor: that layer's first state, same column)
                const xnum nesb = xn_row_shl1(xn_mul(xn_mul(nb[0], pin), b[0]), xn_zero());
                {
                    xnum sum = xn_add(ws[N - 1], xn_mul(nesb, pout));
                    if (endm >> (N - 1) & 1) sum = xn_add(sum, end1);
                    nb[N - 1] = xn_norm(sum);
                }
                if (lact && reach) {
#pragma unroll
                    for (int s = 0; s < N; ++s) {
                        if (a.self_xi_parts && i > 0)               // xi_t(s -> s) = alpha_t(s) a_ss b_s(x_{t+1}) beta_{t+1}(s) / P
                            xi[s] += xn_ratio(av[s], ws[s], inv_pf, P.e);
                        if (a.occ) {
                            const double gm = xn_ratio(av[s], nb[s], inv_pf, P.e);
                            if (OCC_LDS) { if (gm != 0.0) atomicAdd(occ_row + st[s], gm); }   // ds_add_f64
                            else if (gm != 0.0) unsafeAtomicAdd(a.occ + (f0 + t) * a.S + st[s], gm);
                            if (gm > a.occ_floor || gm != gm) { seg_first = t; seg_last = max(seg_last, t); }   // (t descends)
                        }
                    }
                }
                if (OCC_LDS && a.occ) {
                    // the row of S occupancies of this frame: out of LDS as one contiguous store, and back to zero
                    // (LDS executes a wave's instructions in order; the fences only keep the compiler from reordering)
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    double* orow = a.occ + (f0 + t) * a.S;
                    for (int s = k; s < a.S; s += 16) { orow[s] = occ_row[s]; occ_row[s] = 0.0; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
#pragma unroll
                for (int s = 0; s < N; ++s) { be[s] = nb[s]; bn[s] = b[s]; }
            }
        }
    }
    if (a.seg_lo && has_utt && k < GH_SEQ_MAXK) {
        a.seg_lo[u * GH_SEQ_MAXK + k] = lact ? seg_first : 0x7fffffff;
        a.seg_hi[u * GH_SEQ_MAXK + k] = lact ? seg_last : -1;
    }
    if (a.self_xi_parts && lact) {
        double* part = a.self_xi_parts + (size_t)(blockIdx.x % GH_FBSEQ_XI_PARTS) * a.S;
#pragma unroll
        for (int s = 0; s < N; ++s) if (xi[s] != 0.0) unsafeAtomicAdd(part + st[s], xi[s]);
    }
}

// The same forward-backward with ONE UTTERANCE PER WAVE and lane = CELL (layer k, state s; lane = k N + s, K N <= 64).
// With the lane = layer mapping above a lane walks the N states of its word one after the other -- N split exponentials
// and N sums per column on a dependent chain, ~250 fp64 instructions per column forward and ~350 backward -- and a batch
// of 2 000 seven-word utterances is 500 waves: one wave on every other SIMD, nothing to hide the latency of the chain
// behind (2.1 ms for 1.4 M frames, no faster with deeper read-ahead).  Here a column is ONE exponential, one sum and one
// normalisation per lane; the neighbours are wave_shr:1 / wave_shl:1 away (a word's first state has no arc from the cell
// below it -- its p1 is 0 -- so the shift may cross the word boundary), and the same-column hand-over through the
// non-emitting row is a second, short step for the first (forward) / last (backward) state of every layer.  Same
// arithmetic in the same order per cell: the alpha columns ([T, K N] mantissas + exponents, the layout of the kernel
// above), log P, gamma and xi are bit-identical to the lane = layer kernel's whenever a graph has one end row.
// Also written: the frame range of every CELL with occupancy above the floor (row_lo / row_hi), which lets the fused
// statistics kernel walk a state pair's own frames instead of its layer's.
__device__ __forceinline__ double wave_shr1d(double v, double fill) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), 0x138, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shl1d(double v, double fill) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), 0x130, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ xnum xn_wave_shr1(xnum v, xnum fill) {
    return xnum{wave_shr1d(v.f, fill.f), __builtin_amdgcn_update_dpp(fill.e, v.e, 0x138, 0xF, 0xF, false)};
}
__device__ __forceinline__ xnum xn_wave_shl1(xnum v, xnum fill) {
    return xnum{wave_shl1d(v.f, fill.f), __builtin_amdgcn_update_dpp(fill.e, v.e, 0x130, 0xF, 0xF, false)};
}

template <typename ET, int N, bool SKIP, bool OCC_LDS>
__global__ __launch_bounds__(64) void fb_seq_cell_kernel(gh_fbseq_args a) {
#ifndef GH_FBSEQ_CELL_PF
#define GH_FBSEQ_CELL_PF 4
#endif
    constexpr int PF = GH_FBSEQ_CELL_PF;
    extern __shared__ __attribute__((aligned(16))) double seq_lds[];   // OCC_LDS: [S] occupancies of a frame
    __shared__ int lay_lo[GH_SEQ_MAXK], lay_hi[GH_SEQ_MAXK];
    const int lane = threadIdx.x;
    const int64_t slot = a.slot0 + blockIdx.x;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const gh_seqgraph* g = a.graphs + (a.utt_lat ? a.utt_lat[u] : 0);
    const int K = g->K, C = K * N;
    const int k = lane / N, s = lane - k * N;
    const bool lact = lane < C;
    const int64_t f0 = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - f0);
    const double INF = INFINITY;
    if (OCC_LDS) for (int i = lane; i < a.S; i += 64) seq_lds[i] = 0.0;
    if (lane < GH_SEQ_MAXK) { lay_lo[lane] = 0x7fffffff; lay_hi[lane] = -1; }
    __syncthreads();
    const gh_seqword* wd = a.words + (lact ? g->word[k] : 0);
    const xnum Z = xn_zero();
    const xnum p0 = lact ? xn_exp_neg(wd->c0[s]) : Z;
    const xnum p1 = (lact && s >= 1) ? xn_exp_neg(wd->c1[s]) : Z;
    const xnum p2 = (SKIP && lact && s >= 2) ? xn_exp_neg(wd->c2[s]) : Z;
    const xnum pin = (lact && s == 0) ? xn_exp_neg(wd->cin) : Z;          // hand-over INTO a layer: its first state
    const xnum pout = (lact && s == N - 1) ? xn_exp_neg(wd->cout) : Z;    // ... OUT of a layer: its last state
    const int st = lact ? wd->state[s] : 0;
    const bool is_end = lact && a.end_slot[g->row_base + k * (N + 1) + 1 + s] >= 0;
    const ET* nll = static_cast<const ET*>(a.nll) + f0 * a.S + st;
    const int lane_c = lact ? lane : 0;                                     // (idle lanes read along with lane 0: no branch around a load)
    double* alf = a.alpha_scratch + a.scratch_off[slot] + lane_c;           // [T, C]
    int* ale = reinterpret_cast<int*>(a.alpha_scratch + a.scratch_off[slot] + (int64_t)T * C) + lane_c;
    if (T <= 0) {
        if (lane == 0 && a.logp) a.logp[u] = -INF;
        if (a.seg_lo && lane < GH_SEQ_MAXK) { a.seg_lo[u * GH_SEQ_MAXK + lane] = 0x7fffffff; a.seg_hi[u * GH_SEQ_MAXK + lane] = -1; }
        return;
    }
    // The read-ahead rings are filled by UNCONDITIONAL loads from clamped columns, and the columns behind the last one of a
    // PF-group are computed and thrown away: a load inside a branch (or an exec-masked region) comes with an s_waitcnt
    // vmcnt(0) where the paths join -- every column then waited a whole memory round trip, 2 500 cycles for ~100
    // instructions of arithmetic, and no read-ahead depth changed that.
    const int Tp = (T + PF - 1) / PF * PF;
    // ---- forward ----
    ET ring[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) ring[p] = nll[(int64_t)min(p, T - 1) * a.S];
    xnum al = Z;
    for (int t0 = 0; t0 < Tp; t0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int t = t0 + p;
            const bool on = t < T;
            const xnum b = xn_exp_neg_sc((double)ring[p]);
            ring[p] = nll[(int64_t)min(t + PF, T - 1) * a.S];
            const xnum pr1 = xn_wave_shr1(al, Z);
            const xnum a0 = xn_mul(al, p0), a1 = xn_mul(pr1, p1);
            const xnum x = SKIP ? xn_add3(a0, a1, xn_mul(xn_wave_shr1(pr1, Z), p2)) : xn_add(a0, a1);
            const xnum y = xn_norm(xn_mul(x, b));                       // final for every state but a layer's first
            const xnum nin = xn_wave_shr1(xn_mul(y, pout), (t == 0) ? xn_one() : Z);   // lane 0: the start row
            const xnum z = xn_norm(xn_mul(xn_add(a0, xn_mul(nin, pin)), b));
            const xnum an = (s == 0) ? z : y;
            al.f = on ? an.f : al.f;
            al.e = on ? an.e : al.e;
            if (lact && on) { alf[(int64_t)t * C] = al.f; ale[(int64_t)t * C] = al.e; }
        }
    }
    xnum P = is_end ? al : Z;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        xnum q;
        q.f = __shfl_xor(P.f, o);
        q.e = __shfl_xor(P.e, o);
        P = xn_add(P, q);
    }
    P = xn_norm(P);
    if (lane == 0 && a.logp) a.logp[u] = xn_log(P);
    if (!a.occ && !a.self_xi_parts) return;
    const bool reach = P.f > 0.0;                             // log P = -inf: every gamma is 0 (the generic kernel's NaN -> 0)
    const double inv_pf = 1.0 / P.f;
    // ---- backward (column t = T - 1 - i) ----
    xnum be = Z, bn = xn_one();                               // beta and emission probability of column t + 1
    double xi = 0.0;
    int seg_first = 0x7fffffff, seg_last = -1;
    ET ering[PF];
    double arf[PF];
    int are[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int t = max(T - 1 - p, 0);
        ering[p] = nll[(int64_t)t * a.S];
        arf[p] = alf[(int64_t)t * C];
        are[p] = ale[(int64_t)t * C];
    }
    for (int i0 = 0; i0 < Tp; i0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int i = i0 + p;
            const int t = T - 1 - i;
            const bool on = t >= 0;
            const xnum b = xn_exp_neg_sc((double)ering[p]);
            const xnum av = xnum{arf[p], are[p]};
            const xnum w = xn_mul(be, bn);                            // column t + 1 seen from column t, before the arc
            const xnum ws = xn_mul(w, p0);                            // ... through the self arc
            const xnum q1 = xn_wave_shl1(xn_mul(w, p1), Z);           // ... the cell above, through ITS arc from below (0 across words)
            xnum sum = SKIP ? xn_add3(ws, q1, xn_wave_shl1(xn_wave_shl1(xn_mul(w, p2), Z), Z)) : xn_add(ws, q1);
            // (the end rows add 1 in the LAST column only: a scalar branch, taken once, instead of two sums per column)
            if (i == 0 && is_end) sum = xn_add(sum, xn_one());
            const xnum nba = xn_norm(sum);                            // final for every state but a layer's last
            // the non-emitting row in front of the NEXT layer (its only successor: that layer's first state, same column)
            const xnum nesb = xn_wave_shl1(xn_mul(xn_mul(nba, pin), b), Z);
            xnum sl = xn_add(ws, xn_mul(nesb, pout));
            if (i == 0 && is_end) sl = xn_add(sl, xn_one());
            const xnum nb = (s == N - 1) ? xn_norm(sl) : nba;
            if (on) {                                                 // (uniform; nothing in here is loaded from HBM)
                if (lact && reach) {
                    if (a.self_xi_parts && i > 0) xi += xn_ratio(av, ws, inv_pf, P.e);
                    if (a.occ) {
                        const double gm = xn_ratio(av, nb, inv_pf, P.e);
                        if (OCC_LDS) { if (gm != 0.0) atomicAdd(seq_lds + st, gm); }   // ds_add_f64 (a word may stand in several layers)
                        else if (gm != 0.0) unsafeAtomicAdd(a.occ + (f0 + t) * a.S + st, gm);
                        if (gm > a.occ_floor || gm != gm) { seg_first = t; seg_last = max(seg_last, t); }   // (t descends)
                    }
                }
                if (OCC_LDS && a.occ) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    double* orow = a.occ + (f0 + t) * a.S;
                    for (int j = lane; j < a.S; j += 64) { orow[j] = seq_lds[j]; seq_lds[j] = 0.0; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                }
            }
            be = nb;                                                  // (columns behind the first: nothing reads them any more)
            bn = b;
            {   // column t - PF into the ring slot this column has just freed (issued any earlier, the old and the new value
                // are both live and the loop's back edge copies registers -- behind an s_waitcnt vmcnt(0))
                const int tn = max(t - PF, 0);
                ering[p] = nll[(int64_t)tn * a.S];
                arf[p] = alf[(int64_t)tn * C];
                are[p] = ale[(int64_t)tn * C];
            }
        }
    }
    if (a.row_lo && lact) {
        a.row_lo[(u * GH_SEQ_MAXK + k) * GH_LAYERS_MAXN + s] = seg_first;
        a.row_hi[(u * GH_SEQ_MAXK + k) * GH_LAYERS_MAXN + s] = seg_last;
    }
    if (a.seg_lo) {
        if (lact && seg_last >= seg_first) { atomicMin(&lay_lo[k], seg_first); atomicMax(&lay_hi[k], seg_last); }
        __syncthreads();
        if (lane < GH_SEQ_MAXK) {
            a.seg_lo[u * GH_SEQ_MAXK + lane] = lane < K ? lay_lo[lane] : 0x7fffffff;
            a.seg_hi[u * GH_SEQ_MAXK + lane] = lane < K ? lay_hi[lane] : -1;
        }
    }
    if (a.self_xi_parts && lact && xi != 0.0)
        unsafeAtomicAdd(a.self_xi_parts + (size_t)(blockIdx.x % GH_FBSEQ_XI_PARTS) * a.S + st, xi);
}

}  // namespace

size_t gh_seq_bp_entries(int N, int skip, int64_t T) {
    const int hb = N + (skip ? N - 2 : 0);
    const int cpw = 32 / hb;
    return (size_t)((T + cpw - 1) / cpw) * 16 * 2;
}

int gh_launch_viterbi_seq(gh_ctx* ctx, const gh_layers_args& a, int N, int skip, int64_t u_begin, int64_t n_utts, bool f64,
                          bool want_path) {
    if (n_utts <= 0) return GH_OK;
    gh_layers_args b = a;
    b.slot0 = u_begin;
    const dim3 grid((unsigned)((n_utts + 3) / 4)), blk(64);
    const int64_t slot_end = u_begin + n_utts;
#define GH_SQ(ET, NN, SK)                                                                                               \
    do {                                                                                                                \
        if (want_path) hipLaunchKernelGGL((viterbi_seq_kernel<ET, NN, SK, true>), grid, blk, 0, ctx->stream, b, slot_end);  \
        else hipLaunchKernelGGL((viterbi_seq_kernel<ET, NN, SK, false>), grid, blk, 0, ctx->stream, b, slot_end);           \
    } while (0)
#define GH_SQ_N(ET)                                                                          \
    switch (N) {                                                                             \
        case 2: GH_SQ(ET, 2, false); break;                                                  \
        case 3: if (skip) GH_SQ(ET, 3, true); else GH_SQ(ET, 3, false); break;               \
        case 4: if (skip) GH_SQ(ET, 4, true); else GH_SQ(ET, 4, false); break;               \
        case 5: if (skip) GH_SQ(ET, 5, true); else GH_SQ(ET, 5, false); break;               \
        case 6: if (skip) GH_SQ(ET, 6, true); else GH_SQ(ET, 6, false); break;               \
        case 7: if (skip) GH_SQ(ET, 7, true); else GH_SQ(ET, 7, false); break;               \
        case 8: if (skip) GH_SQ(ET, 8, true); else GH_SQ(ET, 8, false); break;               \
        case 12: if (skip) GH_SQ(ET, 12, true); else GH_SQ(ET, 12, false); break;            \
        case 16: if (skip) GH_SQ(ET, 16, true); else GH_SQ(ET, 16, false); break;            \
        default: gh_set_error("gh_viterbi: sequence form with %d states per word", N); return GH_ERR_UNSUPPORTED; \
    }
    if (f64) { GH_SQ_N(double) } else { GH_SQ_N(float) }
#undef GH_SQ_N
#undef GH_SQ
    GH_HIP(hipGetLastError());
    return GH_OK;
}

int gh_launch_seq_backtrace(gh_ctx* ctx, const gh_layers_args& a, int N, int skip, int64_t u_begin, int64_t n_utts) {
    if (n_utts <= 0 || !a.path) return GH_OK;
    gh_layers_args b = a;
    b.slot0 = u_begin;
    const dim3 grid((unsigned)((n_utts + 63) / 64)), blk(64);
    const int64_t slot_end = u_begin + n_utts;
#define GH_SB(NN, SK) hipLaunchKernelGGL((seq_backtrace_kernel<NN, SK>), grid, blk, 0, ctx->stream, b, slot_end)
    switch (N) {
        case 2: GH_SB(2, false); break;
        case 3: if (skip) GH_SB(3, true); else GH_SB(3, false); break;
        case 4: if (skip) GH_SB(4, true); else GH_SB(4, false); break;
        case 5: if (skip) GH_SB(5, true); else GH_SB(5, false); break;
        case 6: if (skip) GH_SB(6, true); else GH_SB(6, false); break;
        case 7: if (skip) GH_SB(7, true); else GH_SB(7, false); break;
        case 8: if (skip) GH_SB(8, true); else GH_SB(8, false); break;
        case 12: if (skip) GH_SB(12, true); else GH_SB(12, false); break;
        case 16: if (skip) GH_SB(16, true); else GH_SB(16, false); break;
        default: gh_set_error("gh_viterbi: sequence form with %d states per word", N); return GH_ERR_UNSUPPORTED;
    }
#undef GH_SB
    GH_HIP(hipGetLastError());
    return GH_OK;
}

// 1: the launch below takes the lane = cell kernel (one utterance per wave; every graph has <= 64 cells), which also
// fills row_lo / row_hi; GMMHMM_FBSEQ=layer keeps the lane = layer kernel
int gh_fb_seq_by_cell(const gh_fbseq_args& a, int N) {
    const char* e = getenv("GMMHMM_FBSEQ");
    return a.max_cells > 0 && a.max_cells <= 64 && N >= 2 && !(e && !strcmp(e, "layer"));
}

int gh_launch_fb_seq(gh_ctx* ctx, const gh_fbseq_args& a, int N, int skip, int64_t u_begin, int64_t n_utts, bool f64) {
    if (n_utts <= 0) return GH_OK;
    gh_fbseq_args b = a;
    b.slot0 = u_begin;
    if (gh_fb_seq_by_cell(a, N)) {
        const dim3 grid((unsigned)n_utts), blk(64);
        const bool occ_lds = a.occ && a.occ_in_lds;
#define GH_FC(ET, NN, SK)                                                                                                    \
    do {                                                                                                                     \
        if (occ_lds) hipLaunchKernelGGL((fb_seq_cell_kernel<ET, NN, SK, true>), grid, blk, (size_t)a.S * 8, ctx->stream, b); \
        else hipLaunchKernelGGL((fb_seq_cell_kernel<ET, NN, SK, false>), grid, blk, 0, ctx->stream, b);                      \
    } while (0)
#define GH_FC_N(ET)                                                                          \
    switch (N) {                                                                             \
        case 2: GH_FC(ET, 2, false); break;                                                  \
        case 3: if (skip) GH_FC(ET, 3, true); else GH_FC(ET, 3, false); break;               \
        case 4: if (skip) GH_FC(ET, 4, true); else GH_FC(ET, 4, false); break;               \
        case 5: if (skip) GH_FC(ET, 5, true); else GH_FC(ET, 5, false); break;               \
        case 6: if (skip) GH_FC(ET, 6, true); else GH_FC(ET, 6, false); break;               \
        case 7: if (skip) GH_FC(ET, 7, true); else GH_FC(ET, 7, false); break;               \
        case 8: if (skip) GH_FC(ET, 8, true); else GH_FC(ET, 8, false); break;               \
        default: gh_set_error("gh_forward_backward: sequence form with %d states per word", N); return GH_ERR_UNSUPPORTED; \
    }
        if (f64) { GH_FC_N(double) } else { GH_FC_N(float) }
#undef GH_FC_N
#undef GH_FC
        GH_HIP(hipGetLastError());
        return GH_OK;
    }
    b.row_lo = b.row_hi = nullptr;
    const dim3 grid((unsigned)((n_utts + 3) / 4)), blk(64);
    const int64_t slot_end = u_begin + n_utts;
    const bool occ_lds = a.occ && a.occ_in_lds;
#define GH_FS(ET, NN, SK)                                                                                                       \
    do {                                                                                                                        \
        if (occ_lds) hipLaunchKernelGGL((fb_seq_kernel<ET, NN, SK, true>), grid, blk, (size_t)4 * a.S * 8, ctx->stream, b, slot_end); \
        else hipLaunchKernelGGL((fb_seq_kernel<ET, NN, SK, false>), grid, blk, 0, ctx->stream, b, slot_end);                    \
    } while (0)
#define GH_FS_N(ET)                                                                          \
    switch (N) {                                                                             \
        case 2: GH_FS(ET, 2, false); break;                                                  \
        case 3: if (skip) GH_FS(ET, 3, true); else GH_FS(ET, 3, false); break;               \
        case 4: if (skip) GH_FS(ET, 4, true); else GH_FS(ET, 4, false); break;               \
        case 5: if (skip) GH_FS(ET, 5, true); else GH_FS(ET, 5, false); break;               \
        case 6: if (skip) GH_FS(ET, 6, true); else GH_FS(ET, 6, false); break;               \
        case 7: if (skip) GH_FS(ET, 7, true); else GH_FS(ET, 7, false); break;               \
        case 8: if (skip) GH_FS(ET, 8, true); else GH_FS(ET, 8, false); break;               \
        default: gh_set_error("gh_forward_backward: sequence form with %d states per word", N); return GH_ERR_UNSUPPORTED; \
    }
    if (f64) { GH_FS_N(double) } else { GH_FS_N(float) }
#undef GH_FS_N
#undef GH_FS
    GH_HIP(hipGetLastError());
    return GH_OK;
}
