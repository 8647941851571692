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
    constexpr int HB = N + (SKIP ? N - 2 : 0);                // decision bits per column and lane
    constexpr int CPW = 32 / HB;
    constexpr int PF = 4;
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
    const char* nllb = static_cast<const char*>(a.nll) + f0 * a.S * (int64_t)sizeof(ET);
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
            if (t >= Tmax) break;
            if (t < T) {
                double e[N];
#pragma unroll
                for (int s = 0; s < N; ++s) e[s] = (double)ring[p][s];
                if (t + PF < T) {
                    const char* colp = nllb + (int64_t)(t + PF) * rowb;
#pragma unroll
                    for (int s = 0; s < N; ++s) ring[p][s] = *reinterpret_cast<const ET*>(colp + sto[s]);
                }
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
    const uint8_t* arcp = a.seqwords[g->word[bk]].arcs;
    while (j != 0) {
        const int64_t want = (int64_t)(j / CPW) * 16 + bk;
        if (want != key) { cw = bpu[want]; key = want; }
        const uint32_t hb = (cw >> ((CPW - 1 - j % CPW) * HB)) & ((1u << HB) - 1u);
        const int arcs = arcp[bs];
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
            arcp = a.seqwords[g->word[bk]].arcs;
            path[len++] = make_int2(bk * (N + 1) + 1 + bs, j);
        }
    }
    if (flag) atomicOr(a.flag, flag);
    a.path_len[u] = (int32_t)len;
}

// ---------------------------------------------------------------------------------------------------------------------
// Forward-backward over the same graphs (definition: fb_kernel, gh_fb.hip -- same arcs, same same-column rule, start row
// in column 0, end rows in the last column, min -> -logsumexp).  Log domain, fp64; alpha columns go to HBM scratch
// [T, K, N] and come back in the backward sweep, beta lives in registers.  Inside a column the backward order is:
// states 0..N-2 of every layer (they only read the next column), the non-emitting rows (one row_shl:1), then the last
// states.  gamma is added into occ[frame, state] with double atomics (a word may stand in several layers of one
// transcript); the expected self transitions are kept per lane and added into one of GH_FBSEQ_XI_PARTS partial vectors.
__device__ __forceinline__ double sq_lse2(double x, double y) {
    const double m = fmax(x, y);
    if (m == -INFINITY) return -INFINITY;
    return m + log1p(exp(fmin(x, y) - m));
}
__device__ __forceinline__ double sq_lse3(double x, double y, double z) {
    const double m = fmax(fmax(x, y), z);
    if (m == -INFINITY) return -INFINITY;
    return m + log(exp(x - m) + exp(y - m) + exp(z - m));
}

template <typename ET, int N, bool SKIP>
__global__ __launch_bounds__(64) void fb_seq_kernel(gh_fbseq_args a, int64_t slot_end) {
    constexpr int PF = 2;
    const int lane = threadIdx.x, kk = lane >> 4, k = lane & 15;
    const int64_t slot = a.slot0 + (int64_t)blockIdx.x * 4 + kk;
    const bool has_utt = slot < slot_end;
    const int64_t u = has_utt ? (a.perm ? a.perm[slot] : slot) : 0;
    const gh_seqgraph* g = a.graphs + ((has_utt && a.utt_lat) ? a.utt_lat[u] : 0);
    const int K = g->K;
    const int64_t f0 = has_utt ? a.utt_off[u] : 0;
    const int T = has_utt ? (int)(a.utt_off[u + 1] - f0) : 0;
    const double INF = INFINITY, NEG = -INFINITY;
    int Tmax = T;
    Tmax = max(Tmax, __shfl_xor(Tmax, 16));
    Tmax = max(Tmax, __shfl_xor(Tmax, 32));
    const bool lact = has_utt && k < K;
    const gh_seqword* wd = a.words + (lact ? g->word[k] : 0);
    double c0[N], c1[N], c2[N];
    int st[N];
    unsigned endm = 0;
#pragma unroll
    for (int s = 0; s < N; ++s) {
        c0[s] = lact ? wd->c0[s] : INF;
        c1[s] = lact ? wd->c1[s] : INF;
        c2[s] = (SKIP && lact) ? wd->c2[s] : INF;
        st[s] = lact ? wd->state[s] : 0;
        if (lact && a.end_slot[g->row_base + k * (N + 1) + 1 + s] >= 0) endm |= 1u << s;
    }
    const double cin = lact ? wd->cin : INF, cout = lact ? wd->cout : INF;
    const ET* nll = static_cast<const ET*>(a.nll) + f0 * a.S;
    double* alpha = a.alpha_scratch + (has_utt ? a.scratch_off[slot] : 0) + (int64_t)k * N;
    const int64_t astride = (int64_t)K * N;
    if (T <= 0) {                                             // (a whole row: no shuffle below involves it)
        if (has_utt && k == 0 && a.logp) a.logp[u] = NEG;
    }
    // ---- forward ----
    ET ring[PF][N];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int s = 0; s < N; ++s) ring[p][s] = (lact && p < T) ? nll[(int64_t)p * a.S + st[s]] : ET(0);
    double al[N];
#pragma unroll
    for (int s = 0; s < N; ++s) al[s] = NEG;
    for (int t0 = 0; t0 < Tmax; t0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int t = t0 + p;
            if (t >= Tmax) break;
            if (t < T) {
                double e[N];
#pragma unroll
                for (int s = 0; s < N; ++s) e[s] = (double)ring[p][s];
                if (lact && t + PF < T) {
#pragma unroll
                    for (int s = 0; s < N; ++s) ring[p][s] = nll[(int64_t)(t + PF) * a.S + st[s]];
                }
                const double self0 = al[0] - c0[0];
#pragma unroll
                for (int s = N - 1; s >= 1; --s) {
                    const double a0 = al[s] - c0[s], a1 = al[s - 1] - c1[s];
                    if (SKIP && s >= 2) al[s] = sq_lse3(a0, a1, al[s - 2] - c2[s]) - e[s];
                    else al[s] = sq_lse2(a0, a1) - e[s];
                }
                const double nin = row_shr1(al[N - 1] - cout, (t == 0) ? 0.0 : NEG);   // lane 0: the start row
                al[0] = sq_lse2(self0, nin - cin) - e[0];
                if (lact) {
#pragma unroll
                    for (int s = 0; s < N; ++s) alpha[(int64_t)t * astride + s] = al[s];
                }
            }
        }
    }
    double lp = NEG;
#pragma unroll
    for (int s = 0; s < N; ++s) if (endm >> s & 1) lp = sq_lse2(lp, al[s]);
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) lp = sq_lse2(lp, __shfl_xor(lp, o));
    const double logp = lp;
    if (has_utt && k == 0 && T > 0 && a.logp) a.logp[u] = logp;
    if (!a.occ && !a.self_xi_parts) return;
    // ---- backward (column t = T - 1 - i of this row's utterance) ----
    double be[N], en[N], xi[N];
#pragma unroll
    for (int s = 0; s < N; ++s) { be[s] = NEG; en[s] = 0.0; xi[s] = 0.0; }
    ET ering[PF][N];
    double aring[PF][N];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int s = 0; s < N; ++s) {
            const int t = T - 1 - p;
            ering[p][s] = (lact && t >= 0) ? nll[(int64_t)t * a.S + st[s]] : ET(0);
            aring[p][s] = (lact && t >= 0) ? alpha[(int64_t)t * astride + s] : NEG;
        }
    for (int i0 = 0; i0 < Tmax; i0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int i = i0 + p;
            if (i >= Tmax) break;
            if (i < T) {
                const int t = T - 1 - i;
                double e[N], av[N];
#pragma unroll
                for (int s = 0; s < N; ++s) { e[s] = (double)ering[p][s]; av[s] = aring[p][s]; }
                if (lact && t - PF >= 0) {
#pragma unroll
                    for (int s = 0; s < N; ++s) {
                        ering[p][s] = nll[(int64_t)(t - PF) * a.S + st[s]];
                        aring[p][s] = alpha[(int64_t)(t - PF) * astride + s];
                    }
                }
                double w[N], nb[N];
#pragma unroll
                for (int s = 0; s < N; ++s) w[s] = be[s] - en[s];      // column t + 1 seen from column t, before the arc cost
#pragma unroll
                for (int s = 0; s < N - 1; ++s) {
                    const double b0 = w[s] - c0[s], b1 = w[s + 1] - c1[s + 1];
                    if (SKIP && s + 2 < N) nb[s] = sq_lse3(b0, b1, w[s + 2] - c2[s + 2]);
                    else nb[s] = sq_lse2(b0, b1);
                    if (i == 0 && (endm >> s & 1)) nb[s] = sq_lse2(nb[s], 0.0);
                }
                // the non-emitting row in front of layer k + 1 (its only successor: that layer's first state, same column)
                const double nesb = row_shl1(nb[0] - cin - e[0], NEG);
                nb[N - 1] = sq_lse2(w[N - 1] - c0[N - 1], nesb - cout);
                if (i == 0 && (endm >> (N - 1) & 1)) nb[N - 1] = sq_lse2(nb[N - 1], 0.0);
                if (lact) {
#pragma unroll
                    for (int s = 0; s < N; ++s) {
                        if (a.self_xi_parts && i > 0) {             // xi_t(s -> s) = alpha_t(s) a_ss b_s(x_{t+1}) beta_{t+1}(s) / P
                            const double x = exp(av[s] + (w[s] - c0[s]) - logp);
                            if (x == x) xi[s] += x;
                        }
                        if (a.occ) {
                            double gm = exp(av[s] + nb[s] - logp);
                            if (gm == gm && gm != 0.0) unsafeAtomicAdd(a.occ + (f0 + t) * a.S + st[s], gm);
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < N; ++s) { be[s] = nb[s]; en[s] = e[s]; }
            }
        }
    }
    if (a.self_xi_parts && lact) {
        double* part = a.self_xi_parts + (size_t)(blockIdx.x % GH_FBSEQ_XI_PARTS) * a.S;
#pragma unroll
        for (int s = 0; s < N; ++s) if (xi[s] != 0.0) unsafeAtomicAdd(part + st[s], xi[s]);
    }
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
        default: gh_set_error("gh_viterbi: sequence form with %d states per word", N); return GH_ERR_UNSUPPORTED;
    }
#undef GH_SB
    GH_HIP(hipGetLastError());
    return GH_OK;
}

int gh_launch_fb_seq(gh_ctx* ctx, const gh_fbseq_args& a, int N, int skip, int64_t u_begin, int64_t n_utts, bool f64) {
    if (n_utts <= 0) return GH_OK;
    gh_fbseq_args b = a;
    b.slot0 = u_begin;
    const dim3 grid((unsigned)((n_utts + 3) / 4)), blk(64);
    const int64_t slot_end = u_begin + n_utts;
#define GH_FS(ET, NN, SK) hipLaunchKernelGGL((fb_seq_kernel<ET, NN, SK>), grid, blk, 0, ctx->stream, b, slot_end)
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
