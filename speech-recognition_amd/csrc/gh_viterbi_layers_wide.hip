// WIDE layer form: the K-layer word lattice of gh_viterbi_layers.hip with MORE THAN 16 WORDS per layer (17 .. 64 -- a
// vocabulary the size of BASELINE configs[3]'s 64 models decoded continuously; build_state_sequences,
// continuous_speech.py:13-53, has no limit).  Same semantics as every other lattice kernel (decode_hmm_states,
// decode.py:80-146): candidates in ascending origin order with a strict '<', arcs that touch a non-emitting row read the SAME
// column, start only in cell (0, 0), last of equal end points, path without the end cell.
//
// gfx950 mapping -- one wave per utterance, no LDS, no barrier -- turned by a quarter against the narrow kernel:
//   * lane = WORD (up to 64), and a lane keeps the N state costs of its word in ALL layers in registers (two sets of four
//     layers: K <= 8, N <= 8 -- 64 doubles).  Inside a word nothing crosses lanes, as before;
//   * the layers of a column run one after the other (layer k's first states take the non-emitting row behind layer k - 1
//     of the SAME column): per layer the states N-1 .. 1 from the previous column, then the non-emitting row = the minimum of
//     the words' last states over the WHOLE wave -- four row_ror steps inside the 16-lane rows, four v_readlane pairs across
//     them -- its `cand == min` bit, then state 0 from (self, non-emitting row in front);
//   * decision bits: N + 1 (+ N - 2 with skip arcs) per layer, four layers to a 64-bit word: two words per column and lane
//     (<= 60 bits each), [column][set][lane];
//   * the emissions of the word's N states are loaded once per column and serve every layer (the layers share the states).
// The back-trace is its own kernel (one lane per utterance, as lattice_backtrace_kernel).
#include "gh_internal.h"
#include "gh_viterbi.h"

namespace {

__device__ __forceinline__ double vmin(double a, double b) {   // IEEE minNum in ONE instruction: a NaN operand loses
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int CTRL> __device__ __forceinline__ double row_rot(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// minimum over the 64 lanes, left in every lane: inside the 16-lane rows by row_ror:1,2,4,8, across them through the scalar unit
__device__ __forceinline__ double wave_min(double v) {
    v = vmin(v, row_rot<0x121>(v));
    v = vmin(v, row_rot<0x122>(v));
    v = vmin(v, row_rot<0x124>(v));
    v = vmin(v, row_rot<0x128>(v));
    auto at = [&](int l) { return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l)); };
    return vmin(vmin(at(0), at(16)), vmin(at(32), at(48)));
}

// word = 2 * word + bit, the bit taken from a compare's lane mask (two v_addc_co_u32: the low half's carry goes on)
__device__ __forceinline__ void push_bit(uint64_t& word, unsigned long long mask) {
    uint32_t lo = (uint32_t)word, hi = (uint32_t)(word >> 32);
    unsigned long long c1, c2;
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(lo), "=s"(c1) : "v"(lo), "s"(mask));
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(hi), "=s"(c2) : "v"(hi), "s"(c1));
    word = ((uint64_t)hi << 32) | lo;
}

constexpr int WL_SETS = 2;                 // sets of four layers: K <= 8

template <typename ET, int N, bool SKIP, bool WANT_BP>
__global__ __launch_bounds__(64) void viterbi_layers_wide_kernel(gh_layers_args a) {
    constexpr int KM = 4 * WL_SETS;
    constexpr int HB = N + 1 + (SKIP ? N - 2 : 0);            // decision bits per column, lane and layer
    constexpr int PF = 2;                                     // columns of emissions in flight
    static_assert(4 * HB <= 64, "four layers' decision bits share a 64-bit word");
    const int lane = threadIdx.x, w = lane;
    const gh_layerform* __restrict__ lf = a.lf;
    const int K = lf->K, W = lf->W, P = lf->P;
    const int64_t slot = a.slot0 + blockIdx.x;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int64_t f0 = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - f0);
    const double INF = INFINITY;
    if (T <= 0) {
        if (lane == 0) {
            if (a.best_end) a.best_end[u] = -1;
            if (a.path_len) a.path_len[u] = 0;
        }
        return;
    }
    const bool wact = w < W;
    const int wc = wact ? w : 0;
    double c0[N], c1[N], c2[N];
    unsigned sto[N];                                          // byte offset of the state's emission inside a matrix row
#pragma unroll
    for (int s = 0; s < N; ++s) {
        c0[s] = wact ? lf->c0[wc][s] : INF;
        c1[s] = wact ? lf->c1[wc][s] : INF;
        c2[s] = (SKIP && wact) ? lf->c2[wc][s] : INF;
        sto[s] = (unsigned)lf->state[wc][s] * (unsigned)sizeof(ET);
    }
    const double cin = wact ? lf->cin[wc] : INF, cout = wact ? lf->cout[wc] : INF;
    const char* nllb = static_cast<const char*>(a.nll) + f0 * a.S * (int64_t)sizeof(ET);   // wave-uniform
    const int64_t rowb = (int64_t)a.S * (int64_t)sizeof(ET);
    ET ring[PF][N];
#pragma unroll
    for (int k = 0; k < PF; ++k)
#pragma unroll
        for (int s = 0; s < N; ++s)
            ring[k][s] = (k < T) ? *reinterpret_cast<const ET*>(nllb + k * rowb + sto[s]) : ET(0);
    double prev[KM][N];
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int s = 0; s < N; ++s) prev[k][s] = INF;
    uint64_t* bp = WANT_BP ? reinterpret_cast<uint64_t*>(a.bp + a.bp_off[slot]) + lane : nullptr;
    __builtin_amdgcn_s_waitcnt(0x0F70);                        // (the ring's first fill drained: see viterbi_layers_kernel)
    auto column = [&](int t, const ET (&ev)[N]) {
        double e[N];
#pragma unroll
        for (int s = 0; s < N; ++s) e[s] = (double)ev[s];
        double carry = (t == 0) ? 0.0 : INF;                  // the start row: cost 0 in column 0 only (decode.py:99-101)
        uint64_t word[WL_SETS];
#pragma unroll
        for (int q = 0; q < WL_SETS; ++q) word[q] = 0;
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            const bool on = k < K;                            // (wave-uniform; a layer behind the last one stays +inf)
            const double base0 = c0[0] + prev[k][0];          // state 0 from its own previous column
#pragma unroll
            for (int s = N - 1; s >= 1; --s) {                // in place, descending: the neighbours are still old
                const double v0 = c0[s] + prev[k][s];
                const double v1 = c1[s] + prev[k][s - 1];
                double best;
                if (SKIP && s >= 2) {                         // ascending origin order: s-2, s-1, s; strict '<'
                    const double v2 = c2[s] + prev[k][s - 2];
                    const bool b_a = v1 < v2;
                    const double m = vmin(v1, v2);
                    const bool b_b = v0 < m;
                    best = vmin(v0, m);
                    if (WANT_BP) { push_bit(word[k >> 2], __ballot(b_a)); push_bit(word[k >> 2], __ballot(b_b)); }
                } else {
                    const bool b = v0 < v1;
                    best = vmin(v0, v1);
                    if (WANT_BP) push_bit(word[k >> 2], __ballot(b));
                }
                prev[k][s] = vmin(best + e[s], INF);          // min(inf, nan) keeps inf (decode.py:124)
            }
            // the non-emitting row behind this layer: minimum over ALL words' last states, same column
            const double cand = prev[k][N - 1] + (on ? cout : INF);
            const double rm = wave_min(cand);
            if (WANT_BP) push_bit(word[k >> 2], __ballot(cand == rm));
            // state 0: the non-emitting row in front of the layer (lower row index: it wins ties) against the self arc
            const double cn = carry + (on ? cin : INF);
            const bool b0 = base0 < cn;
            if (WANT_BP) push_bit(word[k >> 2], __ballot(b0));
            prev[k][0] = vmin(vmin(base0, cn) + e[0], INF);
            carry = rm;
        }
        if (WANT_BP) {
#pragma unroll
            for (int q = 0; q < WL_SETS; ++q) bp[((int64_t)t * WL_SETS + q) * 64] = word[q];
        }
    };
    int t0 = 0;
    for (; t0 + PF <= T; t0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int t = t0 + k;
            column(t, ring[k]);
            const int tn = (t + PF < T) ? t + PF : T - 1;
            const char* colp = nllb + (int64_t)tn * rowb;
#pragma unroll
            for (int s = 0; s < N; ++s) ring[k][s] = *reinterpret_cast<const ET*>(colp + sto[s]);
        }
    }
#pragma unroll
    for (int k = 0; k < PF - 1; ++k)
        if (t0 + k < T) column(t0 + k, ring[k]);

    // ---- end costs, end selection ('>=': the last of equal minima, decode.py:129-134) ----
    double best_v = INF;
    int best_slot = -1;
#pragma unroll
    for (int k = 0; k < KM; ++k)
#pragma unroll
        for (int s = 0; s < N; ++s) {
            if (wact && k < K) {
                const int r = k * (P + 1) + 1 + w * N + s;
                const int es = a.end_slot[r];
                if (es >= 0) {
                    const double v = prev[k][s];
                    if (a.end_cost) a.end_cost[u * a.n_end + es] = v;
                    if (v < best_v || (v == best_v && es > best_slot)) { best_v = v; best_slot = es; }
                }
            }
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(best_v, o);
        const int os = __shfl_xor(best_slot, o);
        if (ov < best_v || (ov == best_v && os > best_slot)) { best_v = ov; best_slot = os; }
    }
    if (lane == 0 && a.best_end) a.best_end[u] = best_slot;
}

// Back-trace (decode.py:143-145), ONE LANE PER UTTERANCE (see lattice_backtrace_kernel, gh_viterbi_layers.hip: 64 walks per
// wave on the vector units).  The decision word of (column j, set of layer bk, word bw) is bpu[(j SETS + (bk >> 2)) 64 + bw];
// layer bk's bits sit (3 - (bk & 3)) HB bits up.  MODE 0: the (row, column) path; MODE 1: only the label sequence.
template <int N, bool SKIP, int MODE>
__global__ __launch_bounds__(64) void lattice_backtrace_wide_kernel(gh_layers_args a, int64_t slot_end) {
    constexpr int HB = N + 1 + (SKIP ? N - 2 : 0);
    __shared__ uint8_t s_arcs[GH_LAYERS_MAXW * GH_LAYERS_MAXN];
    const gh_layerform* __restrict__ lf = a.lf;
    for (int i = threadIdx.x; i < GH_LAYERS_MAXW * GH_LAYERS_MAXN; i += 64) s_arcs[i] = (&lf->arcs[0][0])[i];
    __syncthreads();
    const int W = lf->W, P = lf->P;
    const int64_t slot = a.slot0 + (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (slot >= slot_end) return;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int T = (int)(a.utt_off[u + 1] - a.utt_off[u]);
    const int be = a.best_end[u];
    int32_t* out_n = MODE == 0 ? a.path_len : a.n_labels;
    if (T <= 1 || be < 0) { out_n[u] = 0; return; }
    auto row_of = [&](int k, int ww, int ss) { return k * (P + 1) + 1 + ww * N + ss; };
    int bk, bw, bs;
    {
        const int r = a.end_rows[be];
        bk = (r - 1) / (P + 1);
        const int pos = (r - 1) % (P + 1);
        bw = pos / N;
        bs = pos % N;
    }
    const uint64_t* bpu = reinterpret_cast<const uint64_t*>(a.bp + a.bp_off[slot]);
    int32_t* path = MODE == 0 ? a.path + 2 * a.path_off[u] : nullptr;
    int32_t* labs = MODE == 1 ? a.labels + a.label_off[u] : nullptr;
    const int64_t cap = MODE == 0 ? a.path_off[u + 1] - a.path_off[u] : a.label_off[u + 1] - a.label_off[u];
    int64_t len = 0;
    int prev_label = -1;                                      // MODE 1: label of the cell visited last
    int j = T - 1, kind = 0, kn = 0;                          // kind 0 emitting (bk, bw, bs); 1 non-emitting row kn
    int flag = 0;
    int64_t key = -1;                                         // index of the decision word held in `cw`
    uint64_t cw = 0, pw = 0;                                  // current word, and the same (set, word) one column earlier
    auto visit = [&](int row, int col) {
        if (MODE == 0) {
            if (len >= cap) { flag |= 4; return; }
            reinterpret_cast<int2*>(path)[len] = make_int2(row, col);
            ++len;
        } else {
            const int l = a.row_label[row];
            if (prev_label >= 0 && l < 0) {
                if (len >= cap) { flag |= 8; return; }
                labs[cap - 1 - len] = prev_label;
                ++len;
            }
            prev_label = l;
        }
    };
    while (j != 0 && !flag) {
        if (kind == 0) {
            const int64_t want = ((int64_t)j * WL_SETS + (bk >> 2)) * 64 + bw;
            if (want != key) {
                if (want == key - WL_SETS * 64) cw = pw; else cw = bpu[want];
                key = want;
                pw = (j > 0) ? bpu[want - WL_SETS * 64] : 0ull;
            }
            const uint32_t hb = (uint32_t)((cw >> ((3 - (bk & 3)) * HB)) & ((1ull << HB) - 1ull));
            const int arcs = s_arcs[bw * GH_LAYERS_MAXN + bs];
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
                visit(row_of(bk, bw, bs), j);
            } else {
                const bool self_better = hb & 1;
                const bool take_self = (self_better && (arcs & 1)) || !(arcs & 8);
                if (take_self && !(arcs & 1)) { flag |= 2; break; }
                if (take_self) { --j; visit(row_of(bk, bw, 0), j); }
                else { kind = 1; kn = bk; visit(kn * (P + 1), j); }          // the non-emitting row in front of the layer
            }
        } else {
            if (kn == 0) { flag |= 2; break; }                                 // the start row has no origin
            const int kp = kn - 1;
            const int eq_shift = (3 - (kp & 3)) * HB + 1;
            const ulonglong2* rowp = reinterpret_cast<const ulonglong2*>(bpu + ((int64_t)j * WL_SETS + (kp >> 2)) * 64);
            int found = -1;
            for (int q2 = (W + 1) / 2 - 1; q2 >= 0; --q2) {                    // the words' decision words: 16 bytes at a time
                const ulonglong2 v = rowp[q2];
                if ((v.y >> eq_shift) & 1ull) found = 2 * q2 + 1;
                if ((v.x >> eq_shift) & 1ull) found = 2 * q2;
            }
            if (found < 0 || found >= W) { flag |= 2; break; }                 // lowest word = lowest origin row (np.argmin)
            bw = found;
            bk = kp;
            bs = N - 1;
            kind = 0;
            visit(row_of(bk, bw, bs), j);
        }
    }
    if (flag) atomicOr(a.flag, flag);
    if (MODE == 1) {
        if (!flag && prev_label >= 0) {
            if (len >= cap) atomicOr(a.flag, 8);
            else { labs[cap - 1 - len] = prev_label; ++len; }
        }
        for (int64_t i = 0; i < len; ++i) labs[i] = labs[cap - len + i];       // to the front, start -> end order
    }
    out_n[u] = (int32_t)len;
}

// ---------------------------------------------------------------------------------------------------------------------
// WIDE LOOP form: the word-loop grammar (gh_layerform.loop, see viterbi_loop_kernel) with 17 .. 64 words: ONE utterance per
// wave (the narrow kernel holds four, a DPP row each), lane = word, the N states of the word in registers.  Per column: the
// states from the previous column, the loop row = the minimum over ALL words' last states of this column, the first states
// from (start row, loop row, self) in that candidate order.  N + 2 (+ N - 2 with skip arcs) decision bits per column and
// lane: one 32-bit word, [column][lane].
template <typename ET, int N, bool SKIP, bool WANT_BP>
__global__ __launch_bounds__(64) void viterbi_loop_wide_kernel(gh_layers_args a) {
    constexpr int PF = 4;
    const int lane = threadIdx.x, w = lane;
    const gh_layerform* __restrict__ lf = a.lf;
    const int W = lf->W, Lr = lf->loop_row;
    const int64_t slot = a.slot0 + blockIdx.x;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int64_t f0 = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - f0);
    const double INF = INFINITY;
    const bool wact = w < W;
    const int wc = wact ? w : 0;
    double c0[N], c1[N], c2[N];
    unsigned sto[N];
#pragma unroll
    for (int s = 0; s < N; ++s) {
        c0[s] = wact ? lf->c0[wc][s] : INF;
        c1[s] = wact ? lf->c1[wc][s] : INF;
        c2[s] = (SKIP && wact) ? lf->c2[wc][s] : INF;
        sto[s] = (unsigned)lf->state[wc][s] * (unsigned)sizeof(ET);
    }
    const double cin = wact ? lf->cin[wc] : INF, cin0 = wact ? lf->cin0[wc] : INF, cout = wact ? lf->cout[wc] : INF;
    const char* nllb = static_cast<const char*>(a.nll) + (T > 0 ? f0 : 0) * a.S * (int64_t)sizeof(ET);
    const int64_t rowb = (int64_t)a.S * (int64_t)sizeof(ET);
    ET ring[PF][N];
#pragma unroll
    for (int k = 0; k < PF; ++k)
#pragma unroll
        for (int s = 0; s < N; ++s)
            ring[k][s] = (k < T) ? *reinterpret_cast<const ET*>(nllb + k * rowb + sto[s]) : ET(0);
    double prev[N];
#pragma unroll
    for (int s = 0; s < N; ++s) prev[s] = INF;
    uint32_t* bp = WANT_BP ? reinterpret_cast<uint32_t*>(a.bp + a.bp_off[slot]) + lane : nullptr;
    __builtin_amdgcn_s_waitcnt(0x0F70);
    auto push32 = [](uint32_t& word, unsigned long long mask) {
        unsigned long long carry_out;
        asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(word), "=s"(carry_out) : "v"(word), "s"(mask));
    };
    auto column = [&](int t, const ET (&ev)[N]) {
        double e[N];
#pragma unroll
        for (int s = 0; s < N; ++s) e[s] = (double)ev[s];
        uint32_t word = 0;
        const double base0 = c0[0] + prev[0];
#pragma unroll
        for (int s = N - 1; s >= 1; --s) {
            const double v0 = c0[s] + prev[s];
            const double v1 = c1[s] + prev[s - 1];
            double best;
            if (SKIP && s >= 2) {
                const double v2 = c2[s] + prev[s - 2];
                const bool b_a = v1 < v2;
                const double m = vmin(v1, v2);
                const bool b_b = v0 < m;
                best = vmin(v0, m);
                if (WANT_BP) { push32(word, __ballot(b_a)); push32(word, __ballot(b_b)); }
            } else {
                const bool b = v0 < v1;
                best = vmin(v0, v1);
                if (WANT_BP) push32(word, __ballot(b));
            }
            prev[s] = vmin(best + e[s], INF);
        }
        // the loop row: minimum over the words' last states of THIS column
        const double cand = prev[N - 1] + cout;
        const double rm = wave_min(cand);
        if (WANT_BP) push32(word, __ballot(cand == rm));
        // state 0: start row (row 0), loop row, self -- ascending origin, strict '<'
        const double cs = ((t == 0) ? 0.0 : INF) + cin0;
        const double cl = rm + cin;
        const bool b_l = cl < cs;
        const double m2 = vmin(cl, cs);
        const bool b_s = base0 < m2;
        if (WANT_BP) { push32(word, __ballot(b_l)); push32(word, __ballot(b_s)); }
        prev[0] = vmin(vmin(base0, m2) + e[0], INF);
        if (WANT_BP) bp[(int64_t)t * 64] = word;
    };
    int t0 = 0;
    for (; t0 + PF <= T; t0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int t = t0 + k;
            column(t, ring[k]);
            const int tn = (t + PF < T) ? t + PF : T - 1;
            const char* colp = nllb + (int64_t)tn * rowb;
#pragma unroll
            for (int s = 0; s < N; ++s) ring[k][s] = *reinterpret_cast<const ET*>(colp + sto[s]);
        }
    }
#pragma unroll
    for (int k = 0; k < PF - 1; ++k)
        if (t0 + k < T) column(t0 + k, ring[k]);
    // ---- end costs + end selection ('>=': the last of equal minima) ----
    double best_v = INF;
    int best_slot = -1;
#pragma unroll
    for (int s = 0; s < N; ++s) {
        if (wact) {
            const int r = s == 0 ? Lr + 1 + w : 1 + w * (N - 1) + (s - 1);
            const int es = a.end_slot[r];
            if (es >= 0) {
                const double v = T > 0 ? prev[s] : INF;
                if (a.end_cost) a.end_cost[u * a.n_end + es] = v;
                if (v < best_v || (v == best_v && es > best_slot)) { best_v = v; best_slot = es; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(best_v, o);
        const int os = __shfl_xor(best_slot, o);
        if (ov < best_v || (ov == best_v && os > best_slot)) { best_v = ov; best_slot = os; }
    }
    if (lane == 0 && a.best_end) a.best_end[u] = T > 0 ? best_slot : -1;
}

// its back-trace: one lane per utterance, the decision word of (column j, word bw) is bpu[j 64 + bw]
template <int N, bool SKIP, int MODE>
__global__ __launch_bounds__(64) void loop_backtrace_wide_kernel(gh_layers_args a, int64_t slot_end) {
    constexpr int HB = N + 2 + (SKIP ? N - 2 : 0);
    __shared__ uint8_t s_arcs[GH_LAYERS_MAXW * GH_LAYERS_MAXN];
    const gh_layerform* __restrict__ lf = a.lf;
    for (int i = threadIdx.x; i < GH_LAYERS_MAXW * GH_LAYERS_MAXN; i += 64) s_arcs[i] = (&lf->arcs[0][0])[i];
    __syncthreads();
    const int W = lf->W, Lr = lf->loop_row;
    const int64_t slot = a.slot0 + (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (slot >= slot_end) return;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int T = (int)(a.utt_off[u + 1] - a.utt_off[u]);
    const int be = a.best_end[u];
    int32_t* out_n = MODE == 0 ? a.path_len : a.n_labels;
    if (T <= 1 || be < 0) { out_n[u] = 0; return; }
    auto row_of = [&](int ww, int ss) { return ss == 0 ? Lr + 1 + ww : 1 + ww * (N - 1) + (ss - 1); };
    int bw, bs;
    {
        const int r = a.end_rows[be];
        if (r > Lr) { bw = r - Lr - 1; bs = 0; } else { bw = (r - 1) / (N - 1); bs = (r - 1) % (N - 1) + 1; }
    }
    const uint32_t* bpu = reinterpret_cast<const uint32_t*>(a.bp + a.bp_off[slot]);
    int32_t* path = MODE == 0 ? a.path + 2 * a.path_off[u] : nullptr;
    int32_t* labs = MODE == 1 ? a.labels + a.label_off[u] : nullptr;
    const int64_t cap = MODE == 0 ? a.path_off[u + 1] - a.path_off[u] : a.label_off[u + 1] - a.label_off[u];
    int64_t len = 0;
    int prev_label = -1;
    int j = T - 1, kind = 0;                                  // kind 0 emitting (bw, bs); 1 the loop row; 2 the start row
    int flag = 0;
    auto visit = [&](int row, int col) {
        if (MODE == 0) {
            if (len >= cap) { flag |= 4; return; }
            reinterpret_cast<int2*>(path)[len] = make_int2(row, col);
            ++len;
        } else {
            const int l = a.row_label[row];
            if (prev_label >= 0 && l < 0) {
                if (len >= cap) { flag |= 8; return; }
                labs[cap - 1 - len] = prev_label;
                ++len;
            }
            prev_label = l;
        }
    };
    while (j != 0 && !flag) {
        if (kind == 0) {
            const uint32_t hb = bpu[(int64_t)j * 64 + bw] & (uint32_t)((1ull << HB) - 1ull);
            const int arcs = s_arcs[bw * GH_LAYERS_MAXN + bs];
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
                if (!((arcs >> code) & 1)) code = (arcs & 4) ? 2 : (arcs & 2) ? 1 : (arcs & 1) ? 0 : -1;
                if (code < 0) { flag |= 2; break; }
                bs -= code;
                --j;
                visit(row_of(bw, bs), j);
            } else {
                // candidates in ascending origin order: start row (arcs bit4), loop row (bit3), self (bit0)
                const int b_l = (hb >> 1) & 1, b_s = hb & 1;
                int pick = b_s ? 0 : (b_l ? 3 : 4);
                if (!((arcs >> pick) & 1)) pick = (arcs & 16) ? 4 : (arcs & 8) ? 3 : (arcs & 1) ? 0 : -1;
                if (pick < 0) { flag |= 2; break; }
                if (pick == 0) { --j; visit(row_of(bw, 0), j); }
                else if (pick == 3) { kind = 1; visit(Lr, j); }
                else { kind = 2; visit(0, j); }
            }
        } else if (kind == 1) {
            const uint4* rowp = reinterpret_cast<const uint4*>(bpu + (int64_t)j * 64);
            int found = -1;
            for (int q4 = (W + 3) / 4 - 1; q4 >= 0; --q4) {                    // the words' decision words of this column
                const uint4 v = rowp[q4];
                if ((v.w >> 2) & 1u) found = 4 * q4 + 3;
                if ((v.z >> 2) & 1u) found = 4 * q4 + 2;
                if ((v.y >> 2) & 1u) found = 4 * q4 + 1;
                if ((v.x >> 2) & 1u) found = 4 * q4;
            }
            if (found < 0 || found >= W) { flag |= 2; break; }                 // lowest word = lowest origin row (np.argmin)
            bw = found;
            bs = N - 1;
            kind = 0;
            visit(row_of(bw, bs), j);
        } else {
            flag |= 2;                                                          // the start row, reached in a column > 0
            break;
        }
    }
    if (flag) atomicOr(a.flag, flag);
    if (MODE == 1) {
        if (!flag && prev_label >= 0) {
            if (len >= cap) atomicOr(a.flag, 8);
            else { labs[cap - 1 - len] = prev_label; ++len; }
        }
        for (int64_t i = 0; i < len; ++i) labs[i] = labs[cap - len + i];
    }
    out_n[u] = (int32_t)len;
}

}  // namespace

// decision words of one utterance of T frames in uint16 units: two 64-bit words per column and lane
size_t gh_layers_wide_bp_entries(int64_t T) { return (size_t)T * WL_SETS * 64 * 4; }
// ... wide loop form: one 32-bit word per column and lane
size_t gh_loop_wide_bp_entries(int64_t T) { return (size_t)T * 64 * 2; }

#define GH_LW_CASES(MACRO)                       \
    switch (f.N) {                               \
        case 2: MACRO(2, false); break;          \
        case 3: if (f.skip) MACRO(3, true); else MACRO(3, false); break; \
        case 4: if (f.skip) MACRO(4, true); else MACRO(4, false); break; \
        case 5: if (f.skip) MACRO(5, true); else MACRO(5, false); break; \
        case 6: if (f.skip) MACRO(6, true); else MACRO(6, false); break; \
        case 7: if (f.skip) MACRO(7, true); else MACRO(7, false); break; \
        case 8: if (f.skip) MACRO(8, true); else MACRO(8, false); break; \
        default: gh_set_error("gh_viterbi: wide layer form with %d states per word", f.N); return GH_ERR_UNSUPPORTED; \
    }

int gh_launch_viterbi_layers_wide(gh_ctx* ctx, const gh_layers_args& a, const gh_layerform& f, int64_t u_begin, int64_t n_utts,
                                  bool f64, bool want_path) {
    if (n_utts <= 0) return GH_OK;
    gh_layers_args b = a;
    b.slot0 = u_begin;
    const dim3 grid((unsigned)n_utts), blk(64);
    if (f.loop) {
#define GH_LPW(NN, SK)                                                                                                      \
    do {                                                                                                                    \
        if (f64) {                                                                                                          \
            if (want_path) hipLaunchKernelGGL((viterbi_loop_wide_kernel<double, NN, SK, true>), grid, blk, 0, ctx->stream, b);    \
            else hipLaunchKernelGGL((viterbi_loop_wide_kernel<double, NN, SK, false>), grid, blk, 0, ctx->stream, b);             \
        } else {                                                                                                            \
            if (want_path) hipLaunchKernelGGL((viterbi_loop_wide_kernel<float, NN, SK, true>), grid, blk, 0, ctx->stream, b);     \
            else hipLaunchKernelGGL((viterbi_loop_wide_kernel<float, NN, SK, false>), grid, blk, 0, ctx->stream, b);              \
        }                                                                                                                   \
    } while (0)
        GH_LW_CASES(GH_LPW)
#undef GH_LPW
        GH_HIP(hipGetLastError());
        return GH_OK;
    }
    if (f.K > 4 * WL_SETS) { gh_set_error("gh_viterbi: wide layer form with %d layers", f.K); return GH_ERR_UNSUPPORTED; }
#define GH_LW(NN, SK)                                                                                                       \
    do {                                                                                                                    \
        if (f64) {                                                                                                          \
            if (want_path) hipLaunchKernelGGL((viterbi_layers_wide_kernel<double, NN, SK, true>), grid, blk, 0, ctx->stream, b);  \
            else hipLaunchKernelGGL((viterbi_layers_wide_kernel<double, NN, SK, false>), grid, blk, 0, ctx->stream, b);           \
        } else {                                                                                                            \
            if (want_path) hipLaunchKernelGGL((viterbi_layers_wide_kernel<float, NN, SK, true>), grid, blk, 0, ctx->stream, b);   \
            else hipLaunchKernelGGL((viterbi_layers_wide_kernel<float, NN, SK, false>), grid, blk, 0, ctx->stream, b);            \
        }                                                                                                                   \
    } while (0)
    GH_LW_CASES(GH_LW)
#undef GH_LW
    GH_HIP(hipGetLastError());
    return GH_OK;
}

int gh_launch_lattice_backtrace_wide(gh_ctx* ctx, const gh_layers_args& a, const gh_layerform& f, int64_t u_begin, int64_t n_utts) {
    if (n_utts <= 0 || !(a.path || a.labels)) return GH_OK;
    gh_layers_args b = a;
    b.slot0 = u_begin;
    const dim3 grid((unsigned)((n_utts + 63) / 64)), blk(64);
    const int64_t slot_end = u_begin + n_utts;
    const bool labels = a.labels != nullptr;
#define GH_BW(NN, SK)                                                                                                       \
    do {                                                                                                                    \
        if (f.loop) {                                                                                                       \
            if (labels) hipLaunchKernelGGL((loop_backtrace_wide_kernel<NN, SK, 1>), grid, blk, 0, ctx->stream, b, slot_end); \
            else hipLaunchKernelGGL((loop_backtrace_wide_kernel<NN, SK, 0>), grid, blk, 0, ctx->stream, b, slot_end);        \
        } else {                                                                                                            \
            if (labels) hipLaunchKernelGGL((lattice_backtrace_wide_kernel<NN, SK, 1>), grid, blk, 0, ctx->stream, b, slot_end); \
            else hipLaunchKernelGGL((lattice_backtrace_wide_kernel<NN, SK, 0>), grid, blk, 0, ctx->stream, b, slot_end);        \
        }                                                                                                                   \
    } while (0)
    GH_LW_CASES(GH_BW)
#undef GH_BW
    GH_HIP(hipGetLastError());
    return GH_OK;
}
