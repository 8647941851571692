// Viterbi over a K-layer word lattice in LAYER FORM (gh_layerform, gh_internal.h): the decode lattice of the
// reference's continuous-digit recogniser (build_state_sequences, continuous_speech.py:13-53, as main.py:35 builds
// it: the same W words in each of K layers, one non-emitting row between layers).  Same semantics as the generic
// kernel (decode_hmm_states, decode.py:80-146): candidates in ascending origin order with a strict '<', arcs that
// touch a non-emitting row read the SAME column (so the frame at a word boundary is scored by the last state of one
// word and the first state of the next), start only in cell (0, 0), last of equal end points, path without the end
// cell.
//
// gfx950 mapping -- ONE WAVE PER UTTERANCE, no LDS, no barrier:
//   * lane = (layer mod 4, word): the four 16-lane DPP rows of the wave are four consecutive layers, a lane owns one
//     word of its layer and keeps the word's N state costs in REGISTERS (two register sets: layers 0-3 and 4-7; four for 9-16 layers).
//     The left neighbour of a state is therefore another register of the same lane -- no cross-lane traffic
//     inside a word at all;
//   * the non-emitting row after a layer is the minimum over the words' last states: four row_ror DPP steps leave the
//     row minimum in all 16 lanes; `cand == min` is the back-pointer (lowest set bit = np.argmin's first minimum,
//     found at back-trace time, not in the column loop); row_bcast:15 hands the minimum to the next layer's row,
//     v_readlane carries it from layer 3 to layer 4;
//   * a word's first state takes the non-emitting row's value of the SAME column after the reduction; nothing else in
//     the column depends on it (words have >= 2 states), so the layers of a column are independent instruction
//     streams for the scheduler;
//   * back-pointers are decision BITS (self arc strictly better?), shifted into one 32-bit word per lane with
//     v_addc_co_u32 straight from the compare's SGPR mask: 2 (N + 1) bits per column and lane, 128 B per column for
//     N = 5 against 716 B of uint16 back-pointers per column in the row-per-lane kernel;
//   * each lane streams the N emissions of its word from the resident [N, S] matrix (saddr + per-lane byte offset,
//     PF columns in flight);
//   * end selection runs in the same wave; the back-trace is a second kernel with one LANE per utterance
//     (lattice_backtrace_kernel below).
// ~110 VALU instructions per column for all 8 layer slots (the row-per-lane kernel: three LDS phases with
// barriers, ~5 900 cycles per column).
#include "gh_internal.h"
#include "gh_viterbi.h"

namespace {

__device__ __forceinline__ double vmin(double a, double b) {   // IEEE minNum in ONE instruction: a NaN operand loses
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_f64(double old, double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}

// rotation inside the 16-lane rows: every lane has a source, so no `old` value has to be set up (v_mov_b32_dpp only)
template <int CTRL> __device__ __forceinline__ double row_rot(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// minimum over the 16 lanes of a DPP row, left in every lane of the row (row_ror:1,2,4,8)
__device__ __forceinline__ double row_min16(double v) {
    v = vmin(v, row_rot<0x121>(v));
    v = vmin(v, row_rot<0x122>(v));
    v = vmin(v, row_rot<0x124>(v));
    v = vmin(v, row_rot<0x128>(v));
    return v;
}

// word = 2 * word + bit, the bit taken from a compare's lane mask: one VALU instruction
__device__ __forceinline__ void push_bit(uint32_t& word, unsigned long long mask) {
    unsigned long long carry_out;
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(word), "=s"(carry_out) : "v"(word), "s"(mask));
}

// ... and for the 64-bit decision words of the wide word models (12 / 16 states): the carry of the low half goes on
__device__ __forceinline__ void push_bit(uint64_t& word, unsigned long long mask) {
    uint32_t lo = (uint32_t)word, hi = (uint32_t)(word >> 32);
    unsigned long long c1, c2;
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(lo), "=s"(c1) : "v"(lo), "s"(mask));
    asm("v_addc_co_u32 %0, %1, %2, %2, %3" : "=v"(hi), "=s"(c2) : "v"(hi), "s"(c1));
    word = ((uint64_t)hi << 32) | lo;
}

// decision bits of one column and lane in the layer form: H register sets (2: up to 8 layers, 4: up to 16) of N + 1 (+ N - 2
// with skip arcs); two sets fit a 32-bit word up to 8 states per word (and 12 without skip arcs), a 64-bit word up to 16;
// four sets fit 64 bits up to 8 states per word
template <int N, bool SKIP, int H = 2> struct LayerBits {
    static constexpr int HB = N + 1 + (SKIP ? N - 2 : 0);
    static constexpr int BITS = H * HB;
    static constexpr bool WIDE = BITS > 32;
    static constexpr int CPW = (WIDE ? 64 : 32) / BITS;
    static_assert(BITS <= 64, "the decision bits of a column must fit one word");
};
template <bool WIDE> struct DecisionWord { using type = uint32_t; };
template <> struct DecisionWord<true> { using type = uint64_t; };

template <typename ET, int N, bool SKIP, bool WANT_BP, int H = 2>     // H register sets: layers 0-3, 4-7 (, 8-11, 12-15)
__global__ __launch_bounds__(64) void viterbi_layers_kernel(gh_layers_args a) {
    constexpr int HB = LayerBits<N, SKIP, H>::HB;             // decision bits per column and register set
    constexpr int BITS = LayerBits<N, SKIP, H>::BITS;
    constexpr int CPW = LayerBits<N, SKIP, H>::CPW;           // columns per decision word
    constexpr int PF = N > 8 ? 2 : 4;                         // columns of emissions in flight
    using WT = typename DecisionWord<LayerBits<N, SKIP, H>::WIDE>::type;
    static_assert(CPW >= 1, "decision bits of a column must fit one word");
    const int lane = threadIdx.x, kk = lane >> 4, w = lane & 15;
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
    double cin[H], cout[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const bool act = wact && (4 * h + kk) < K;            // unused layer slots / word lanes: everything stays +inf
        cin[h] = act ? lf->cin[wc] : INF;
        cout[h] = act ? lf->cout[wc] : INF;
    }
    const char* nllb = static_cast<const char*>(a.nll) + f0 * a.S * (int64_t)sizeof(ET);   // wave-uniform
    const int64_t rowb = (int64_t)a.S * (int64_t)sizeof(ET);
    ET ring[PF][N];
#pragma unroll
    for (int k = 0; k < PF; ++k)
#pragma unroll
        for (int s = 0; s < N; ++s)
            ring[k][s] = (k < T) ? *reinterpret_cast<const ET*>(nllb + k * rowb + sto[s]) : ET(0);
    double prev[H][N];
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int s = 0; s < N; ++s) prev[h][s] = INF;
    WT word = 0;
    WT* bp = WANT_BP ? reinterpret_cast<WT*>(a.bp + a.bp_off[slot]) + lane : nullptr;
    int cw = 0;                                               // columns already pushed into `word`

    // Read-ahead (round 4, found in the ISA): a load under `if (t + PF < T)`, a `break` out of the unrolled group, a
    // per-column `if (t < T)` or a refill issued while the slot's old value is still live each put an s_waitcnt vmcnt(0)
    // (behind register copies) into every column or every group -- the ring never ran ahead.  So: the main loop takes
    // WHOLE groups of PF columns with no condition around a column, a slot is refilled, unconditionally and from a clamped
    // column, at the END of the column that consumed it, and the last T mod PF columns run from the ring without refills.
    // (the ring's first fill sits in branches: with it still in flight at the loop header the compiler can only count on
    //  "no load issued behind it" and waits for vmcnt(0) at the top of EVERY group; drained here, the header's wait is
    //  the back edge's -- the 15 loads of the three younger slots stay in flight)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    auto column = [&](int t, const ET (&ev)[N]) {
        double e[N];
#pragma unroll
        for (int s = 0; s < N; ++s) e[s] = (double)ev[s];
        double carry = (t == 0) ? 0.0 : INF;              // the start row: cost 0 in column 0 only (decode.py:99-101)
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const double base0 = c0[0] + prev[h][0];      // state 0 from its own previous column
            // states N-1 .. 1 from the previous column (in place, descending: the neighbours are still old)
#pragma unroll
            for (int s = N - 1; s >= 1; --s) {
                const double v0 = c0[s] + prev[h][s];
                const double v1 = c1[s] + prev[h][s - 1];
                double best;
                if (SKIP && s >= 2) {                     // ascending origin order: s-2, s-1, s; strict '<'
                    const double v2 = c2[s] + prev[h][s - 2];
                    const bool b_a = v1 < v2;
                    const double m = vmin(v1, v2);
                    const bool b_b = v0 < m;
                    best = vmin(v0, m);
                    if (WANT_BP) { push_bit(word, __ballot(b_a)); push_bit(word, __ballot(b_b)); }
                } else {
                    const bool b = v0 < v1;
                    best = vmin(v0, v1);
                    if (WANT_BP) push_bit(word, __ballot(b));
                }
                prev[h][s] = vmin(best + e[s], INF);      // min(inf, nan) keeps inf (decode.py:124)
            }
            // the non-emitting row behind this layer: minimum over the words' last states, same column
            const double cand = prev[h][N - 1] + cout[h];
            const double rm = row_min16(cand);
            if (WANT_BP) push_bit(word, __ballot(cand == rm));
            // ... handed to the next layer: rows 1-3 take the row above, row 0 the start row / layer 3
            const double nin = dpp_f64<0x142, 0xE>(carry, rm);
            carry = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rm), 63),
                                     __builtin_amdgcn_readlane(__double2loint(rm), 63));
            // state 0: the non-emitting row (lower row index: it wins ties) against the self arc
            const double cn = nin + cin[h];
            const bool b0 = base0 < cn;
            if (WANT_BP) push_bit(word, __ballot(b0));
            prev[h][0] = vmin(vmin(base0, cn) + e[0], INF);
        }
        if (WANT_BP) {
            if (++cw == CPW || t == T - 1) {
                if (CPW > 1 && cw < CPW) word <<= BITS * (CPW - cw);   // last, partly filled word: left aligned
                bp[(int64_t)(t / CPW) * 64] = word;
                word = 0;
                cw = 0;
            }
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
    int best_slot = -1, best_row = 0;
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
        for (int s = 0; s < N; ++s) {
            const int layer = 4 * h + kk;
            if (wact && layer < K) {
                const int r = layer * (P + 1) + 1 + w * N + s;
                const int es = a.end_slot[r];
                if (es >= 0) {
                    const double v = prev[h][s];
                    if (a.end_cost) a.end_cost[u * a.n_end + es] = v;
                    if (v < best_v || (v == best_v && es > best_slot)) { best_v = v; best_slot = es; best_row = r; }
                }
            }
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(best_v, o);
        const int os = __shfl_xor(best_slot, o), orow = __shfl_xor(best_row, o);
        if (ov < best_v || (ov == best_v && os > best_slot)) { best_v = ov; best_slot = os; best_row = orow; }
    }
    if (lane == 0 && a.best_end) a.best_end[u] = best_slot;
}


// =====================================================================================================================
// LOOP form (gh_layerform.loop): the word-loop grammar.  One "layer" whose non-emitting row feeds its own first states
// (same column), so a DPP row is a complete decode: FOUR UTTERANCES PER WAVE (row = utterance, lane = word, the N
// states of the word in registers).  Per column and row: the states from the previous column, the loop row = row
// minimum over the last states (row_ror), the first states from (start row, loop row, self) in that candidate
// order.  Rows whose utterance has ended are switched off by EXEC (DPP row operations never cross rows).  Decision
// bits: N + 2 (+ N - 2 with skip arcs) per column and lane, 16 lanes x 4 B per CPW columns and utterance.
// The back-trace is a second kernel (lattice_backtrace_kernel, one lane per utterance).
template <int N, bool SKIP> struct LoopBits {
    static constexpr int HB = N + 2 + (SKIP ? N - 2 : 0);
    static constexpr int CPW = 32 / HB;
};

template <typename ET, int N, bool SKIP, bool WANT_BP>
__global__ __launch_bounds__(64) void viterbi_loop_kernel(gh_layers_args a, int64_t slot_end) {
    constexpr int HB = LoopBits<N, SKIP>::HB, CPW = LoopBits<N, SKIP>::CPW;
    constexpr int PF = N > 8 ? 2 : 4;
    static_assert(CPW >= 1, "decision bits of a column must fit one word");
    const int lane = threadIdx.x, kk = lane >> 4, w = lane & 15;
    const gh_layerform* __restrict__ lf = a.lf;
    const int W = lf->W, Lr = lf->loop_row;
    const int64_t slot = a.slot0 + (int64_t)blockIdx.x * 4 + kk;
    const bool has_utt = slot < slot_end;
    const int64_t u = has_utt ? (a.perm ? a.perm[slot] : slot) : 0;
    const int64_t f0 = has_utt ? a.utt_off[u] : 0;
    const int T = has_utt ? (int)(a.utt_off[u + 1] - f0) : 0;
    const double INF = INFINITY;
    int Tmax = T;
    Tmax = max(Tmax, __shfl_xor(Tmax, 16));
    Tmax = max(Tmax, __shfl_xor(Tmax, 32));
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
    const char* nllb = static_cast<const char*>(a.nll) + (T > 0 ? f0 : 0) * a.S * (int64_t)sizeof(ET);   // per row (no frames: frame 0)
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
    uint32_t word = 0;
    uint32_t* bp = (WANT_BP && has_utt) ? reinterpret_cast<uint32_t*>(a.bp + a.bp_off[slot]) + w : nullptr;

    for (int t0 = 0; t0 < Tmax; t0 += PF) {                    // (the columns behind Tmax in the last group: no row is in them)
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int t = t0 + k;
            double e[N];
#pragma unroll
            for (int s = 0; s < N; ++s) e[s] = (double)ring[k][s];
            if (t < T) {                                       // row-uniform: the rows of finished utterances sit out
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
                        if (WANT_BP) { push_bit(word, __ballot(b_a)); push_bit(word, __ballot(b_b)); }
                    } else {
                        const bool b = v0 < v1;
                        best = vmin(v0, v1);
                        if (WANT_BP) push_bit(word, __ballot(b));
                    }
                    prev[s] = vmin(best + e[s], INF);
                }
                // the loop row: minimum over the words' last states of THIS column
                const double cand = prev[N - 1] + cout;
                const double rm = row_min16(cand);
                if (WANT_BP) push_bit(word, __ballot(cand == rm));
                // state 0: start row (row 0), loop row, self -- ascending origin, strict '<'
                const double cs = ((t == 0) ? 0.0 : INF) + cin0;
                const double cl = rm + cin;
                const bool b_l = cl < cs;
                const double m2 = vmin(cl, cs);
                const bool b_s = base0 < m2;
                if (WANT_BP) { push_bit(word, __ballot(b_l)); push_bit(word, __ballot(b_s)); }
                prev[0] = vmin(vmin(base0, m2) + e[0], INF);
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
                // use of the old value (see viterbi_layers_kernel); rows without frames read frame 0 of the matrix (some
                // row has frames, or this loop would not run)
                const int tn = (t + PF < T) ? t + PF : (T > 0 ? T - 1 : 0);
                const char* colp = nllb + (int64_t)tn * rowb;
#pragma unroll
                for (int s = 0; s < N; ++s) ring[k][s] = *reinterpret_cast<const ET*>(colp + sto[s]);
            }
        }
    }
    if (!has_utt) return;
    // ---- end costs + end selection inside the row ('>=': the last of equal minima) ----
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
    for (int o = 8; o >= 1; o >>= 1) {
        const double ov = __shfl_xor(best_v, o);
        const int os = __shfl_xor(best_slot, o);
        if (ov < best_v || (ov == best_v && os > best_slot)) { best_v = ov; best_slot = os; }
    }
    if (w == 0 && a.best_end) a.best_end[u] = T > 0 ? best_slot : -1;
}

// ---------------------------------------------------------------------------------------------------------------------
// Back-trace of both forms (decode.py:143-145): ONE LANE PER UTTERANCE.  A first version walked one utterance per
// wave with wave-uniform (scalar) code and v_readlane on register-resident decision words; correct, but a CU has ONE
// scalar unit for its four SIMDs, and ~100 scalar instructions per cell x 330 cells x 125 000 utterances is 6 ms of
// pure SALU issue (measured: 5.5 ms, more than the forward sweep).  Per-lane walks put the same work on the vector
// units, 64 utterances per wave.  A lane keeps the decision word of its current (word index, word) in a register and
// the one below it prefetched, so a memory round trip is only exposed when the path changes words.
// MODE 0: the (row, column) path as decode_hmm_states returns it; MODE 1: only the label sequence of main.py:59-67
// (gh_viterbi_labels) -- cells are visited end -> start, a run of labelled rows reports the label of its first row
// in start -> end order, i.e. of the last one visited before an unlabelled row; labels are collected from the back of
// the utterance's slot and moved to its front at the end.
template <int N, bool SKIP, bool LOOP, int MODE, int H = 2>
__global__ __launch_bounds__(64) void lattice_backtrace_kernel(gh_layers_args a, int64_t slot_end) {
    constexpr int HB = LOOP ? LoopBits<N, SKIP>::HB : LayerBits<N, SKIP, H>::HB;
    constexpr int BITS = LOOP ? HB : H * HB;
    constexpr bool WIDE = !LOOP && LayerBits<N, SKIP, H>::WIDE;  // 64-bit decision words (viterbi_layers_kernel)
    using WT = typename DecisionWord<WIDE>::type;
    constexpr int CPW = (WIDE ? 64 : 32) / BITS;
    constexpr int LPW = LOOP ? 16 : 64;                       // decision words per word index
    __shared__ uint8_t s_arcs[GH_LAYERS_MAXW * GH_LAYERS_MAXN];
    const gh_layerform* __restrict__ lf = a.lf;
    for (int i = threadIdx.x; i < GH_LAYERS_MAXW * GH_LAYERS_MAXN; i += 64) s_arcs[i] = (&lf->arcs[0][0])[i];
    __syncthreads();
    const int W = lf->W, P = lf->P, Lr = lf->loop_row;
    const int64_t slot = a.slot0 + (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (slot >= slot_end) return;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int T = (int)(a.utt_off[u + 1] - a.utt_off[u]);
    const int be = a.best_end[u];
    int32_t* out_n = MODE == 0 ? a.path_len : a.n_labels;
    if (T <= 1 || be < 0) { out_n[u] = 0; return; }
    auto row_of = [&](int k, int ww, int ss) {
        return LOOP ? (ss == 0 ? Lr + 1 + ww : 1 + ww * (N - 1) + (ss - 1)) : k * (P + 1) + 1 + ww * N + ss;
    };
    int bk = 0, bw, bs;
    {
        const int r = a.end_rows[be];
        if (LOOP) {
            if (r > Lr) { bw = r - Lr - 1; bs = 0; } else { bw = (r - 1) / (N - 1); bs = (r - 1) % (N - 1) + 1; }
        } else {
            bk = (r - 1) / (P + 1);
            const int pos = (r - 1) % (P + 1);
            bw = pos / N;
            bs = pos % N;
        }
    }
    const WT* bpu = reinterpret_cast<const WT*>(a.bp + a.bp_off[slot]);
    int32_t* path = MODE == 0 ? a.path + 2 * a.path_off[u] : nullptr;
    int32_t* labs = MODE == 1 ? a.labels + a.label_off[u] : nullptr;
    const int64_t cap = MODE == 0 ? a.path_off[u + 1] - a.path_off[u] : a.label_off[u + 1] - a.label_off[u];
    int64_t len = 0;
    int prev_label = -1;                                      // MODE 1: label of the cell visited last
    int j = T - 1, kind = 0, kn = 0;                          // kind 0 emitting (bk, bw, bs); 1 non-emitting row kn; 2 start row
    int flag = 0;
    int64_t key = -1;                                         // index of the decision word held in `cw`
    WT cw = 0, pw = 0;                                        // current word, and the one LPW words below it
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
        const int wi = j / CPW;
        const int shift = (CPW - 1 - j % CPW) * BITS;
        if (kind == 0) {
            const int64_t want = (int64_t)wi * LPW + (LOOP ? bw : (bk & 3) * 16 + bw);
            if (want != key) {
                if (want == key - LPW) cw = pw; else cw = bpu[want];
                key = want;
                pw = (wi > 0) ? bpu[want - LPW] : WT(0);
            }
            const uint32_t hb = (uint32_t)((cw >> (shift + (LOOP ? 0 : (H - 1 - (bk >> 2)) * HB))) & (WT)((1ull << HB) - 1ull));
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
            } else if (LOOP) {
                // candidates in ascending origin order: start row (arcs bit4), loop row (bit3), self (bit0)
                const int b_l = (hb >> 1) & 1, b_s = hb & 1;
                int pick = b_s ? 0 : (b_l ? 3 : 4);
                if (!((arcs >> pick) & 1)) pick = (arcs & 16) ? 4 : (arcs & 8) ? 3 : (arcs & 1) ? 0 : -1;
                if (pick < 0) { flag |= 2; break; }
                if (pick == 0) { --j; visit(row_of(0, bw, 0), j); }
                else if (pick == 3) { kind = 1; visit(Lr, j); }
                else { kind = 2; visit(0, j); }
            } else {
                const bool self_better = hb & 1;
                const bool take_self = (self_better && (arcs & 1)) || !(arcs & 8);
                if (take_self && !(arcs & 1)) { flag |= 2; break; }
                if (take_self) { --j; visit(row_of(bk, bw, 0), j); }
                else { kind = 1; kn = bk; visit(kn * (P + 1), j); }          // the non-emitting row in front of the layer
            }
        } else if (kind == 1) {
            if (!LOOP && kn == 0) { flag |= 2; break; }                       // the start row has no origin
            const int kp = LOOP ? 0 : kn - 1;
            const int eq_shift = shift + (LOOP ? 2 : (H - 1 - (kp >> 2)) * HB + 1);
            int found = -1;
            if constexpr (WIDE) {
                const ulonglong2* rowp = reinterpret_cast<const ulonglong2*>(bpu + (int64_t)wi * LPW + (kp & 3) * 16);
#pragma unroll
                for (int q2 = 7; q2 >= 0; --q2) {                               // 16 word lanes = 128 contiguous bytes
                    const ulonglong2 v = rowp[q2];
                    if ((v.y >> eq_shift) & 1ull) found = 2 * q2 + 1;
                    if ((v.x >> eq_shift) & 1ull) found = 2 * q2;
                }
            } else {
                const uint4* rowp = reinterpret_cast<const uint4*>(bpu + (int64_t)wi * LPW + (LOOP ? 0 : (kp & 3) * 16));
#pragma unroll
                for (int q4 = 3; q4 >= 0; --q4) {                               // 16 word lanes = 64 contiguous bytes
                    const uint4 v = rowp[q4];
                    if ((v.w >> eq_shift) & 1u) found = 4 * q4 + 3;
                    if ((v.z >> eq_shift) & 1u) found = 4 * q4 + 2;
                    if ((v.y >> eq_shift) & 1u) found = 4 * q4 + 1;
                    if ((v.x >> eq_shift) & 1u) found = 4 * q4;
                }
            }
            if (found < 0 || found >= W) { flag |= 2; break; }                 // lowest word = lowest origin row (np.argmin)
            bw = found;
            bk = kp;
            bs = N - 1;
            kind = 0;
            visit(row_of(bk, bw, bs), j);
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
        for (int64_t i = 0; i < len; ++i) labs[i] = labs[cap - len + i];       // to the front, start -> end order
    }
    out_n[u] = (int32_t)len;
}

}  // namespace

// back-pointer scratch of one utterance of T frames, in uint16 units (the lattice kernels' common unit)
size_t gh_layers_bp_entries(const gh_layerform& f, int64_t T) {
    if (f.W > GH_LAYERS_ROWW) return f.loop ? gh_loop_wide_bp_entries(T) : gh_layers_wide_bp_entries(T);
    if (f.loop) {
        const int hb = f.N + 2 + (f.skip ? f.N - 2 : 0);
        const int cpw = 32 / hb;
        return (size_t)((T + cpw - 1) / cpw) * 16 * 2;
    }
    const int hb = f.N + 1 + (f.skip ? f.N - 2 : 0), sets = f.K > 8 ? 4 : 2;
    const int wbits = sets * hb > 32 ? 64 : 32;               // (LayerBits)
    const int cpw = wbits / (sets * hb);
    return (size_t)((T + cpw - 1) / cpw) * 64 * (wbits / 16);
}

#define GH_LY_CASES(ET, MACRO)                   \
    switch (f.N) {                               \
        case 2: MACRO(ET, 2, false); break;      \
        case 3: if (f.skip) MACRO(ET, 3, true); else MACRO(ET, 3, false); break; \
        case 4: if (f.skip) MACRO(ET, 4, true); else MACRO(ET, 4, false); break; \
        case 5: if (f.skip) MACRO(ET, 5, true); else MACRO(ET, 5, false); break; \
        case 6: if (f.skip) MACRO(ET, 6, true); else MACRO(ET, 6, false); break; \
        case 7: if (f.skip) MACRO(ET, 7, true); else MACRO(ET, 7, false); break; \
        case 8: if (f.skip) MACRO(ET, 8, true); else MACRO(ET, 8, false); break; \
        case 12: if (f.skip) MACRO(ET, 12, true); else MACRO(ET, 12, false); break; \
        case 16: if (f.skip) MACRO(ET, 16, true); else MACRO(ET, 16, false); break; \
        default: gh_set_error("gh_viterbi: layer form with %d states per word", f.N); return GH_ERR_UNSUPPORTED; \
    }

int gh_launch_viterbi_layers(gh_ctx* ctx, const gh_layers_args& a, const gh_layerform& f, int64_t u_begin, int64_t n_utts,
                             bool f64, bool want_path) {
    if (n_utts <= 0) return GH_OK;
    if (f.W > GH_LAYERS_ROWW) return gh_launch_viterbi_layers_wide(ctx, a, f, u_begin, n_utts, f64, want_path);
    gh_layers_args b = a;
    b.slot0 = u_begin;
    const dim3 blk(64);
    if (f.loop) {
        const dim3 grid((unsigned)((n_utts + 3) / 4));
        const int64_t slot_end = u_begin + n_utts;
#define GH_LP(ET, NN, SK)                                                                                                  \
    do {                                                                                                                   \
        if (want_path) hipLaunchKernelGGL((viterbi_loop_kernel<ET, NN, SK, true>), grid, blk, 0, ctx->stream, b, slot_end);  \
        else hipLaunchKernelGGL((viterbi_loop_kernel<ET, NN, SK, false>), grid, blk, 0, ctx->stream, b, slot_end);           \
    } while (0)
        if (f64) { GH_LY_CASES(double, GH_LP) } else { GH_LY_CASES(float, GH_LP) }
#undef GH_LP
        GH_HIP(hipGetLastError());
        return GH_OK;
    }
    const dim3 grid((unsigned)n_utts);
    // (more than 8 layers: four register sets, built for up to 8 states per word -- gh_layerform_ok)
#define GH_LY(ET, NN, SK)                                                                                      \
    do {                                                                                                       \
        if constexpr (NN <= 8) {                                                                               \
            if (f.K > 8) {                                                                                     \
                if (want_path) hipLaunchKernelGGL((viterbi_layers_kernel<ET, NN, SK, true, 4>), grid, blk, 0, ctx->stream, b); \
                else hipLaunchKernelGGL((viterbi_layers_kernel<ET, NN, SK, false, 4>), grid, blk, 0, ctx->stream, b);          \
                break;                                                                                         \
            }                                                                                                  \
        }                                                                                                      \
        if (want_path) hipLaunchKernelGGL((viterbi_layers_kernel<ET, NN, SK, true>), grid, blk, 0, ctx->stream, b); \
        else hipLaunchKernelGGL((viterbi_layers_kernel<ET, NN, SK, false>), grid, blk, 0, ctx->stream, b);          \
    } while (0)
    if (f64) { GH_LY_CASES(double, GH_LY) } else { GH_LY_CASES(float, GH_LY) }
#undef GH_LY
    GH_HIP(hipGetLastError());
    return GH_OK;
}

// the path (a.path) or the label sequences (a.labels) of the utterances [u_begin, u_begin + n_utts) from the decision words
int gh_launch_lattice_backtrace(gh_ctx* ctx, const gh_layers_args& a, const gh_layerform& f, int64_t u_begin, int64_t n_utts) {
    if (n_utts <= 0 || !(a.path || a.labels)) return GH_OK;
    if (f.W > GH_LAYERS_ROWW) return gh_launch_lattice_backtrace_wide(ctx, a, f, u_begin, n_utts);
    gh_layers_args b = a;
    b.slot0 = u_begin;
    const dim3 grid((unsigned)((n_utts + 63) / 64)), blk(64);
    const int64_t slot_end = u_begin + n_utts;
    const bool labels = a.labels != nullptr;
#define GH_BT(ET, NN, SK)                                                                                               \
    do {                                                                                                                \
        if (f.loop) {                                                                                                   \
            if (labels) hipLaunchKernelGGL((lattice_backtrace_kernel<NN, SK, true, 1>), grid, blk, 0, ctx->stream, b, slot_end);  \
            else hipLaunchKernelGGL((lattice_backtrace_kernel<NN, SK, true, 0>), grid, blk, 0, ctx->stream, b, slot_end);         \
        } else {                                                                                                        \
            if constexpr (NN <= 8) {                                                                                    \
                if (f.K > 8) {                                                                                          \
                    if (labels) hipLaunchKernelGGL((lattice_backtrace_kernel<NN, SK, false, 1, 4>), grid, blk, 0, ctx->stream, b, slot_end); \
                    else hipLaunchKernelGGL((lattice_backtrace_kernel<NN, SK, false, 0, 4>), grid, blk, 0, ctx->stream, b, slot_end);        \
                    break;                                                                                              \
                }                                                                                                       \
            }                                                                                                           \
            if (labels) hipLaunchKernelGGL((lattice_backtrace_kernel<NN, SK, false, 1>), grid, blk, 0, ctx->stream, b, slot_end); \
            else hipLaunchKernelGGL((lattice_backtrace_kernel<NN, SK, false, 0>), grid, blk, 0, ctx->stream, b, slot_end);        \
        }                                                                                                               \
    } while (0)
    GH_LY_CASES(double, GH_BT)
#undef GH_BT
    GH_HIP(hipGetLastError());
    return GH_OK;
}
