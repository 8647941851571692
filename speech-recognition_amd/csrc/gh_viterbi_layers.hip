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
//     word of its layer and keeps the word's N state costs in REGISTERS (two register sets: layers 0-3 and 4-7).
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
//   * end selection and back-trace run in the same wave: the walk is wave-uniform (scalar registers), the decision
//     words of 16 words x CPW columns sit in registers (v_readlane), the next 16 are in flight.
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

template <typename ET, int N, bool SKIP, bool WANT_BP>
__global__ __launch_bounds__(64) void viterbi_layers_kernel(gh_layers_args a) {
    constexpr int H = 2;                                      // register sets: layers 0-3 and 4-7
    constexpr int HB = N + 1 + (SKIP ? N - 2 : 0);            // decision bits per column and register set
    constexpr int BITS = H * HB;
    constexpr int CPW = 32 / BITS;                            // columns per 32-bit decision word
    constexpr int PF = 4;                                     // columns of emissions in flight
    static_assert(BITS <= 32 && CPW >= 1, "decision bits of a column must fit one word");
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
    uint32_t word = 0;
    uint32_t* bp = WANT_BP ? reinterpret_cast<uint32_t*>(a.bp + a.bp_off[slot]) + lane : nullptr;
    int cw = 0;                                               // columns already pushed into `word`

    for (int t0 = 0; t0 < T; t0 += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int t = t0 + k;
            if (t >= T) break;
            double e[N];
#pragma unroll
            for (int s = 0; s < N; ++s) e[s] = (double)ring[k][s];
            if (t + PF < T) {
                const char* colp = nllb + (int64_t)(t + PF) * rowb;
#pragma unroll
                for (int s = 0; s < N; ++s) ring[k][s] = *reinterpret_cast<const ET*>(colp + sto[s]);
            }
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
        }
    }

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
    if (!WANT_BP) return;
    if (!a.path) return;
    if (T <= 1 || best_slot < 0) {
        if (lane == 0) a.path_len[u] = 0;
        return;
    }

    // ---- back-trace (decode.py:143-145): wave-uniform walk, decision words in registers ----
    int32_t* path = a.path + 2 * a.path_off[u];
    const int64_t cap = a.path_off[u + 1] - a.path_off[u];
    int j = T - 1;
    bool on_nes = false;
    int kn = 0;                                               // non-emitting row index (0 .. K) while on_nes
    int bk, bw, bs;                                           // emitting cell: layer, word, state
    {
        const int r = __builtin_amdgcn_readfirstlane(best_row);
        bk = (r - 1) / (P + 1);
        const int pos = (r - 1) % (P + 1);
        bw = pos / N;
        bs = pos % N;
    }
    int pr = 0, pc = 0, nbuf = 0;                             // path cells parked in lanes 0 .. nbuf-1
    int64_t len = 0;
    bool stop = false;
    auto emit = [&](int row, int col) {
        pr = (lane == nbuf) ? row : pr;
        pc = (lane == nbuf) ? col : pc;
        ++nbuf;
        if (len + nbuf > cap) { if (lane == 0) atomicOr(a.flag, 4); stop = true; --nbuf; return; }
        if (nbuf == 64) {
            reinterpret_cast<int2*>(path)[len + lane] = make_int2(pr, pc);
            len += 64;
            nbuf = 0;
        }
    };
    constexpr int WCH = 8;                                    // decision words per register chunk
    const uint32_t* bpr = reinterpret_cast<const uint32_t*>(a.bp + a.bp_off[slot]) + lane;
    const int wi_hi = (T - 1) / CPW;
    uint32_t wr[WCH], nx[WCH];
    int cb = wi_hi & ~(WCH - 1);
#pragma unroll
    for (int i = 0; i < WCH; ++i) wr[i] = (cb + i <= wi_hi) ? bpr[(int64_t)(cb + i) * 64] : 0u;
    for (; cb >= 0 && !stop && j != 0; cb -= WCH) {
#pragma unroll
        for (int i = 0; i < WCH; ++i) nx[i] = (cb - WCH + i >= 0) ? bpr[(int64_t)(cb - WCH + i) * 64] : 0u;
#pragma unroll
        for (int i = WCH - 1; i >= 0; --i) {
            const int wi = cb + i;
            while (!stop && j != 0 && j / CPW == wi) {
                const int shift = (CPW - 1 - j % CPW) * BITS;
                if (!on_nes) {
                    const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)wr[i], (bk & 3) * 16 + bw);
                    const uint32_t hb = (x >> (shift + (H - 1 - (bk >> 2)) * HB)) & ((1u << HB) - 1u);
                    const int arcs = lf->arcs[bw][bs];
                    if (bs >= 1) {
                        // pushes before state bs: states N-1 .. bs+1 (two bits each when they have a skip arc slot)
                        int before = 0;
                        for (int s2 = N - 1; s2 > bs; --s2) before += (SKIP && s2 >= 2) ? 2 : 1;
                        int code;
                        if (SKIP && bs >= 2) {
                            const int b_a = (hb >> (HB - 1 - before)) & 1, b_b = (hb >> (HB - 2 - before)) & 1;
                            code = b_b ? 0 : (b_a ? 1 : 2);
                        } else {
                            code = ((hb >> (HB - 1 - before)) & 1) ? 0 : 1;
                        }
                        if (!((arcs >> code) & 1)) {          // every candidate was +inf: the first existing arc (lowest origin)
                            code = (arcs & 4) ? 2 : (arcs & 2) ? 1 : (arcs & 1) ? 0 : -1;
                        }
                        if (code < 0) { if (lane == 0) atomicOr(a.flag, 2); stop = true; break; }
                        bs -= code;
                        --j;
                        emit(bk * (P + 1) + 1 + bw * N + bs, j);
                    } else {
                        const int self_better = hb & 1;
                        const bool take_self = (self_better && (arcs & 1)) || !(arcs & 8);
                        if (take_self && !(arcs & 1)) { if (lane == 0) atomicOr(a.flag, 2); stop = true; break; }
                        if (take_self) {
                            --j;
                            emit(bk * (P + 1) + 1 + bw * N, j);
                        } else {                              // the non-emitting row in front of the layer, same column
                            on_nes = true;
                            kn = bk;
                            emit(kn * (P + 1), j);
                        }
                    }
                } else {
                    if (kn == 0) { if (lane == 0) atomicOr(a.flag, 2); stop = true; break; }   // the start row has no origin
                    const int kp = kn - 1;
                    const uint32_t eq = (wr[i] >> (shift + (H - 1 - (kp >> 2)) * HB + 1)) & 1u;
                    const unsigned long long m = __ballot(eq != 0);
                    const unsigned sl = (unsigned)(m >> (16 * (kp & 3))) & ((1u << W) - 1u);
                    if (sl == 0) { if (lane == 0) atomicOr(a.flag, 2); stop = true; break; }
                    bw = __builtin_ctz(sl);                   // lowest word index = lowest origin row: np.argmin's first minimum
                    bk = kp;
                    bs = N - 1;
                    on_nes = false;
                    emit(bk * (P + 1) + 1 + bw * N + bs, j);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < WCH; ++i) wr[i] = nx[i];
    }
    if (nbuf > 0 && lane < nbuf) reinterpret_cast<int2*>(path)[len + lane] = make_int2(pr, pc);
    if (lane == 0) a.path_len[u] = (int32_t)(len + nbuf);
}

}  // namespace

// back-pointer scratch of one utterance of T frames, in uint16 units (the lattice kernels' common unit)
size_t gh_layers_bp_entries(const gh_layerform& f, int64_t T) {
    const int hb = f.N + 1 + (f.skip ? f.N - 2 : 0);
    const int cpw = 32 / (2 * hb);
    return (size_t)((T + cpw - 1) / cpw) * 64 * 2;
}

int gh_launch_viterbi_layers(gh_ctx* ctx, const gh_layers_args& a, const gh_layerform& f, int64_t u_begin, int64_t n_utts,
                             bool f64, bool want_path) {
    if (n_utts <= 0) return GH_OK;
    gh_layers_args b = a;
    b.slot0 = u_begin;
    const dim3 grid((unsigned)n_utts), blk(64);
#define GH_LY(ET, NN, SK, BP) hipLaunchKernelGGL((viterbi_layers_kernel<ET, NN, SK, BP>), grid, blk, 0, ctx->stream, b)
#define GH_LY_B(ET, NN, SK) do { if (want_path) GH_LY(ET, NN, SK, true); else GH_LY(ET, NN, SK, false); } while (0)
#define GH_LY_S(ET, NN) do { if (f.skip) GH_LY_B(ET, NN, true); else GH_LY_B(ET, NN, false); } while (0)
#define GH_LY_N(ET)                              \
    switch (f.N) {                               \
        case 2: GH_LY_B(ET, 2, false); break;    \
        case 3: GH_LY_S(ET, 3); break;           \
        case 4: GH_LY_S(ET, 4); break;           \
        case 5: GH_LY_S(ET, 5); break;           \
        case 6: GH_LY_S(ET, 6); break;           \
        case 7: GH_LY_S(ET, 7); break;           \
        case 8: GH_LY_S(ET, 8); break;           \
        default: gh_set_error("gh_viterbi: layer form with %d states per word", f.N); return GH_ERR_UNSUPPORTED; \
    }
    if (f64) { GH_LY_N(double) } else { GH_LY_N(float) }
#undef GH_LY_N
#undef GH_LY_S
#undef GH_LY_B
#undef GH_LY
    GH_HIP(hipGetLastError());
    return GH_OK;
}
