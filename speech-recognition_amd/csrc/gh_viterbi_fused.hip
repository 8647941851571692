// Fused single-Gaussian decode: distance AND dynamic program in one sweep, nothing materialised.
//
// Reference: HMM.evaluate for a model with one Gaussian per state, sr/recognition/hmm.py:131-135 --
//   dtw(x, self.mu, mahalanobis, self.transitions, self.sigma)          (use_gmm=False; hmm_state.py:48-58, decode.py:7-77)
//   decode_hmm_states(x, [GMM(mu, sigma, 1)], self.transitions)          (a one-component mixture; hmm_state.py:114-120)
// both score cell (state, frame) with  1/2 log((2 pi)^D prod var) + 1/2 sum (x - mu)^2 / var  (- log w for the mixture)
// while they fill the cost matrix.  The two-kernel form (gh_loglik -> [N, S] matrix -> gh_viterbi) moves
// 8 S bytes per frame out to HBM and back; SURVEY.md 8(d) prices the fused sweep at 8 D + 4 bytes per frame.
//
// gfx950 mapping.  ONE WAVE per (utterance, group of <= 64 rows holding whole chains), as in gh_viterbi_chain.hip:
// the cost column is one VGPR pair per lane, the neighbour's previous cost arrives by a DPP wave shift.  New here:
//   * each lane keeps ITS state's Gaussian in registers as s_d = sqrt(1/(2 var_d)) and ms_d = -mu_d s_d, so a cell's
//     distance is 2 D fma:  e_d = fma(x_d, s_d, ms_d);  q = fma(e_d, e_d, q)   (no division, no subtraction; the
//     accumulator starts at the normaliser, so q IS the cell's cost);
//   * the frame x_t is the same for all 64 lanes: a tile of FT frames is staged ONCE in LDS (coalesced loads of the
//     row-major [frames, D] matrix: the only HBM traffic of the sweep) and read back as 16-byte LDS broadcasts, requested
//     for column t+1 as soon as column t's fma block has read its registers;
//   * the next tile travels from HBM into registers while the current one is computed; a workgroup is one wave, so
//     there is no barrier anywhere;
//   * a wave walks several utterances (serpentine over the longest-first order) and loads its 2 D + 2 constants once.
// The sweep is bound by the fp64 vector pipe -- 2 D fma + 11 other instructions per cell column at 4 cycles each
// (rocprofv3: SQ_ACTIVE_INST_VALU = 94 % of a SIMD's cycles) -- not by HBM; bench.py reports both fractions.
#include "gh_internal.h"
#include "gh_viterbi.h"
#include <type_traits>

namespace {

constexpr int FT = 32;  // frames per LDS tile

__device__ __forceinline__ double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// IEEE minNum in ONE instruction: a NaN operand loses (fmin() adds canonicalising v_max x,x around it)
__device__ __forceinline__ double vmin(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// lane i <- lane i-1, lane 0 <- +0.0 (bound_ctrl): the first row of a lane group starts a chain, its r-1 / r-2 arcs cost
// +inf, and +inf + 0 is still +inf -- no fill registers
__device__ __forceinline__ double wave_shr1z(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// sqrt(1/(2 var)), -mean sqrt(1/(2 var)), -logc and the underflow threshold of every lattice row, k-major ([2 DVp + 2][Rp])
// so that a wave's loads are contiguous.  Rows >= R repeat row 0 (never active).
__global__ void fused_params_kernel(const double* __restrict__ mean, const double* __restrict__ ivar,
                                    const double* __restrict__ logc, const int32_t* __restrict__ row_state, int R, int Rp,
                                    int D, int DVp, double thr, double* __restrict__ par) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= Rp) return;
    const int st = row_state[r < R ? r : 0];
    for (int d = 0; d < DVp; ++d) {
        double s = 0.0, ms = 0.0;
        if (d < D) {
            s = sqrt(0.5 * ivar[(size_t)st * D + d]);
            ms = -(mean[(size_t)st * D + d] * s);
        }
        par[(size_t)d * Rp + r] = s;
        par[(size_t)(DVp + d) * Rp + r] = ms;
    }
    const double konst = -logc[st];
    par[(size_t)(2 * DVp) * Rp + r] = konst;
    // GMM.evaluate works in the linear domain (hmm_state.py:114-120): np.exp(-q) is 0 below ln 2^-1075 whatever the
    // normaliser is, and so is w * norm * exp(-q) once the product is: +inf when max(q, konst + q) > thr.  The kernel
    // compares the cell's cost konst + q:  konst + q > thr + min(konst, 0).
    par[(size_t)(2 * DVp + 1) * Rp + r] = thr + (konst < 0.0 ? konst : 0.0);
}

// DV: LDS row stride in elements (rows are 16-byte multiples: every broadcast read is a ds_read_b128); DX <= DV: the
// dimensions that are computed (DX == D when the instantiation is made for that D; otherwise D <= DX == DV and the pad
// dimensions, zero in LDS and in the constants, add zeros).  A table in LDS maps a tile element (row stride D in memory)
// to its LDS slot (row stride DV).
// LIN: the linear-domain underflow rule of GMM.evaluate (a compare + a select per cell); off for mahalanobis() models.
template <typename ET, int DV, int DX, bool WANT_BP, bool WANT_COSTS, bool SKIP, bool LIN>
__global__ __launch_bounds__(64) void viterbi_fused_kernel(gh_fused_args fa) {
    constexpr int NLD = (FT * DV + 63) / 64;   // tile elements a lane moves (upper bound: D <= DV)
    constexpr int VW = 16 / sizeof(ET);        // elements per 16-byte LDS read
    static_assert(DV % VW == 0 && DX <= DV, "LDS rows are 16-byte multiples");
    typedef ET vec_t __attribute__((ext_vector_type(VW)));
    __shared__ __attribute__((aligned(16))) ET tile[(FT + 1) * DV];   // (+1 row: the read-ahead of the last column stays inside)
    __shared__ uint16_t slot_of[NLD * 64];
    const gh_chain_args& a = fa.c;
    const int lane = threadIdx.x;
    const int D = fa.D, Rp = fa.Rp, R = a.R;
    const double INF = INFINITY;

    for (int e = lane; e < NLD * 64; e += 64) {
        const int fr = e / D;
        slot_of[e] = (uint16_t)(fr * DV + (e - fr * D));
    }
    for (int i = lane; i < (FT + 1) * DV; i += 64) tile[i] = ET(0);   // the pad dimensions stay zero for good

    ET s[DX], ms[DX];
    double konst = 0, cthr = 0, c0 = INF, c1 = INF, c2 = INF;
    uint8_t first_code = 3;
    bool is_start = false, act = false;
    int cur_g = -1, rr = 0, es = -1;

    const int64_t G = gridDim.x;
    // several row groups per utterance (items = utterance x group) and a grid that is a multiple of the group count: the
    // serpentine runs over the utterances of ONE group, so a wave keeps its group -- and the 2 D + 2 constants it loaded --
    // in odd rounds too (G - 1 - blockIdx.x has another residue than blockIdx.x)
    const int ng = a.n_groups;
    const bool own_group = ng > 1 && G % ng == 0;
    const int64_t Gq = own_group ? G / ng : G, q = own_group ? blockIdx.x / ng : blockIdx.x, gl = own_group ? blockIdx.x % ng : 0;
    for (int64_t round = 0;; ++round) {
        // serpentine over the longest-first launch order: the waves' frame totals stay within one utterance of each other
        const int64_t sq = round * Gq + ((round & 1) ? Gq - 1 - q : q);
        const int64_t item = own_group ? sq * ng + gl : sq;
        if (round * G >= fa.n_items) break;
        if (item >= fa.n_items) continue;
        const int64_t slot = a.slot0 + item / a.n_groups;
        const int g = (int)(item % a.n_groups);
        if (g != cur_g) {
            cur_g = g;
            const int r0 = a.group_row0[g];
            const int nrows = a.group_row0[g + 1] - r0;
            act = lane < nrows;
            rr = act ? r0 + lane : r0;
            const double* p = fa.par + rr;
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                s[d] = (ET)p[(size_t)d * Rp];
                ms[d] = (ET)p[(size_t)(fa.DVp + d) * Rp];
            }
            konst = p[(size_t)(2 * fa.DVp) * Rp];
            cthr = p[(size_t)(2 * fa.DVp + 1) * Rp];
            c0 = act ? a.cost0[rr] : INF;
            c1 = act ? a.cost1[rr] : INF;
            c2 = (SKIP && act) ? a.cost2[rr] : INF;
            const uint8_t info = act ? a.row_info[rr] : 0x0F;
            first_code = info & 3;
            is_start = (info & 4) != 0;
            es = act ? a.end_slot[rr] : -1;
            // vmcnt(0) HERE: otherwise the first use of these constants inside the column loop waits for every load in
            // flight -- the next tile's included (the counter is in order) -- in every column
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        const int64_t u = a.perm ? a.perm[slot] : slot;
        const int64_t f0 = a.utt_off[u];
        const int T = (int)(a.utt_off[u + 1] - f0);
        if (T <= 0) {
            if (fa.select_end && lane == 0) a.best_end[u] = -1;
            continue;
        }
        const ET* src = static_cast<const ET*>(fa.feats) + f0 * D;

        ET pre[NLD];
        auto issue = [&](int t0) {
            const int nE = ((T - t0 < FT) ? T - t0 : FT) * D;
            const ET* p = src + (int64_t)t0 * D;
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = lane + 64 * i;
                pre[i] = e < nE ? p[e] : ET(0);
            }
        };
        auto commit = [&]() {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const int e = lane + 64 * i;
                if (e < FT * D) tile[slot_of[e]] = pre[i];
            }
        };
        // the frame of one column, broadcast out of LDS into registers: exactly DX elements (a 16-byte read whose upper
        // part nobody uses leaves registers the allocator hands to temporaries -- and the first of those then waits for
        // the read)
        constexpr int NV = DX / VW, NT = DX % VW;   // 16-byte pieces of a frame + single elements behind them
        vec_t xv[NV > 0 ? NV : 1];
        ET xt[NT > 0 ? NT : 1];
        auto fetch = [&](int k) {
            const vec_t* xr = reinterpret_cast<const vec_t*>(tile + k * DV);
#pragma unroll
            for (int j = 0; j < NV; ++j) xv[j] = xr[j];
#pragma unroll
            for (int j = 0; j < NT; ++j) xt[j] = tile[k * DV + NV * VW + j];
        };
        issue(0);
        commit();
        fetch(0);

        double prev = INF;
        uint8_t* bp = WANT_BP ? a.bp + a.bp_off[slot] + rr : nullptr;
        double* co = WANT_COSTS ? a.costs + a.costs_off[u] + (int64_t)rr * T : nullptr;

        // one column: cost of the frame in xv under the lane's Gaussian (hmm_state.py:48-58), then the recurrence
        // (decode.py:97-124, as viterbi_chain_kernel)
        auto column = [&](int k, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            // (ONE accumulator chain: the other resident waves cover its latency, a second chain costs an add per cell)
            ET q0 = sizeof(ET) == 8 ? (ET)konst : ET(0);   // fp64: the accumulator starts at the normaliser
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                const ET e0 = fma_(d < NV * VW ? xv[d / VW][d % VW] : xt[d % VW], s[d], ms[d]);
                q0 = fma_(e0, e0, q0);
            }
            // The NEXT column's frame is requested here: its registers have just been read, and it travels during the
            // recurrence.  (The empty asm pins the order by a data dependence: left alone, the optimiser sinks the fma
            // block below the reads, keeps two frames in registers and copies one onto the other in every column.)
            asm volatile("" : "+v"(q0) :: "memory");
            fetch(k + 1);
            const double q = sizeof(ET) == 8 ? (double)q0 : konst + (double)q0;
            double c;
            uint8_t code = 3;
            if (FIRST) {
                const double e = (LIN && q > cthr) ? INF : q;
                c = is_start ? e : INF;                     // decode.py:99-101; every other cell of column 0 stays +inf
            } else {
                // beyond the underflow threshold the cost is +inf: ONE 32-bit select on the high word makes it +inf or a
                // NaN, and a NaN sum is cleaned to +inf below anyway
                const double e = LIN ? __hiloint2double((q > cthr) ? 0x7FF00000 : __double2hiint(q), __double2loint(q)) : q;
                const double p1 = wave_shr1z(prev);
                if (!WANT_BP) {
                    double best = vmin(c1 + p1, c0 + prev);     // (prev is never NaN: cleaned below)
                    if (SKIP) best = vmin(best, c2 + wave_shr1z(p1));
                    c = vmin(best + e, INF);                    // min(inf, nan) keeps inf (decode.py:124)
                } else {
                    double best = INF;
                    code = first_code;
                    if (SKIP) {
                        const double v2 = c2 + wave_shr1z(p1);
                        if (v2 < best) { best = v2; code = 2; }
                    }
                    const double v1 = c1 + p1;
                    if (v1 < best) { best = v1; code = 1; }
                    const double v0 = c0 + prev;
                    if (v0 < best) { best = v0; code = 0; }
                    c = best + e;
                    c = (c != c) ? INF : c;
                    if (first_code == 3) c = INF;               // row without arcs stays +inf (decode.py:116-117)
                }
            }
            prev = c;
            if (WANT_BP && act) { *bp = code; bp += R; }
            if (WANT_COSTS && act) { *co = c; co += 1; }
        };

        for (int t0 = 0; t0 < T; t0 += FT) {
            const bool more = t0 + FT < T;
            if (more) issue(t0 + FT);
            const int nF = (T - t0 < FT) ? T - t0 : FT;
            int k = 0;
            if (t0 == 0) { column(0, std::true_type()); k = 1; }
            for (; k < nF; ++k) column(k, std::false_type());
            if (more) { commit(); fetch(0); }
        }
        if (es >= 0) a.end_cost[u * a.n_end + es] = prev;
        if (fa.select_end) {
            // end selection in the sweep (one lane group holds every end row): the cheapest end, the LAST of equal minima
            // (decode.py:129-134 keeps an end when `best >= cost`) -- a butterfly over (cost, end slot), lanes without
            // an end row carry (+inf, -1) and lose every tie
            double v = es >= 0 ? prev : INF;
            int kk = es;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const double ov = __shfl_xor(v, o);
                const int ok = __shfl_xor(kk, o);
                const bool take = ov < v || (ov == v && ok > kk);
                v = take ? ov : v;
                kk = take ? ok : kk;
            }
            if (lane == 0) a.best_end[u] = kk;
        }
    }
}

// ---- PACKED form: a wave's 64 lanes are a WINDOW of `pw` consecutive rows of the endless sequence
//   [rows of launch slot 0 | rows of launch slot 1 | ...]          (R rows each, pw a multiple of the chains' length)
// so that a model of R = 50 rows (configs[0]: 10 words x 5 states) fills 60 of 64 lanes -- 12 whole chains, the last two of
// them the next utterance's -- instead of 50.  A window holds rows of at most TWO utterances (the host picks pw so); each
// has its own tile in LDS, a lane reads the frame of ITS utterance (two distinct addresses per broadcast read).  Both
// utterances advance together, column by column; the shorter one's lanes run on past its end on stale frames -- a chain
// never reads across a chain boundary (arc costs +inf there), so nothing of it reaches the longer one's rows -- and its
// final costs are taken when it ends.  Waves stride over the windows by a multiple of the pattern's period, so a lane keeps
// the same model row and loads its constants once.  End costs go to memory; the best end per utterance is a kernel of
// its own (an utterance's chains may sit in two waves).
constexpr int FT2 = 16;   // frames per tile and utterance in the packed form

template <typename ET, int DV, int DX, bool WANT_BP, bool WANT_COSTS, bool SKIP, bool LIN>
__global__ __launch_bounds__(64) void viterbi_fused_packed_kernel(gh_fused_args fa) {
    constexpr int NLD = (FT2 * DV + 63) / 64;
    constexpr int VW = 16 / sizeof(ET);
    constexpr int TILE = (FT2 + 1) * DV;       // elements of one utterance's tile (+1 row: read-ahead of the last column)
    static_assert(DV % VW == 0 && DX <= DV, "LDS rows are 16-byte multiples");
    typedef ET vec_t __attribute__((ext_vector_type(VW)));
    __shared__ __attribute__((aligned(16))) ET tile[2 * TILE];
    __shared__ uint16_t slot_of[NLD * 64];
    const gh_chain_args& a = fa.c;
    const int lane = threadIdx.x;
    const int D = fa.D, Rp = fa.Rp, R = a.R, PW = fa.pw;
    const int64_t n_utts = fa.n_items;          // (one lane group: items are utterances)
    const double INF = INFINITY;

    for (int e = lane; e < NLD * 64; e += 64) {
        const int fr = e / D;
        slot_of[e] = (uint16_t)(fr * DV + (e - fr * D));
    }
    for (int i = lane; i < 2 * TILE; i += 64) tile[i] = ET(0);

    ET s[DX], ms[DX];
    double konst = 0, cthr = 0, c0 = INF, c1 = INF, c2 = INF;
    uint8_t first_code = 3;
    bool is_start = false;
    int cur_row = -1, es = -1;

    for (int64_t w = blockIdx.x; w < fa.n_windows; w += gridDim.x) {
        const int64_t v0 = w * PW;
        const int64_t sA = v0 / R;
        const int o = (int)(v0 - sA * R);                  // first row of the window inside utterance A
        const bool inwin = lane < PW && (sA * R + o + lane) < n_utts * R;
        const bool isB = o + lane >= R;
        const int row = inwin ? (isB ? o + lane - R : o + lane) : 0;
        if (row != cur_row) {                              // (the stride keeps the pattern: true once per wave)
            cur_row = row;
            const double* p = fa.par + row;
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                s[d] = (ET)p[(size_t)d * Rp];
                ms[d] = (ET)p[(size_t)(fa.DVp + d) * Rp];
            }
            konst = p[(size_t)(2 * fa.DVp) * Rp];
            cthr = p[(size_t)(2 * fa.DVp + 1) * Rp];
            c0 = a.cost0[row];
            c1 = a.cost1[row];
            c2 = SKIP ? a.cost2[row] : INF;
            const uint8_t info = a.row_info[row];
            first_code = info & 3;
            is_start = (info & 4) != 0;
            es = a.end_slot[row];
            __builtin_amdgcn_s_waitcnt(0x0F70);
        }
        const bool hasB = (o + PW > R) && (sA + 1 < n_utts);        // wave-uniform
        const int64_t slotA = a.slot0 + sA, slotB = slotA + (hasB ? 1 : 0);
        const int64_t uA = a.perm ? a.perm[slotA] : slotA, uB = a.perm ? a.perm[slotB] : slotB;
        const int64_t fA = a.utt_off[uA], fB = a.utt_off[uB];
        const int TA = (int)(a.utt_off[uA + 1] - fA), TB = hasB ? (int)(a.utt_off[uB + 1] - fB) : 0;
        const int T1 = hasB ? (TA < TB ? TA : TB) : TA, T2 = TA > TB ? TA : TB;     // both run to T1, the longer one to T2
        if (T2 <= 0) continue;
        const ET* srcA = static_cast<const ET*>(fa.feats) + fA * D;
        const ET* srcB = static_cast<const ET*>(fa.feats) + fB * D;
        const int T_mine = isB ? TB : TA;
        const int64_t u_mine = isB ? uB : uA, slot_mine = isB ? slotB : slotA;
        const bool act = inwin && (!isB || hasB) && T_mine > 0;
        const int xbase = isB ? TILE : 0;

        ET pre[2][NLD];
        auto issue = [&](int t0) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int Tj = j ? TB : TA;
                int nF = Tj - t0;
                nF = nF < 0 ? 0 : (nF > FT2 ? FT2 : nF);
                const int nE = nF * D;
                const ET* p = (j ? srcB : srcA) + (int64_t)t0 * D;
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int e = lane + 64 * i;
                    pre[j][i] = e < nE ? p[e] : ET(0);
                }
            }
        };
        auto commit = [&]() {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int e = lane + 64 * i;
                    if (e < FT2 * D) tile[j * TILE + slot_of[e]] = pre[j][i];
                }
        };
        constexpr int NV = DX / VW, NT = DX % VW;
        vec_t xv[NV > 0 ? NV : 1];
        ET xt[NT > 0 ? NT : 1];
        auto fetch = [&](int k) {
            const ET* base = tile + xbase + k * DV;
            const vec_t* xr = reinterpret_cast<const vec_t*>(base);
#pragma unroll
            for (int j = 0; j < NV; ++j) xv[j] = xr[j];
#pragma unroll
            for (int j = 0; j < NT; ++j) xt[j] = base[NV * VW + j];
        };
        issue(0);
        commit();
        fetch(0);

        double prev = INF, fin = INF;
        uint8_t* bp = WANT_BP ? a.bp + a.bp_off[slot_mine] + row : nullptr;
        double* co = WANT_COSTS ? a.costs + a.costs_off[u_mine] + (int64_t)row * T_mine : nullptr;

        auto column = [&](int k, int t, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            ET q0 = sizeof(ET) == 8 ? (ET)konst : ET(0);
#pragma unroll
            for (int d = 0; d < DX; ++d) {
                const ET e0 = fma_(d < NV * VW ? xv[d / VW][d % VW] : xt[d % VW], s[d], ms[d]);
                q0 = fma_(e0, e0, q0);
            }
            asm volatile("" : "+v"(q0) :: "memory");
            fetch(k + 1);
            const double q = sizeof(ET) == 8 ? (double)q0 : konst + (double)q0;
            double c;
            uint8_t code = 3;
            if (FIRST) {
                const double e = (LIN && q > cthr) ? INF : q;
                c = is_start ? e : INF;
            } else {
                const double e = LIN ? __hiloint2double((q > cthr) ? 0x7FF00000 : __double2hiint(q), __double2loint(q)) : q;
                const double p1 = wave_shr1z(prev);
                if (!WANT_BP) {
                    double best = vmin(c1 + p1, c0 + prev);
                    if (SKIP) best = vmin(best, c2 + wave_shr1z(p1));
                    c = vmin(best + e, INF);
                } else {
                    double best = INF;
                    code = first_code;
                    if (SKIP) {
                        const double v2 = c2 + wave_shr1z(p1);
                        if (v2 < best) { best = v2; code = 2; }
                    }
                    const double v1 = c1 + p1;
                    if (v1 < best) { best = v1; code = 1; }
                    const double v0 = c0 + prev;
                    if (v0 < best) { best = v0; code = 0; }
                    c = best + e;
                    c = (c != c) ? INF : c;
                    if (first_code == 3) c = INF;
                }
            }
            prev = c;
            if ((WANT_BP || WANT_COSTS) && act && t < T_mine) {
                if (WANT_BP) { *bp = code; bp += R; }
                if (WANT_COSTS) { *co = c; co += 1; }
            }
        };

        for (int t0 = 0; t0 < T2; t0 += FT2) {
            const bool more = t0 + FT2 < T2;
            if (more) issue(t0 + FT2);
            const int nF = (T2 - t0 < FT2) ? T2 - t0 : FT2;
            int k = 0;
            if (t0 == 0) { column(0, 0, std::true_type()); k = 1; if (T1 == 1) fin = prev; }
            for (; k < nF; ++k) {
                column(k, t0 + k, std::false_type());
                if (t0 + k + 1 == T1) fin = prev;           // (wave-uniform test: the shorter utterance ends here)
            }
            if (more) { commit(); fetch(0); }
        }
        if (es >= 0 && act) a.end_cost[u_mine * a.n_end + es] = (T_mine == T2) ? prev : fin;
    }
}

// the cheapest end row per utterance, the LAST of equal minima (decode.py:129-134 keeps an end when `best >= cost`): one
// lane per utterance -- what chain_backtrace_kernel does with a wave per utterance when no path is wanted
__global__ void end_select_kernel(gh_chain_args a, int64_t u_begin, int64_t n_utts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_utts) return;
    const int64_t slot = u_begin + i;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int T = (int)(a.utt_off[u + 1] - a.utt_off[u]);
    double best = INFINITY;
    int bi = -1;
    for (int k = 0; k < a.n_end; ++k) {
        const double c = a.end_cost[u * a.n_end + k];
        if (best >= c) { best = c; bi = k; }
    }
    a.best_end[u] = T <= 0 ? -1 : bi;
}

template <typename ET, int DV, int DX, bool BP, bool CO, bool SK, bool LIN>
int launch_one(gh_ctx* ctx, const gh_fused_args& fa) {
    int occ = 0;
    if (fa.pw > 0) {       // packed windows (one lane group, uniform chains)
        GH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)viterbi_fused_packed_kernel<ET, DV, DX, BP, CO, SK, LIN>, 64, 0));
        if (occ < 1) occ = 1;
        int64_t grid = (int64_t)ctx->n_cu * occ;
        if (const char* e = getenv("GMMHMM_FUSED_WAVES")) grid = (int64_t)ctx->n_cu * std::max(1, atoi(e));
        grid = std::min<int64_t>(grid, fa.n_windows);
        if (grid > fa.period) grid -= grid % fa.period;        // (a lane keeps its model row from window to window)
        hipLaunchKernelGGL((viterbi_fused_packed_kernel<ET, DV, DX, BP, CO, SK, LIN>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, fa);
        GH_HIP(hipGetLastError());
        if (fa.select_end) {
            hipLaunchKernelGGL(end_select_kernel, dim3((unsigned)((fa.n_items + 255) / 256)), dim3(256), 0, ctx->stream, fa.c, fa.c.slot0, fa.n_items);
            GH_HIP(hipGetLastError());
        }
        return GH_OK;
    }
    GH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)viterbi_fused_kernel<ET, DV, DX, BP, CO, SK, LIN>, 64, 0));
    if (occ < 1) occ = 1;
    int64_t grid = (int64_t)ctx->n_cu * occ;
    if (const char* e = getenv("GMMHMM_FUSED_WAVES")) grid = (int64_t)ctx->n_cu * std::max(1, atoi(e));   // tuning knob
    grid = std::min<int64_t>(grid, fa.n_items);
    // (a multiple of the group count keeps a wave on ONE row group: its constants are loaded once)
    if (fa.c.n_groups > 1 && grid > fa.c.n_groups) grid -= grid % fa.c.n_groups;
    hipLaunchKernelGGL((viterbi_fused_kernel<ET, DV, DX, BP, CO, SK, LIN>), dim3((unsigned)grid), dim3(64), 0, ctx->stream, fa);
    GH_HIP(hipGetLastError());
    return GH_OK;
}

template <typename ET, int DV, int DX, bool SK, bool LIN>
int launch_mode(gh_ctx* ctx, const gh_fused_args& fa, bool want_bp, bool want_costs) {
    return want_costs ? launch_one<ET, DV, DX, true, true, SK, LIN>(ctx, fa)
           : want_bp  ? launch_one<ET, DV, DX, true, false, SK, LIN>(ctx, fa)
                      : launch_one<ET, DV, DX, false, false, SK, LIN>(ctx, fa);
}

template <typename ET, int DV, int DX>
int launch_dv(gh_ctx* ctx, const gh_fused_args& fa, bool want_bp, bool want_costs) {
    if (fa.skip) return fa.lin ? launch_mode<ET, DV, DX, true, true>(ctx, fa, want_bp, want_costs)
                               : launch_mode<ET, DV, DX, true, false>(ctx, fa, want_bp, want_costs);
    return fa.lin ? launch_mode<ET, DV, DX, false, true>(ctx, fa, want_bp, want_costs)
                  : launch_mode<ET, DV, DX, false, false>(ctx, fa, want_bp, want_costs);
}

template <typename ET>
int launch_et(gh_ctx* ctx, const gh_fused_args& fa, bool want_bp, bool want_costs) {
    constexpr int VW = 16 / sizeof(ET);   // LDS rows are multiples of 16 bytes
    const int D = fa.D;
    if (D == 13) return launch_dv<ET, (13 + VW - 1) / VW * VW, 13>(ctx, fa, want_bp, want_costs);    // BASELINE configs[0]
    if (D == 39) return launch_dv<ET, 40, 39>(ctx, fa, want_bp, want_costs);                         // MFCC + delta + delta-delta
    if (D <= 8) return launch_dv<ET, 8, 8>(ctx, fa, want_bp, want_costs);
    if (D <= 16) return launch_dv<ET, 16, 16>(ctx, fa, want_bp, want_costs);
    if (D <= 28) return launch_dv<ET, 28, 28>(ctx, fa, want_bp, want_costs);
    if (D <= 40) return launch_dv<ET, 40, 40>(ctx, fa, want_bp, want_costs);
    gh_set_error("gh_viterbi_fused: internal: no instantiation for %d dimensions", D);
    return GH_ERR_INVALID;
}

}  // namespace

// Window width of the packed form for a graph of R rows in chains of `unit` rows: the widest multiple of `unit` <= 64 whose
// windows never hold rows of more than two utterances (with g = gcd(pw, R): R - g + pw <= 2 R), provided it beats one
// utterance per wave; 0: keep the one-utterance form.  *period_out: windows after which the lane -> row map repeats.
int gh_fused_window(int R, int unit, int* period_out) {
    if (const char* e = getenv("GMMHMM_FUSED_PACK")) if (atoi(e) == 0) return 0;
    if (R <= 0 || unit <= 0 || R > 64 || R % unit) return 0;
    auto gcd = [](int x, int y) { while (y) { const int t = x % y; x = y; y = t; } return x; };
    int best = 0;
    for (int pw = (64 / unit) * unit; pw > R; pw -= unit)
        if (R - gcd(pw, R) + pw <= 2 * R) { best = pw; break; }
    if (best == 0 && 2 * R <= 64) best = 2 * R;          // small models: two whole utterances per wave
    if (best <= R) return 0;
    *period_out = R / gcd(best, R);
    return best;
}

// rows of the constants table per half (>= the dimensions any instantiation that takes D computes); 0: D not covered
int gh_fused_dv(int D) { return D <= 8 ? 8 : D <= 16 ? 16 : D <= 28 ? 28 : D <= 40 ? 40 : 0; }

int gh_launch_fused_params(gh_ctx* ctx, const gh_gmm* g, const int32_t* d_row_state, int R, int Rp, int DVp, double thr,
                           double* d_par) {
    hipLaunchKernelGGL(fused_params_kernel, dim3((unsigned)((Rp + 63) / 64)), dim3(64), 0, ctx->stream, g->dMean, g->dIvar, g->dLogc,
                       d_row_state, R, Rp, g->D, DVp, thr, d_par);
    GH_HIP(hipGetLastError());
    return GH_OK;
}

int gh_launch_viterbi_fused(gh_ctx* ctx, const gh_fused_args& fa, int64_t u_begin, int64_t n_utts, bool f64,
                            bool want_bp, bool want_costs) {
    if (n_utts <= 0) return GH_OK;
    gh_fused_args b = fa;
    b.c.slot0 = u_begin;
    b.n_items = n_utts * fa.c.n_groups;
    if (b.pw > 0) b.n_windows = (n_utts * (int64_t)fa.c.R + b.pw - 1) / b.pw;
    return f64 ? launch_et<double>(ctx, b, want_bp, want_costs) : launch_et<float>(ctx, b, want_bp, want_costs);
}
