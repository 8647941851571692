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
//     distance is 2 D fma:  e_d = fma(x_d, s_d, ms_d);  q = fma(e_d, e_d, q)   (no division, no subtraction);
//   * the frame x_t is the same for all 64 lanes: a tile of FT frames is staged ONCE in LDS (coalesced 8-byte loads of
//     the row-major [frames, D] matrix: the only HBM traffic of the sweep) and read back as LDS broadcasts;
//   * the next tile travels from HBM into registers while the current one is computed; a workgroup is one wave, so
//     there is no barrier anywhere;
//   * a wave walks several utterances (serpentine over the longest-first order) and loads its 2 D + 2 constants once.
// The kernel is bound by the fp64 vector pipe (2 D fma + ~12 DP instructions per cell), not by HBM: the roofline entry
// of bench.py reports both fractions.
#include "gh_internal.h"
#include "gh_viterbi.h"

namespace {

constexpr int FT = 32;  // frames per LDS tile

// lane i <- lane i-1 (lane 0 keeps `fill`)
__device__ __forceinline__ double wave_shr1(double v, double fill) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), 0x138, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// sqrt(1/(2 var)), -mean sqrt(1/(2 var)), -logc and the underflow threshold of every lattice row, k-major ([2 DV + 2][Rp])
// so that a wave's loads are contiguous.  Rows >= R repeat row 0 (never active).
__global__ void fused_params_kernel(const double* __restrict__ mean, const double* __restrict__ ivar,
                                    const double* __restrict__ logc, const int32_t* __restrict__ row_state, int R, int Rp,
                                    int D, int DV, double thr, double* __restrict__ par) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= Rp) return;
    const int st = row_state[r < R ? r : 0];
    for (int d = 0; d < DV; ++d) {
        double s = 0.0, ms = 0.0;
        if (d < D) {
            s = sqrt(0.5 * ivar[(size_t)st * D + d]);
            ms = -(mean[(size_t)st * D + d] * s);
        }
        par[(size_t)d * Rp + r] = s;
        par[(size_t)(DV + d) * Rp + r] = ms;
    }
    const double konst = -logc[st];
    par[(size_t)(2 * DV) * Rp + r] = konst;
    // GMM.evaluate works in the linear domain (hmm_state.py:114-120): np.exp(-q) is 0 below ln 2^-1075 whatever the
    // normaliser is, and so is w * norm * exp(-q) once the product is: +inf when max(q, konst + q) > thr
    par[(size_t)(2 * DV + 1) * Rp + r] = thr - (konst > 0.0 ? konst : 0.0);
}

// EXACT: the feature dimension IS DV (the LDS tile is the memory image); otherwise D <= DV, LDS rows are padded to DV
// with zeros and a table (in LDS) maps a tile element to its slot.
template <typename ET, int DV, bool EXACT, bool WANT_BP, bool WANT_COSTS>
__global__ __launch_bounds__(64) void viterbi_fused_kernel(gh_fused_args fa) {
    constexpr int NLD = (FT * DV + 63) / 64;   // tile elements a lane moves (upper bound: D <= DV)
    __shared__ __attribute__((aligned(16))) ET tile[FT * DV];
    __shared__ uint16_t slot_of[EXACT ? 1 : NLD * 64];
    const gh_chain_args& a = fa.c;
    const int lane = threadIdx.x;
    const int D = EXACT ? DV : fa.D, Rp = fa.Rp, R = a.R;
    const double INF = INFINITY;

    if (!EXACT) {
        // LDS slot (row stride DV) of tile element e (row stride D in memory)
        for (int e = lane; e < NLD * 64; e += 64) {
            const int fr = e / D;
            slot_of[e] = (uint16_t)(fr * DV + (e - fr * D));
        }
        for (int i = lane; i < FT * DV; i += 64) tile[i] = ET(0);   // the pad dimensions stay zero for good
    }

    ET s[DV], ms[DV];
    double konst = 0, qthr = 0, c0 = INF, c1 = INF, c2 = INF;
    uint8_t first_code = 3;
    bool is_start = false, act = false;
    int cur_g = -1, rr = 0, es = -1;

    const int64_t G = gridDim.x;
    for (int64_t round = 0;; ++round) {
        // serpentine over the longest-first launch order: the waves' frame totals stay within one utterance of each other
        const int64_t item = round * G + ((round & 1) ? G - 1 - blockIdx.x : blockIdx.x);
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
            for (int d = 0; d < DV; ++d) {
                s[d] = (ET)p[(size_t)d * Rp];
                ms[d] = (ET)p[(size_t)(DV + d) * Rp];
            }
            konst = p[(size_t)(2 * DV) * Rp];
            qthr = p[(size_t)(2 * DV + 1) * Rp];
            c0 = act ? a.cost0[rr] : INF;
            c1 = act ? a.cost1[rr] : INF;
            c2 = (fa.skip && act) ? a.cost2[rr] : INF;
            const uint8_t info = act ? a.row_info[rr] : 0x0F;
            first_code = info & 3;
            is_start = (info & 4) != 0;
            es = act ? a.end_slot[rr] : -1;
        }
        const int64_t u = a.perm ? a.perm[slot] : slot;
        const int64_t f0 = a.utt_off[u];
        const int T = (int)(a.utt_off[u + 1] - f0);
        if (T <= 0) continue;
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
                if (e < FT * D) tile[EXACT ? e : (int)slot_of[e]] = pre[i];
            }
        };
        issue(0);
        commit();

        double prev = INF;
        uint8_t* bp = WANT_BP ? a.bp + a.bp_off[slot] + rr : nullptr;
        double* co = WANT_COSTS ? a.costs + a.costs_off[u] + (int64_t)rr * T : nullptr;

        for (int t0 = 0; t0 < T; t0 += FT) {
            const bool more = t0 + FT < T;
            if (more) issue(t0 + FT);
            const int nF = (T - t0 < FT) ? T - t0 : FT;
            for (int k = 0; k < nF; ++k) {
                // ---- distance of frame t0 + k to the lane's Gaussian (hmm_state.py:48-58) ----
                const ET* xr = tile + k * DV;
                ET q0 = 0, q1 = 0;
#pragma unroll
                for (int d = 0; d + 1 < DV; d += 2) {
                    const ET e0 = __builtin_fma(xr[d], s[d], ms[d]);
                    const ET e1 = __builtin_fma(xr[d + 1], s[d + 1], ms[d + 1]);
                    q0 = __builtin_fma(e0, e0, q0);
                    q1 = __builtin_fma(e1, e1, q1);
                }
                if (DV & 1) {
                    const ET e0 = __builtin_fma(xr[DV - 1], s[DV - 1], ms[DV - 1]);
                    q0 = __builtin_fma(e0, e0, q0);
                }
                const double q = (double)(q0 + q1);
                double e = konst + q;
                if (q > qthr) e = INF;
                // ---- the column of the dynamic program (decode.py:97-124), as viterbi_chain_kernel ----
                const int t = t0 + k;
                const double p1 = wave_shr1(prev, INF);
                double best = INF;
                uint8_t code = first_code;
                if (fa.skip) {
                    const double p2 = wave_shr1(p1, INF);
                    const double v2 = c2 + p2;
                    if (v2 < best) { best = v2; code = 2; }
                }
                const double v1 = c1 + p1;
                if (v1 < best) { best = v1; code = 1; }
                const double v0 = c0 + prev;
                if (v0 < best) { best = v0; code = 0; }
                double c = best + e;
                c = (c != c) ? INF : c;                      // min(inf, nan) keeps inf (decode.py:124)
                if (first_code == 3) c = INF;                // row without arcs stays +inf (decode.py:116-117)
                if (t == 0 && is_start) { c = e; code = 3; } // decode.py:99-101
                prev = c;
                if (WANT_BP && act) { *bp = code; bp += R; }
                if (WANT_COSTS && act) { *co = c; co += 1; }
            }
            if (more) commit();
        }
        if (es >= 0) a.end_cost[u * a.n_end + es] = prev;
    }
}

template <typename ET, int DV, bool EXACT>
int launch_dv(gh_ctx* ctx, const gh_fused_args& fa, bool want_bp, bool want_costs) {
    int occ = 0;
    const void* fn = want_costs ? (const void*)viterbi_fused_kernel<ET, DV, EXACT, true, true>
                     : want_bp  ? (const void*)viterbi_fused_kernel<ET, DV, EXACT, true, false>
                                : (const void*)viterbi_fused_kernel<ET, DV, EXACT, false, false>;
    GH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 64, 0));
    if (occ < 1) occ = 1;
    int64_t grid = (int64_t)ctx->n_cu * occ;
    if (const char* e = getenv("GMMHMM_FUSED_WAVES")) grid = (int64_t)ctx->n_cu * std::max(1, atoi(e));   // tuning knob
    grid = std::min<int64_t>(grid, fa.n_items);
    // (a multiple of the group count keeps a wave on ONE row group: its constants are loaded once)
    if (fa.c.n_groups > 1 && grid > fa.c.n_groups) grid -= grid % fa.c.n_groups;
    dim3 g((unsigned)grid), blk(64);
    if (want_costs) hipLaunchKernelGGL((viterbi_fused_kernel<ET, DV, EXACT, true, true>), g, blk, 0, ctx->stream, fa);
    else if (want_bp) hipLaunchKernelGGL((viterbi_fused_kernel<ET, DV, EXACT, true, false>), g, blk, 0, ctx->stream, fa);
    else hipLaunchKernelGGL((viterbi_fused_kernel<ET, DV, EXACT, false, false>), g, blk, 0, ctx->stream, fa);
    GH_HIP(hipGetLastError());
    return GH_OK;
}

template <typename ET>
int launch_et(gh_ctx* ctx, const gh_fused_args& fa, int DV, bool want_bp, bool want_costs) {
    switch (DV) {
        case 13: return launch_dv<ET, 13, true>(ctx, fa, want_bp, want_costs);    // BASELINE configs[0]
        case 39: return launch_dv<ET, 39, true>(ctx, fa, want_bp, want_costs);    // MFCC + delta + delta-delta
        case 8: return launch_dv<ET, 8, false>(ctx, fa, want_bp, want_costs);
        case 16: return launch_dv<ET, 16, false>(ctx, fa, want_bp, want_costs);
        case 26: return launch_dv<ET, 26, false>(ctx, fa, want_bp, want_costs);
        case 40: return launch_dv<ET, 40, false>(ctx, fa, want_bp, want_costs);
    }
    gh_set_error("gh_viterbi_fused: internal: no instantiation for %d dimensions", DV);
    return GH_ERR_INVALID;
}

}  // namespace

int gh_fused_dv(int D) { return D == 13 || D == 39 ? D : D <= 8 ? 8 : D <= 16 ? 16 : D <= 26 ? 26 : D <= 40 ? 40 : 0; }

int gh_launch_fused_params(gh_ctx* ctx, const gh_gmm* g, const int32_t* d_row_state, int R, int Rp, int DV, double thr,
                           double* d_par) {
    hipLaunchKernelGGL(fused_params_kernel, dim3((unsigned)((Rp + 63) / 64)), dim3(64), 0, ctx->stream, g->dMean, g->dIvar, g->dLogc,
                       d_row_state, R, Rp, g->D, DV, thr, d_par);
    GH_HIP(hipGetLastError());
    return GH_OK;
}

int gh_launch_viterbi_fused(gh_ctx* ctx, const gh_fused_args& fa, int DV, int64_t u_begin, int64_t n_utts, bool f64,
                            bool want_bp, bool want_costs) {
    if (n_utts <= 0) return GH_OK;
    gh_fused_args b = fa;
    b.c.slot0 = u_begin;
    b.n_items = n_utts * fa.c.n_groups;
    return f64 ? launch_et<double>(ctx, b, DV, want_bp, want_costs) : launch_et<float>(ctx, b, DV, want_bp, want_costs);
}
