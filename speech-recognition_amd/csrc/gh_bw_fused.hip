// Baum-Welch sufficient statistics on the matrix cores end to end, for batches whose forward-backward ran on one-word
// chain graphs (gh_fbchain: isolated-word EM, BASELINE configs[2]).  NOT in the reference (it trains by Viterbi
// alignment); the statistics are those of gh_bw_accumulate (include/gmmhmm.h):
//     r_nsm = gamma_ns * w_sm pdf_sm(x_n) / sum_m' w_sm' pdf_sm'(x_n)
//     stats[s, m] = [ sum r | sum r (x - mean_sm) | sum r (x - mean_sm)^2 ].
//
// What the first version (bw_stats_kernel, gh_train.hip) paid for: component densities on the VALU (lane = (frame,
// component), 39 x 3 flops each), a frame x state occupancy matrix [N, S] that is cleared and 90 % zeros, and a
// read-modify-write of the workgroup's 253 KB slab per (utterance, state).  Here
//   * gamma comes straight from fb_chain_kernel as a compact [N, 8] matrix (one column per chain row);
//   * the component log-densities are a GEMM, D[frame, comp] = C[comp] + Z'[frame, :] . P[comp, :] with
//     Z' = [x^2 | x] -- v_mfma_f64_16x16x4 with the FRAMES on the rows, so the result registers (lane = component,
//     registers = 4 frames) are directly the B operand of the accumulation GEMM: nothing is transposed, nothing
//     goes through LDS between the two GEMMs;
//   * the log-sum-exp over a state's 8 components runs across 8 lanes (two quad_perm steps + row_half_mirror), in the
//     scaled log domain of the likelihood kernel (exp2 by table + degree-4 polynomial, gh_loglik_mfma.hip);
//   * accumulation G^T[Zcol, comp] += Z[frame, Zcol] * r[frame, comp] with Z = [x - c, 1 | (x - c)^2] (c = a centre
//     shared by the two states a wave owns; the component-centred sums follow algebraically at the very end; the two
//     halves are padded to whole 16-column tiles so that a tile is either linear or squared for every lane, and the
//     ones column is the first padding column of the linear half: x = 0 there, "centre" -1),
//     accumulators PERSISTENT in registers over all utterances of the workgroup: utterances are grouped by graph
//     (= word), a wave owns a PAIR of states (16 = 2 x 8 component columns), a workgroup = the pairs of one word;
//   * one raw [80, 16] tile per wave leaves the workgroup; a small kernel sums the tiles of a pair over its
//     workgroups in a fixed order and converts them (deterministic, no float atomics).
// Mixtures of more than 8 components (WIDE; BASELINE configs[3]: 16 states x 32 mixtures per word): a wave owns 16
// CONSECUTIVE COMPONENTS OF ONE STATE (a "column group": state r, components 16 c .. 16 c + 15), so a state's mixture
// spans several waves and the normaliser sum_m' w_m' pdf_m'(x) cannot be formed inside one; it is not recomputed either:
// it IS the state's likelihood, which the likelihood kernel wrote into the batch's [N, S] matrix for the forward-backward
// that produced gamma (same model: gh_gmm::serial == gh_batch::nll_serial is checked).  r = gamma exp(log(w pdf) + nll).
#include "gh_internal.h"
#include "gh_host.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

struct bwf_wg { int32_t graph, u_begin, u_end, pad; };           // utterances ulist[u_begin, u_end) of one graph
struct bwf_pair { int32_t sa, sb, wg_begin, wg_end, p, m0; };     // column group p of a graph (states sa, sb (or -1), or -- WIDE --
                                                                  // components m0 .. m0 + 15 of state sa), its workgroups

__device__ __forceinline__ double bwf_vmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <int CTRL> __device__ __forceinline__ double bwf_dpp(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// all-reduce over the 8 lanes {8g .. 8g+7}: xor 1 (quad_perm 1,0,3,2), xor 2 (quad_perm 2,3,0,1), then the mirror image
// inside the half row (lane i <-> 7 - i): every lane of one quad meets a lane of the other
__device__ __forceinline__ double max8(double v) {
    v = bwf_vmax(v, bwf_dpp<0xB1>(v));
    v = bwf_vmax(v, bwf_dpp<0x4E>(v));
    v = bwf_vmax(v, bwf_dpp<0x141>(v));
    return v;
}
__device__ __forceinline__ double sum8(double v) {
    v += bwf_dpp<0xB1>(v);
    v += bwf_dpp<0x4E>(v);
    v += bwf_dpp<0x141>(v);
    return v;
}
// 2^(y/128) for finite y <= 0 (or NaN): table + degree-4 polynomial (see gh_loglik_mfma.hip, exp2s)
__device__ __forceinline__ double bwf_exp2s(double y, const double* __restrict__ tab) {
    const double n = __builtin_rint(y);
    const double r = y - n;
    const int ni = (int)n;
    const double t = tab[ni & 127];
    double p = fma(r, 3.583032305400251285e-11, 2.6466421444330968834e-08);
    p = fma(p, r, 1.4662262387640424337e-05);
    p = fma(p, r, 5.4152123481245727298e-03);
    p = p * r;
    return __builtin_ldexp(fma(t, p, t), ni >> 7);
}

// KS = k-steps of the density GEMM (K = 4 KS = 2 KP), LT = 16-column tiles of each half of Z (D + 1 <= 16 LT)
// One WAVE per (utterance group of a word, state pair): the wave walks its utterances in 16-frame blocks, looks at the
// pair's two gamma columns first and SKIPS a block whose 16 frames carry no occupancy for either state -- posteriors are
// sharp: a few frames away from the aligned region gamma underflows to exactly 0.0 (fb_chain_kernel returns alpha beta / P
// as a double), so of the 21 (block, pair) combinations of a 100-frame, 5-state utterance ~9 do any work, and a skipped
// block would only have added exact zeros.  Blocks that do work are staged by the wave itself (16 x D frames through LDS
// into the two MFMA operand layouts); no other wave is involved: no barrier, no idle "staging" wave, no waiting for the
// slowest pair of a shared tile.
// (Round 2 / early round 3: a 256-thread workgroup per utterance group -- three waves = three pairs sharing a staged
// 32-frame tile, a fourth wave that only staged, two barriers per tile.  Diagnostic builds showed that kernel pipe bound --
// density MFMAs 28 %, accumulation MFMAs 32 %, responsibilities 19 %, HBM 5 %, barriers 1.5 % of its 0.72 ms -- so the way
// down was less work, not fewer stalls; skipping zero blocks inside that structure only gained 14 % because the waves of a
// workgroup still met at every tile.)
// NORM: the mixture normaliser comes from the batch's likelihood matrix (always with WIDE; with state pairs when the caller
// vouches for the matrix -- the device-resident session, or matching model / matrix stamps): r = gamma 2^((y + K nll) / 128),
// no maximum, no sum across the 8 lanes, no reciprocal (~24 of the ~45 VALU instructions per frame register).
template <int KS, int LT, bool WIDE, bool NORM>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void bw_fused_kernel(const double* __restrict__ X, int D, int M,
                                                     const double* __restrict__ mean, const double* __restrict__ ivar,
                                                     const double* __restrict__ logc, const double* __restrict__ gam,
                                                     double occ_floor, int gam_stride, int gam_by_state, const int64_t* __restrict__ seg_first,
                                                     const int32_t* __restrict__ seg_len,
                                                     const int32_t* __restrict__ ulist, const bwf_wg* __restrict__ wgs,
                                                     const gh_fbchain* __restrict__ chains, const double* __restrict__ tables,
                                                     double* __restrict__ partial, int slot_shift,
                                                     const double* __restrict__ nll, int nll_S, const int32_t* __restrict__ rng) {
    constexpr int KP = 2 * KS;            // padded feature length
    constexpr int TF = 16;                // frames per block
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int DP = KP | 1;                // odd LDS row stride >= KP: columns D .. KP-1 stay zero, so the operand reads
                                          // below need no bounds test
    double* xt = sm;                      // [TF][DP]
    double* gt = xt + TF * DP;            // [TF][2]  gamma of the block's frames for the pair's two states
    double* nt = gt + TF * 2;             // [TF][2]  NORM: K x the states' negative log-likelihoods of the block's frames
    double* tab = nt + (NORM ? TF * 2 : 0);   // [128]    2^(j/128)
    const int lane = threadIdx.x;
    const int j = lane & 15, q = lane >> 4;
    const int p = blockIdx.x & ((1 << slot_shift) - 1);   // column group (slots beyond the word's groups leave)
    const bwf_wg wg = wgs[blockIdx.x >> slot_shift];
    const gh_fbchain* ch = chains + wg.graph;
    const int n = ch->n;
    const int chunks = WIDE ? (M + 15) >> 4 : 1;          // 16-component pieces of a state's mixture
    const int row_a = WIDE ? p / chunks : 2 * p;           // chain row(s) of this wave
    const int row_b = (!WIDE && 2 * p + 1 < n) ? 2 * p + 1 : -1;
    if (row_a >= n) return;
    for (int i = lane; i < 128; i += 64) tab[i] = tables[i];
    for (int i = lane; i < TF * DP; i += 64) xt[i] = 0.0;
    // ---- this wave's pair of states: operands that stay in registers for all its utterances ----
    const int sa = ch->state[row_a];
    const int sb = (row_b >= 0) ? ch->state[row_b] : -1;
    const int s_j = (WIDE || j < 8) ? sa : sb;
    const int m_j = WIDE ? (p - row_a * chunks) * 16 + j : (j & 7);
    const bool valid = s_j >= 0 && m_j < M;
    const int64_t g_j = valid ? (int64_t)s_j * M + m_j : 0;
    // (round 4: the set-up used to be ~30 dependent memory round trips per wave -- every operand under `if (valid && d < D)`
    //  was followed by s_waitcnt vmcnt(0), and sum mu^2 / var walked its D dimensions one load pair at a time.  Now: this
    //  lane's KS / 2 dimensions d = 4 ks + q of mean and 1 / var in ONE batch of unconditional loads from clamped indices;
    //  they are both operands AND this lane's share of sum mu^2 / var, which two shuffles add up over the four lane groups.)
    double P[KS];
    double mu_r[KS / 2], iv_r[KS / 2];
#pragma unroll
    for (int ks = 0; ks < KS / 2; ++ks) {
        const int d = 4 * ks + q;
        const int64_t at = g_j * D + (d < D ? d : 0);
        mu_r[ks] = mean[at];
        iv_r[ks] = ivar[at];
    }
    const double lc = logc[g_j];
    double sm2 = 0.0;
#pragma unroll
    for (int ks = 0; ks < KS / 2; ++ks) {
        const bool in = valid && 4 * ks + q < D;
        const double iv = in ? iv_r[ks] : 0.0, mu = in ? mu_r[ks] : 0.0;
        P[ks] = -0.5 * iv * GH_LSE_SCALE64;                    // k = 4 ks + q < KP: the x^2 half
        P[KS / 2 + ks] = mu * iv * GH_LSE_SCALE64;            // k - KP: the x half, same dimension
        sm2 = fma(mu * iv, mu, sm2);
    }
    sm2 += __shfl_xor(sm2, 16);
    sm2 += __shfl_xor(sm2, 32);
    double Cj = GH_LSE_OFF64;
    if (valid) {
        const double c = lc - 0.5 * sm2;
        Cj = (c == -INFINITY) ? GH_LSE_OFF64 : bwf_vmax(c * GH_LSE_SCALE64, GH_LSE_OFF64);
    }
    // accumulation operand A = Z^T: this lane feeds column 16 t + j of the linear tiles (x[d] - c[d]; d = D: the ones
    // column, read from the zero padding with "centre" -1) and of the squared tiles
    constexpr int NCT = 2 * LT;
    int dd[LT];
    double cs[LT];
    v4d acc[NCT];
#pragma unroll
    for (int t = 0; t < LT; ++t) {
        const int d = t * 16 + j;
        dd[t] = (d <= D && d < DP) ? d : D;                     // columns behind the ones column read the zero padding too
        const double cm = mean[(int64_t)sa * M * D + (d < D ? d : 0)];   // (unconditional: see above)
        cs[t] = (d < D) ? cm : (d == D ? -1.0 : 0.0);
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = (v4d){0, 0, 0, 0};
    // gamma columns of the pair: compact [N, 8] (column = chain row) or the frame x state occupancy matrix
    // (gam_by_state: stride S, column = the chain row's state -- sequence-form forward-backward)
    const int st = q & 1;                                       // lanes (frame j, state st): the upper half repeats the lower
    const int grow_l = st ? row_b : row_a;
    const int gcol = (grow_l >= 0) ? (gam_by_state ? ch->state[grow_l] : grow_l) : -1;
    constexpr int PX = (TF * KP + 63) / 64;                    // frame elements per lane of a staged block
    const int gcol_a = gam_by_state ? sa : row_a;
    const int gcol_b = (row_b >= 0) ? (gam_by_state ? sb : row_b) : -1;
    __syncthreads();
    double px[PX], pg = 0.0, pn = 0.0;                         // the staged-ahead block: frames, gamma (and likelihoods) in registers
    // block [bf, bf + 16) of the segment (f0, T) -> registers (its X rows, the pair's gamma, WIDE: the state's likelihoods)
    auto fetch = [&](int64_t f0, int T, int bf) {
        const int nf_ = (T - bf < TF) ? T - bf : TF;
        const double* src = X + (f0 + bf) * D;
#pragma unroll
        for (int e = 0; e < PX; ++e) {
            const int i = lane + 64 * e;
#ifdef BWF_NOSTAGE   // diagnostic builds (tools/bwf_variants.sh): timing without HBM reads / MFMAs / responsibilities
            px[e] = (i < nf_ * D) ? 0.25 * (double)(i & 15) : 0.0;
#else
            px[e] = (i < nf_ * D) ? src[i] : 0.0;
#endif
        }
#ifdef BWF_NOSTAGE
        pg = (j < nf_ && gcol >= 0) ? 0.125 : 0.0;
#else
        pg = (j < nf_ && gcol >= 0) ? gam[(f0 + bf + j) * gam_stride + gcol] : 0.0;
#endif
        if (NORM) {   // lanes (frame j, state st) like gamma
            const int s_n = st ? sb : sa;
            pn = (lane < 32 && j < nf_ && s_n >= 0) ? nll[(f0 + bf + j) * (int64_t)nll_S + s_n] : 0.0;
        }
    };
    // the fetched block -> LDS [16][DP] (rows >= nf zero) + gamma [16][2] (+ likelihoods [16])
    auto park = [&]() {
#pragma unroll
        for (int e = 0; e < PX; ++e) {
            const int i = lane + 64 * e;
            if (i < TF * D) { const int f = i / D, d = i - f * D; xt[f * DP + d] = px[e]; }
        }
        if (lane < 32) gt[j * 2 + st] = pg;
        if (NORM && lane < 32) nt[j * 2 + st] = pn * GH_LSE_SCALE64;
    };
    // densities, responsibilities and accumulation of the parked block (nf frames)
    auto compute = [&](int nf) {
        // ---- component log-densities of 16 frames x 16 components (scaled log domain) ----
        v4d da = (v4d){Cj, Cj, Cj, Cj};
        const double* xr = xt + j * DP + q;              // A operand: row = frame j, columns q, q + 4, ...
        double xa[KS / 2];
#pragma unroll
        for (int ks = 0; ks < KS / 2; ++ks) xa[ks] = xr[4 * ks];             // (KS is even: k < KP <=> ks < KS / 2)
#ifdef BWF_NODENS
#pragma unroll
        for (int ks = 0; ks < KS / 2; ++ks) da[ks & 3] += xa[ks] * P[ks];
#else
#pragma unroll
        for (int ks = 0; ks < KS / 2; ++ks) da = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ks] * xa[ks], P[ks], da, 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < KS / 2; ++ks) da = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[ks], P[KS / 2 + ks], da, 0, 0, 0);
#endif
        // ---- responsibilities: lane = component j, register r = frame q + 4 r ----
        double R[4];
#ifdef BWF_NOEPI
#pragma unroll
        for (int r = 0; r < 4; ++r) R[r] = da[r];
#else
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int fr = q + 4 * r;
            const double y = da[r];
            if (NORM) {
                // log(w pdf) - log sum_m' w pdf = y + K nll: the likelihood kernel's own normaliser
                const int hs = WIDE ? 0 : (j >> 3);
                const double e = bwf_exp2s(y + nt[fr * 2 + hs], tab);
                const double g = gt[fr * 2 + hs];
                const double wgt = ((g > occ_floor) | (g != g)) ? g : 0.0;
                R[r] = (valid & (fr < nf) & (wgt != 0.0)) ? e * wgt : 0.0;
            } else {
                const double mx = max8(y);
                const double e = bwf_exp2s(y - mx, tab);
                const double s8 = sum8(e);
                const double g = gt[fr * 2 + (j >> 3)];
                const double wgt = ((g > occ_floor) | (g != g)) ? g : 0.0;
                double inv = __builtin_amdgcn_rcp(s8);
                inv = fma(fma(-s8, inv, 1.0), inv, inv);      // two Newton steps: full double accuracy
                inv = fma(fma(-s8, inv, 1.0), inv, inv);
                const double rv = e * (wgt * inv);
                R[r] = (valid & (fr < nf) & (wgt != 0.0)) ? rv : 0.0;   // (every component off: s8 = 8, e = 1 -- killed by `valid`)
            }
        }
#endif
        // ---- accumulate: G^T[Zcol, comp] += Z[frame, Zcol] r[frame, comp], k-step r = frames {0..3} + 4 r ----
        double zx[4][LT];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < LT; ++t) zx[r][t] = xt[(q + 4 * r) * DP + dd[t]];   // all reads in flight
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < LT; ++t) {
                const double xv = zx[r][t] - cs[t];
#ifdef BWF_NOACC
                acc[t][r] += xv * R[r];
#else
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv, R[r], acc[t], 0, 0, 0);
                // (the ones column squares to 1 as well: row D of the squared half holds sum r, unused)
                acc[LT + t] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv * xv, R[r], acc[LT + t], 0, 0, 0);
#endif
            }
    };
    if (rng) {
        // ---- block lists from the forward-backward's occupancy ranges: 64 utterances at a time, lane = utterance ----
        // (rng[u][row] = first / last frame of the row with gamma above the floor: no pass over gamma, and the block that
        //  travels from HBM while the current one is computed may belong to the NEXT utterance -- with the gamma scan below a
        //  wave pays two memory round trips per utterance before its first MFMA)
        for (int ui0 = wg.u_begin; ui0 < wg.u_end; ui0 += 64) {
            const bool have = ui0 + lane < wg.u_end;
            const int32_t u_l = have ? ulist[ui0 + lane] : 0;
            const int64_t f0_l = have ? seg_first[u_l] : 0;
            const int T_l = have ? seg_len[u_l] : 0;
            int lo_l = 0x7fffffff, hi_l = -1;
            if (have) {
                const int32_t* ra = rng + ((int64_t)u_l * GH_FBCHAIN_MAX + row_a) * 2;
                lo_l = ra[0]; hi_l = ra[1];
                if (row_b >= 0) {
                    const int32_t* rb = rng + ((int64_t)u_l * GH_FBCHAIN_MAX + row_b) * 2;
                    const int lb = rb[0], hb = rb[1];
                    if (hb >= lb) { lo_l = (hi_l >= lo_l && lo_l < lb) ? lo_l : lb; hi_l = (hi_l > hb) ? hi_l : hb; }
                }
                if (hi_l >= T_l) hi_l = T_l - 1;
                if (lo_l < 0) lo_l = 0;
            }
            unsigned long long mask = __ballot(have && hi_l >= lo_l);
            const int f0lo_l = (int)(f0_l & 0xffffffffll), f0hi_l = (int)(f0_l >> 32);
            // the item being computed (c*) and the one in flight (n*): segment, block, last block of the segment
            int64_t nf0 = 0;
            int cT = 0, cb = 0, clast = -1, nT = 0, nb = 0, nlast = -1;
            bool nvalid = false;
            auto open_next_utt = [&]() {       // next utterance of the batch with a non-empty range -> (n*)
                if (!mask) { nvalid = false; return; }
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                const int sl = __builtin_amdgcn_readfirstlane(l);
                nf0 = ((int64_t)__builtin_amdgcn_readlane(f0hi_l, sl) << 32) | (uint32_t)__builtin_amdgcn_readlane(f0lo_l, sl);
                nT = __builtin_amdgcn_readlane(T_l, sl);
                nb = __builtin_amdgcn_readlane(lo_l, sl) / TF;
                nlast = __builtin_amdgcn_readlane(hi_l, sl) / TF;
                nvalid = true;
            };
            open_next_utt();
            if (nvalid) fetch(nf0, nT, nb * TF);
            while (nvalid) {
                cT = nT; cb = nb; clast = nlast;
                park();
                if (cb < clast) nb = cb + 1; else open_next_utt();
                if (nvalid) fetch(nf0, nT, nb * TF);     // travels from HBM while this block is computed
                __syncthreads();                         // (one wave: orders the LDS stores above before the operand reads below)
                compute((cT - cb * TF < TF) ? cT - cb * TF : TF);
                __syncthreads();                         // (the next block's stores come after these reads)
            }
        }
    } else
    for (int ui = wg.u_begin; ui < wg.u_end; ++ui) {
        const int32_t u = ulist[ui];
        const int64_t f0 = seg_first[u];
        const int T = seg_len[u];
        // 1024 frames (64 blocks) at a time: lane b looks at the pair's gamma in block b -- 32 independent loads, ONE
        // memory round trip for the whole utterance -- and the ballot is the list of blocks that have work
        for (int c0 = 0; c0 < T; c0 += 64 * TF) {
            unsigned long long todo;
            {
                const int fb = c0 + lane * TF;
                bool nz = false;
#ifdef BWF_NOSTAGE
                nz = fb < T;
#else
                double ga[TF], gb[TF];
#pragma unroll
                for (int f = 0; f < TF; ++f) {
                    const bool in = fb + f < T;
                    ga[f] = in ? gam[(f0 + fb + f) * gam_stride + gcol_a] : 0.0;
                    gb[f] = (in && gcol_b >= 0) ? gam[(f0 + fb + f) * gam_stride + gcol_b] : 0.0;
                }
#pragma unroll
                for (int f = 0; f < TF; ++f) nz |= (ga[f] > occ_floor) | (ga[f] != ga[f]) | (gb[f] > occ_floor) | (gb[f] != gb[f]);
#endif
#ifdef BWF_NOSKIP
                nz = fb < T;
#endif
                todo = __ballot(nz);     // same test as `wgt` in compute: a block without occupancy adds exact zeros -- not a pruning
            }
            int blk = todo ? __builtin_ctzll(todo) : -1;
            if (blk >= 0) fetch(f0, T, c0 + blk * TF);
            while (blk >= 0) {
                const int bf = c0 + blk * TF;
                park();
                todo &= todo - 1;
                const int nxt = todo ? __builtin_ctzll(todo) : -1;
                if (nxt >= 0) fetch(f0, T, c0 + nxt * TF);      // travels from HBM while this block is computed
                __syncthreads();               // (one wave: orders the LDS stores above before the operand reads below)
                compute((T - bf < TF) ? T - bf : TF);
                __syncthreads();               // (the next block's stores come after these reads)
                blk = nxt;
            }
        }
    }
    // ---- one raw tile per wave: partial[group][pair slot][Zcol = 16 ct + q + 4 reg][comp j] ----
    double* out = partial + (int64_t)blockIdx.x * (NCT * 16 * 16);
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(ct * 16 + q + 4 * r) * 16 + j] = acc[ct][r];
}

// sum a pair's raw tiles over its workgroups (in order: deterministic) -- one thread per tile element, grid (pairs,
// len / 256): with one block per pair every thread walked 6 elements x ~100 workgroups back to back (0.17 ms)
__global__ __launch_bounds__(256) void bw_fused_sum_kernel(const double* __restrict__ partial, const bwf_pair* __restrict__ pairs,
                                                           int waves_per_wg, int len, double* __restrict__ gsum /*[pairs][len]*/) {
    const bwf_pair pr = pairs[blockIdx.x];
    const int i = blockIdx.y * blockDim.x + threadIdx.x;
    if (i >= len) return;
    const double* src = partial + (int64_t)pr.p * len + i;
    double a = 0;
#pragma unroll 8
    for (int w = pr.wg_begin; w < pr.wg_end; ++w) a += src[(int64_t)w * waves_per_wg * len];
    gsum[(int64_t)blockIdx.x * len + i] = a;
}

// re-centre a pair's summed tile on the component means, write [S, M, 1 + 2D]
__global__ __launch_bounds__(256) void bw_fused_reduce_kernel(const double* __restrict__ gsum, const bwf_pair* __restrict__ pairs,
                                                              int lt, int D, int M,
                                                              const double* __restrict__ mean, double* __restrict__ out) {
    const bwf_pair pr = pairs[blockIdx.x];
    const int len = 2 * lt * 16 * 16;
    const double* G = gsum + (int64_t)blockIdx.x * len;   // [2*lt*16][16]: rows d < D linear, row D the occupancy, rows 16 lt + d squared
    const int W = 1 + 2 * D;
    const bool wide = M > 8;
    for (int i = threadIdx.x; i < 16 * (D + 1); i += blockDim.x) {
        const int jj = i / (D + 1), d = i - jj * (D + 1);         // tile column jj = (state slot, component)
        const int s = (wide || jj < 8) ? pr.sa : pr.sb;
        const int m = wide ? pr.m0 + jj : (jj & 7);
        if (s < 0 || m >= M) continue;
        const double s0 = G[D * 16 + jj];
        double* o = out + ((int64_t)s * M + m) * W;
        if (d == D) {
            o[0] = s0;
        } else {
            const double dl = mean[((int64_t)s * M + m) * D + d] - mean[(int64_t)pr.sa * M * D + d];
            const double g1 = G[d * 16 + jj], g2 = G[(16 * lt + d) * 16 + jj];
            o[1 + d] = g1 - dl * s0;
            o[1 + D + d] = g2 - dl * (2.0 * g1 - dl * s0);
        }
    }
}

// compact gamma -> the [N, S] occupancy matrix of the generic statistics kernel (pre-cleared by the caller)
__global__ void expand_gam_kernel(const double* __restrict__ gam, int lanes, const int64_t* __restrict__ utt_off, const int32_t* __restrict__ utt_graph,
                                  const gh_fbchain* __restrict__ chains, int S, double* __restrict__ occ, int32_t* __restrict__ occ_states) {
    const int64_t u = blockIdx.x;
    const gh_fbchain* ch = chains + (utt_graph ? utt_graph[u] : 0);
    const int n = ch->n;
    if (threadIdx.x < GH_FBCHAIN_MAX) occ_states[u * GH_FBCHAIN_MAX + threadIdx.x] = threadIdx.x < n ? ch->state[threadIdx.x] : -1;
    const int64_t f0 = utt_off[u], f1 = utt_off[u + 1];
    for (int64_t i = f0 * lanes + threadIdx.x; i < f1 * lanes; i += blockDim.x) {
        const int64_t f = i / lanes;
        const int jx = (int)(i - f * lanes);
        if (jx < n) occ[f * S + ch->state[jx]] = gam[i];
    }
}

// Sequence-form forward-backward -> occupancy ranges of the fused statistics kernel, on the device (the EM session over
// word strings, gh_em_create_transcripts).  The occupancy matrix is per STATE: a word that stands in several layers of a
// transcript has their occupancies added up in its columns, so the frames of the word's layers must be walked ONCE -- the
// layers of a word are merged wherever their ranges share a 16-frame block (the kernel computes whole blocks); the j-th
// merged range goes to the word's j-th layer slot, the other slots of the word stay empty.  One thread per utterance,
// slot = slot_off[u] + layer; every chain row of the slot gets the layer's range.
// row_lo / row_hi [U, GH_SEQ_MAXK, GH_LAYERS_MAXN] (optional; the lane = cell forward-backward writes them): the range of
// every chain row of a layer instead of the layer's -- a state pair then walks its own ~40 % of the layer's blocks.
__global__ __launch_bounds__(64) void bw_seq_ranges_kernel(const gh_seqgraph* __restrict__ graphs, const int32_t* __restrict__ utt_graph,
                                                           const int64_t* __restrict__ slot_off, const int32_t* __restrict__ seg_lo,
                                                           const int32_t* __restrict__ seg_hi, const int32_t* __restrict__ row_lo,
                                                           const int32_t* __restrict__ row_hi, int64_t U, int n,
                                                           int32_t* __restrict__ rng) {
    constexpr int TF = 16;                // frames per block of bw_fused_kernel
    const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= U) return;
    const gh_seqgraph* g = graphs + (utt_graph ? utt_graph[u] : 0);
    const int K = g->K < GH_SEQ_MAXK ? g->K : GH_SEQ_MAXK;
    int lo[GH_SEQ_MAXK], hi[GH_SEQ_MAXK], olo[GH_SEQ_MAXK], ohi[GH_SEQ_MAXK];
    int8_t into[GH_SEQ_MAXK];             // the slot a layer's frames are walked by (-1: none)
    for (int k = 0; k < K; ++k) {
        lo[k] = seg_lo[u * GH_SEQ_MAXK + k]; hi[k] = seg_hi[u * GH_SEQ_MAXK + k];
        olo[k] = 0x7fffffff; ohi[k] = -1; into[k] = -1;
    }
    unsigned done = 0;
    for (int k = 0; k < K; ++k) {
        if (done >> k & 1) continue;
        const int w = g->word[k];
        int idx[GH_SEQ_MAXK], ord[GH_SEQ_MAXK], ni = 0, no = 0;
        for (int k2 = k; k2 < K; ++k2)
            if (g->word[k2] == w) {
                done |= 1u << k2;
                idx[ni++] = k2;
                if (hi[k2] >= lo[k2]) {          // insertion by first frame
                    int p = no++;
                    while (p > 0 && lo[ord[p - 1]] > lo[k2]) { ord[p] = ord[p - 1]; --p; }
                    ord[p] = k2;
                }
            }
        int out = 0, clo = 0, chi = -1;
        for (int i = 0; i < no; ++i) {
            const int a = lo[ord[i]], b = hi[ord[i]];
            if (chi >= clo && a / TF <= chi / TF) { chi = b > chi ? b : chi; into[ord[i]] = (int8_t)idx[out]; continue; }
            if (chi >= clo) { olo[idx[out]] = clo; ohi[idx[out]] = chi; ++out; }
            clo = a; chi = b;
            into[ord[i]] = (int8_t)idx[out];
        }
        if (chi >= clo) { olo[idx[out]] = clo; ohi[idx[out]] = chi; }
    }
    for (int k = 0; k < K; ++k) {
        int32_t* r = rng + (slot_off[u] + k) * GH_FBCHAIN_MAX * 2;
        if (!row_lo) {
            for (int j = 0; j < n; ++j) { r[2 * j] = olo[k]; r[2 * j + 1] = ohi[k]; }
            continue;
        }
        // per chain row: the hull of the row's ranges over the layers this slot walks (all inside [olo, ohi], whose blocks
        // no other slot of the word touches)
        for (int j = 0; j < n; ++j) {
            int a = 0x7fffffff, b = -1;
            for (int k2 = 0; k2 < K; ++k2)
                if (into[k2] == k) {
                    const int l2 = row_lo[(u * GH_SEQ_MAXK + k2) * GH_LAYERS_MAXN + j], h2 = row_hi[(u * GH_SEQ_MAXK + k2) * GH_LAYERS_MAXN + j];
                    if (h2 >= l2) { a = l2 < a ? l2 : a; b = h2 > b ? h2 : b; }
                }
            r[2 * j] = a; r[2 * j + 1] = b;
        }
    }
}

}  // namespace

int gh_bwf_seq_ranges_launch(gh_ctx* ctx, const gh_seqgraph* graphs, const int32_t* utt_graph, const int64_t* slot_off,
                             const int32_t* seg_lo, const int32_t* seg_hi, const int32_t* row_lo, const int32_t* row_hi,
                             int64_t U, int n, int32_t* rng) {
    if (U <= 0) return GH_OK;
    hipLaunchKernelGGL(bw_seq_ranges_kernel, dim3((unsigned)((U + 63) / 64)), dim3(64), 0, ctx->stream, graphs, utt_graph, slot_off,
                       seg_lo, seg_hi, row_lo, row_hi, U, n, rng);
    GH_HIP(hipGetLastError());
    return GH_OK;
}

// shapes the fused kernel does not cover: the compact gamma becomes the full occupancy matrix for the generic kernel
int gh_bw_expand_gamma(gh_ctx* ctx, gh_batch* b, int S) {
    hipStream_t st = ctx->stream;
    if (b->occ && b->occ_S != S) { GH_HIP(hipStreamSynchronize(st)); GH_HIP(hipFree(b->occ)); b->occ = nullptr; }
    if (!b->occ) { GH_HIP(hipMalloc((void**)&b->occ, (size_t)b->N * S * 8)); b->occ_S = S; }
    if (!b->d_occ_states) GH_HIP(hipMalloc((void**)&b->d_occ_states, (size_t)b->U * GH_FBCHAIN_MAX * 4));
    GH_HIP(hipMemsetAsync(b->occ, 0, (size_t)b->N * S * 8, st));
    void* base;
    const size_t L = b->gam_chains.size();
    int rc = gh_scratch(ctx, L * sizeof(gh_fbchain) + 256 + (size_t)b->U * 4, &base);
    if (rc) return rc;
    gh_fbchain* d_chains = (gh_fbchain*)base;
    int32_t* d_ug = (int32_t*)((char*)base + ((L * sizeof(gh_fbchain) + 255) & ~size_t(255)));
    GH_HIP(hipMemcpyAsync(d_chains, b->gam_chains.data(), L * sizeof(gh_fbchain), hipMemcpyHostToDevice, st));
    if (!b->gam_utt_graph.empty()) GH_HIP(hipMemcpyAsync(d_ug, b->gam_utt_graph.data(), (size_t)b->U * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(expand_gam_kernel, dim3((unsigned)b->U), dim3(256), 0, st, b->gam, b->gam_lanes, b->d_offsets,
                       b->gam_utt_graph.empty() ? nullptr : d_ug, d_chains, S, b->occ, b->d_occ_states);
    GH_HIP(hipGetLastError());
    GH_HIP(hipStreamSynchronize(st));
    b->occ_valid = true;
    return GH_OK;
}

// returns 1 when the fused path does not cover the shapes (the caller uses the generic kernels), < 0 on error.
// Two sources of gamma: the compact [N, 8] matrix of a chain-form forward-backward (one "segment" per utterance), or --
// seq = true -- the occupancy matrix of a sequence-form one, walked as (utterance, layer) segments grouped by word.
int gh_bw_accumulate_fused(gh_ctx* ctx, const gh_gmm* g, const gh_batch* b, double occ_floor, double* stats_dev, double** d_result,
                           bool seq) {
    const int S = g->S, M = g->M, D = g->D, KS = g->KP / 2;
    const int lt = (D + 1 + 15) / 16;     // 16-column tiles of each half of Z = [x - c, 1 | (x - c)^2]
    const std::vector<gh_fbchain>& chains = seq ? b->seq_word_chains : b->gam_chains;
    if (seq ? !(b->seq_seg_valid && b->occ && b->occ_valid && b->occ_S == S) : !b->gam) return 1;
    if (chains.empty() || M > 64 || lt > 3) return 1;
    if (KS != 2 && KS != 4 && KS != 8 && KS != 12 && KS != 20) return 1;
    // M > 8: the mixture normalisers are the batch's likelihoods -- they must be this model's (else: generic kernel)
    const bool wide = M > 8;
    const bool own_nll = b->nll && b->nll_S == S && b->dtype == GH_F64 && b->nll_serial == g->serial && g->serial != 0;
    if (wide && !own_nll) return 1;
    const int L = (int)chains.size();
    // a state may only sit in one place of one graph (the re-centring kernel writes every state once)
    std::vector<int> owner(S, -1);
    int max_pairs = 1;
    for (int l = 0; l < L; ++l) {
        const gh_fbchain& fc = chains[l];
        if (fc.n > GH_FBCHAIN_MAX) return 1;
        for (int jx = 0; jx < fc.n; ++jx) {
            if (fc.state[jx] < 0 || fc.state[jx] >= S || owner[fc.state[jx]] >= 0) return 1;
            owner[fc.state[jx]] = l;
        }
        max_pairs = std::max(max_pairs, (fc.n + 1) / 2);
    }
    if (!wide && max_pairs > 8) return 1;
    // segments: (first frame, length, graph); grouped by graph, longest first inside a graph
    const int64_t U = b->U;
    std::vector<int64_t> seg_first;
    std::vector<int32_t> seg_len;
    std::vector<std::vector<int32_t>> by_graph(L);
    if (!seq) {
        seg_first.resize(U); seg_len.resize(U);
        for (int64_t u = 0; u < U; ++u) { seg_first[u] = b->offsets[u]; seg_len[u] = (int32_t)(b->offsets[u + 1] - b->offsets[u]); }
        for (int64_t k = 0; k < U; ++k) {
            const int64_t u = b->perm[k];
            if (seg_len[u] > 0) by_graph[b->gam_utt_graph.empty() ? 0 : b->gam_utt_graph[u]].push_back((int32_t)u);
        }
    } else {
        // the occupancy matrix is per STATE: a word that stands in several layers of a transcript has their occupancies
        // added up in its columns, so its segments are the UNION of its layers' frame ranges (every frame once)
        for (int64_t u = 0; u < U; ++u) {
            struct iv { int w, lo, hi; };
            iv v[GH_SEQ_MAXK];
            int nv = 0;
            for (int k = 0; k < b->seq_utt_K[u] && k < GH_SEQ_MAXK; ++k) {
                const int lo = b->seq_seg_lo[(size_t)u * GH_SEQ_MAXK + k], hi = b->seq_seg_hi[(size_t)u * GH_SEQ_MAXK + k];
                const int w = b->seq_utt_word[(size_t)u * GH_SEQ_MAXK + k];
                if (hi < lo || w < 0 || w >= L) continue;
                v[nv++] = iv{w, lo, hi};
            }
            std::sort(v, v + nv, [](const iv& x, const iv& y) { return x.w != y.w ? x.w < y.w : x.lo < y.lo; });
            for (int i = 0; i < nv;) {
                int hi = v[i].hi, j = i + 1;
                while (j < nv && v[j].w == v[i].w && v[j].lo <= hi + 1) { hi = std::max(hi, v[j].hi); ++j; }
                by_graph[v[i].w].push_back((int32_t)seg_first.size());
                seg_first.push_back(b->offsets[u] + v[i].lo);
                seg_len.push_back(hi - v[i].lo + 1);
                i = j;
            }
        }
        // longest first inside a graph: a counting sort over the lengths (stable; a comparison sort of ~10^5 segments
        // cost more than the host's share of the whole call)
        int max_len = 0;
        for (int32_t l : seg_len) max_len = std::max(max_len, l);
        std::vector<int32_t> cnt((size_t)max_len + 2), sorted;
        for (auto& v : by_graph) {
            std::fill(cnt.begin(), cnt.end(), 0);
            for (int32_t i : v) cnt[max_len - seg_len[i] + 1]++;
            for (int k = 0; k <= max_len; ++k) cnt[k + 1] += cnt[k];
            sorted.resize(v.size());
            for (int32_t i : v) sorted[cnt[max_len - seg_len[i]]++] = i;
            v.swap(sorted);
        }
    }
    gh_bwf_plan plan;
    int rc = gh_bwf_plan_build(ctx, S, M, D, g->KP, chains, seg_first, seg_len, by_graph, /*persistent=*/false, &plan);
    if (rc) return rc;
    double* d_out = stats_dev ? stats_dev : plan.d_own;
    *d_result = d_out;
    rc = gh_bwf_launch(ctx, plan, g, (const double*)b->feats, seq ? b->occ : b->gam, seq ? S : b->gam_lanes, seq ? 1 : 0, occ_floor, nullptr,
                       d_out, own_nll ? (const double*)b->nll : nullptr, S);
    if (rc) return rc;
    GH_HIP(hipStreamSynchronize(ctx->stream));   // the plan lives in the context's scratch: the next call may overwrite it
    return GH_OK;
}

// Work lists of the fused statistics kernel -> device.  persistent: own allocation (a trainer builds them once: the
// utterance -> word grouping never changes between EM iterations); else carved from the context's scratch.
int gh_bwf_plan_build(gh_ctx* ctx, int S, int M, int D, int KP, const std::vector<gh_fbchain>& chains,
                      const std::vector<int64_t>& seg_first, const std::vector<int32_t>& seg_len,
                      const std::vector<std::vector<int32_t>>& by_graph, bool persistent, gh_bwf_plan* out) {
    memset(out, 0, sizeof *out);
    const int KS = KP / 2;
    const int lt = (D + 1 + 15) / 16;
    const int L = (int)chains.size();
    if (chains.empty() || M > 64 || lt > 3) return 1;
    if (KS != 2 && KS != 4 && KS != 8 && KS != 12 && KS != 20) return 1;
    const bool wide = M > 8;
    const int chunks = wide ? (M + 15) / 16 : 1;
    int max_groups = 1;                    // column groups (= waves of a workgroup) of the widest word
    for (const gh_fbchain& fc : chains) {
        if (fc.n > GH_FBCHAIN_MAX) return 1;
        max_groups = std::max(max_groups, wide ? fc.n * chunks : (fc.n + 1) / 2);
    }
    int slot_shift = 2;
    while ((1 << slot_shift) < max_groups) ++slot_shift;
    if (slot_shift > 6) return 1;
    const int slots = 1 << slot_shift;
    const int64_t n_seg = (int64_t)seg_first.size();
    // workgroups of 3 waves (5 states = 3 pairs): FOUR per CU put exactly three waves on every SIMD (with three per CU
    // one SIMD carries three waves, the others two: the kernel ran at the pace of the fullest one)
    // utterance groups: (groups of a word) x (its state pairs) waves in all; 8 waves per CU are resident at this kernel's
    // register count, so ~8 n_cu / 3 groups for 5-state words put every wave on the chip at once (GMMHMM_BWF_WGS: groups
    // per CU, default 2: measured 1.44 / 1.09 / 1.13 / 1.12 ms per EM iteration for 1 / 2 / 3 / 4)
    // (words of many column groups -- 32 for configs[3] -- fill the chip with fewer utterance groups)
    const int target_wgs = std::max(1, (getenv("GMMHMM_BWF_WGS") ? atoi(getenv("GMMHMM_BWF_WGS")) : 2) * ctx->n_cu * 3 / std::max(3, max_groups));
    const int per_wg = (int)std::max<int64_t>(4, (n_seg + target_wgs - 1) / target_wgs);
    std::vector<int32_t> ulist;
    std::vector<bwf_wg> wgs;
    std::vector<bwf_pair> pairs;
    for (int l = 0; l < L; ++l) {
        const int wg_begin = (int)wgs.size();
        const auto& v = by_graph[l];
        // deal the (length sorted) segments round-robin so that every workgroup of the graph gets the same mix
        const int nw = (int)((v.size() + per_wg - 1) / per_wg);
        for (int w = 0; w < nw; ++w) {
            bwf_wg x{l, (int32_t)ulist.size(), 0, 0};
            for (size_t i = w; i < v.size(); i += nw) ulist.push_back(v[i]);
            x.u_end = (int32_t)ulist.size();
            wgs.push_back(x);
        }
        const gh_fbchain& fc = chains[l];
        if (wide) {
            for (int r = 0; r < fc.n; ++r)
                for (int c = 0; c < chunks; ++c)
                    pairs.push_back(bwf_pair{fc.state[r], -1, wg_begin, (int32_t)wgs.size(), r * chunks + c, 16 * c});
        } else {
            for (int p = 0; 2 * p < fc.n; ++p)
                pairs.push_back(bwf_pair{fc.state[2 * p], 2 * p + 1 < fc.n ? fc.state[2 * p + 1] : -1, wg_begin, (int32_t)wgs.size(), p, 0});
        }
    }
    const int W = 1 + 2 * D;
    const int tile_len = 2 * lt * 16 * 16;
    out->KS = KS; out->lt = lt; out->S = S; out->M = M; out->D = D; out->L = L;
    out->n_wgs = (int)wgs.size(); out->n_pairs = (int)pairs.size(); out->tile_len = tile_len;
    out->max_pairs = max_groups; out->slot_shift = slot_shift;
    bwf_wg* d_wgs; bwf_pair* d_pairs;
    // layout: [own result | lists (one upload) | partial tiles | pair sums]
    UploadLayout lay;
    lay.add((void**)&out->d_own, (size_t)S * M * W * 8, nullptr);
    lay.add((void**)&out->d_ulist, std::max<size_t>(1, ulist.size()) * 4, ulist.data(), ulist.size() * 4);
    lay.add((void**)&d_wgs, std::max<size_t>(1, wgs.size()) * sizeof(bwf_wg), wgs.data(), wgs.size() * sizeof(bwf_wg));
    lay.add((void**)&d_pairs, std::max<size_t>(1, pairs.size()) * sizeof(bwf_pair), pairs.data(), pairs.size() * sizeof(bwf_pair));
    lay.add((void**)&out->d_chains, (size_t)L * sizeof(gh_fbchain), chains.data(), (size_t)L * sizeof(gh_fbchain));
    lay.add((void**)&out->d_segfirst, std::max<size_t>(1, seg_first.size()) * 8, seg_first.data(), seg_first.size() * 8);
    lay.add((void**)&out->d_seglen, std::max<size_t>(1, seg_len.size()) * 4, seg_len.data(), seg_len.size() * 4);
    lay.add((void**)&out->d_part, std::max<size_t>(1, wgs.size()) * (size_t)slots * tile_len * 8, nullptr);
    lay.add((void**)&out->d_gsum, std::max<size_t>(1, pairs.size()) * (size_t)tile_len * 8, nullptr);
    void* base = nullptr;
    if (persistent) {
        GH_HIP(hipMalloc(&base, lay.total));
        out->d_arena = base;
    } else {
        int rc = gh_scratch(ctx, lay.total, &base);
        if (rc) return rc;
    }
    int rc = lay.commit(base, ctx->stream, /*sync=*/true);
    if (rc) { if (persistent) { hipFree(base); out->d_arena = nullptr; } return rc; }
    out->d_wgs = d_wgs; out->d_pairs = d_pairs;
    return GH_OK;
}

void gh_bwf_plan_free(gh_bwf_plan* p) {
    if (p && p->d_arena) hipFree(p->d_arena);
    if (p) memset(p, 0, sizeof *p);
}

// enqueue the fused statistics kernel + its two reduction kernels on the context's stream (no host sync).
// d_chains: the chains (n, state[]) on the device, or null for the copy the plan was built with.
int gh_bwf_launch(gh_ctx* ctx, const gh_bwf_plan& pl, const gh_gmm* g, const double* feats, const double* gam, int gam_stride,
                  int gam_by_state, double occ_floor, const gh_fbchain* d_chains, double* d_out, const double* nll, int nll_S,
                  const int32_t* rng) {
    // (rng with gam_by_state: ranges per SEGMENT of the plan -- gh_bwf_seq_ranges_launch, the EM session over word strings)
    hipStream_t st = ctx->stream;
    const int S = pl.S, M = pl.M, D = pl.D, KS = pl.KS, lt = pl.lt;
    const int W = 1 + 2 * D;
    GH_HIP(hipMemsetAsync(d_out, 0, (size_t)S * M * W * 8, st));
    if (pl.n_wgs == 0) return GH_OK;
    const bool wide = M > 8;
    if (wide && !nll) { gh_set_error("gh_bwf_launch: internal: M = %d needs the batch's likelihoods", M); return GH_ERR_INVALID; }
    const bool norm = nll != nullptr;     // (state pairs: the kernel's own log-sum-exp when the caller has no likelihoods to vouch for)
    const size_t lds = ((size_t)16 * ((2 * KS) | 1) + 16 * 2 + (norm ? 32 : 0) + 128) * 8;
    const dim3 grid((unsigned)pl.n_wgs << pl.slot_shift), blk(64);   // one wave per (utterance group, column group slot)
    const gh_fbchain* chains = d_chains ? d_chains : pl.d_chains;
    const bwf_wg* d_wgs = (const bwf_wg*)pl.d_wgs;
    const bwf_pair* d_pairs = (const bwf_pair*)pl.d_pairs;
#define GH_BWF_W(ks, nc, wd, nm)                                                                                         \
    hipLaunchKernelGGL((bw_fused_kernel<ks, nc, wd, nm>), grid, blk, lds, st, feats, D, M, g->dMean, g->dIvar,           \
                       g->dLogc, gam, occ_floor, gam_stride, gam_by_state, pl.d_segfirst, pl.d_seglen, pl.d_ulist, d_wgs, chains, \
                       ctx->d_fp64_tables, pl.d_part, pl.slot_shift, nll, nll_S, rng)
#define GH_BWF(ks, nc) do { if (wide) GH_BWF_W(ks, nc, true, true); else if (norm) GH_BWF_W(ks, nc, false, true); else GH_BWF_W(ks, nc, false, false); } while (0)
#define GH_BWF_N(ks) switch (lt) { case 1: GH_BWF(ks, 1); break; case 2: GH_BWF(ks, 2); break; default: GH_BWF(ks, 3); break; }
    switch (KS) {
        case 2: GH_BWF_N(2) break;
        case 4: GH_BWF_N(4) break;
        case 8: GH_BWF_N(8) break;
        case 12: GH_BWF_N(12) break;
        default: GH_BWF_N(20) break;
    }
#undef GH_BWF_N
#undef GH_BWF
#undef GH_BWF_W
    GH_HIP(hipGetLastError());
    hipLaunchKernelGGL(bw_fused_sum_kernel, dim3((unsigned)pl.n_pairs, (unsigned)((pl.tile_len + 255) / 256)), dim3(256), 0, st, pl.d_part,
                       d_pairs, 1 << pl.slot_shift, pl.tile_len, pl.d_gsum);
    hipLaunchKernelGGL(bw_fused_reduce_kernel, dim3((unsigned)pl.n_pairs), dim3(256), 0, st, pl.d_gsum, d_pairs, lt, D, M, g->dMean, d_out);
    GH_HIP(hipGetLastError());
    return GH_OK;
}
