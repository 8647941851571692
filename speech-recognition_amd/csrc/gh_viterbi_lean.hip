// Lean Viterbi kernel for the graphs that matter in practice (isolated word chains, K-layer word
// lattices, forced-alignment lattices): <= 3 levels, every level fits one row per lane, S <=
// blockDim, no NaN arc costs, no same-column self arc.  Same semantics as the generic kernel in
// gh_viterbi.hip (reference: decode_hmm_states, sr/recognition/decode.py:80-146); everything a
// lane needs per column is precomputed ONCE into registers so that the per-column work of a
// row is ~20 instructions:
//   * up to three register arcs per row (cost + 32-bit LDS byte addresses for even and odd columns; a
//     missing / dead arc is padded with cost +inf -- it can never win the strict '<');
//   * rows with more arcs (non-emitting rows collecting all word ends) get 16 lanes each: the
//     lanes scan the LDS-resident arc list in parallel and a DPP row reduction (value, then
//     lowest arc index: np.argmin's first minimum) replaces a serial chain of LDS round trips;
//   * the column loop is unrolled by two so the prev/cur swap costs nothing;
//   * emission vectors arrive in chunks of CH columns, prefetched one chunk ahead (registers ->
//     LDS), with a zero slot for non-emitting rows;
//   * back-pointer stores use running per-level pointers; the back-trace is buffered in LDS.
#include "gh_internal.h"
#include "gh_viterbi.h"
#include <type_traits>

namespace {

constexpr uint16_t BP_NONE = 0xFFFFu;
constexpr int NPRE = 8;

__device__ __forceinline__ double& lds_at(char* smem, unsigned off) { return *reinterpret_cast<double*>(smem + off); }

// lane i <- lane i+K of the same 16-lane row (rows do not wrap: lanes past the end keep their value)
template <int K> __device__ __forceinline__ int row_shl(int v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x100 | K, 0xF, 0xF, false);
}
// Workgroup barrier that orders LDS traffic only (__syncthreads() also waits for every outstanding GLOBAL
// access, vmcnt(0) -- the emission prefetch and the back-pointer flush should stay in flight across it).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int K> __device__ __forceinline__ void min_step(double& best, int& idx) {
    const int lo = row_shl<K>(__double2loint(best)), hi = row_shl<K>(__double2hiint(best));
    const int oi = row_shl<K>(idx);
    const double ov = __hiloint2double(hi, lo);
    const bool take = (ov < best) || (ov == best && oi < idx);
    best = take ? ov : best;
    idx = take ? oi : idx;
}

// MAXB = largest block the instantiation is launched with.  The kernel is latency bound per workgroup, so
// what counts is workgroups per CU: the 512 variant is held to 128 VGPRs (4 waves per SIMD, three 320-lane
// workgroups per CU for the K = 7 lattice: 3.4 -> 2.3 ms against the 141 VGPRs the compiler takes otherwise).
template <typename ET, bool WANT_PATH, bool WANT_COSTS, int NL, int MAXB>
__global__ __launch_bounds__(MAXB, (MAXB == 512 ? 4 : 1)) void viterbi_lean_kernel(gh_vit_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int s_bi;
    __shared__ int s_state[4];
    const int tid = threadIdx.x, bd = blockDim.x;
    const int64_t slot = a.u_begin + blockIdx.x;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int l = a.utt_lat ? a.utt_lat[u] : 0;
    const gh_lattices::desc dsc = a.descs[l];
    const int R = dsc.R, n_end = dsc.n_end;
    const int32_t* row_state = a.row_state + dsc.row_base;
    const uint8_t* row_start = a.row_start + dsc.row_base;
    const int32_t* pred_ptr = a.pred_ptr + dsc.ptr_base;
    const uint32_t* pred_row = a.pred_row + dsc.arc_base;
    const double* pred_cost = a.pred_cost + dsc.arc_base;
    const int32_t* order = a.order + dsc.row_base;
    const int32_t* level_ptr = a.level_ptr + dsc.lev_base;
    const int32_t* level_narrow = a.level_narrow + dsc.lev_base;
    const int32_t* end_rows = a.end_rows + dsc.end_base;
    const int64_t f0 = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - f0);
    const int S = a.S, S1 = S + 1;  // slot S of every emission vector is 0.0 (non-emitting rows)
    const ET* nll = static_cast<const ET*>(a.nll) + f0 * S;
    const double INF = INFINITY;
    // LDS map (bytes): [colA r_pad*8][colB r_pad*8][em 2*CH*S1*8][arc cost arc_cap*8][arc row arc_cap*4]
    const unsigned COLB = (unsigned)a.r_pad * 8u;
    const unsigned EM0 = 2u * COLB;
    const int CH = a.em_chunk;
    const unsigned EMCH = (unsigned)(CH * S1) * 8u;
    const unsigned ARC0 = EM0 + 2u * EMCH;
    double* s_cost = reinterpret_cast<double*>(smem + ARC0);
    uint32_t* s_w = reinterpret_cast<uint32_t*>(smem + ARC0 + (unsigned)a.arc_cap * 8u);
    double* costs = WANT_COSTS ? a.costs + a.costs_off[u] : nullptr;  // full matrices: tests / one-utterance API
    uint16_t* bp = WANT_PATH ? a.bp + a.bp_off[slot] : nullptr;

    if (T <= 0) {
        if (tid == 0) {
            if (a.best_end) a.best_end[u] = -1;
            if (a.path_len) a.path_len[u] = 0;
        }
        return;
    }
    for (int r = tid; r < R; r += bd) { lds_at(smem, r * 8u) = INF; lds_at(smem, COLB + r * 8u) = INF; }
    {   // arc list of the graph -> LDS (for the wide rows)
        const int n_arcs = pred_ptr[R];
        for (int p = tid; p < n_arcs && p < a.arc_cap; p += bd) { s_cost[p] = pred_cost[p]; s_w[p] = pred_row[p]; }
    }
    // ---- per-lane row descriptors ------------------------------------------------------------
    int f_r[NL], f_na[NL], f_p0[NL];
    bool f_start[NL];
    unsigned f_em[NL];           // byte offset of the row's emission inside an emission vector
    unsigned f_a0[NL], f_a1[NL], f_a2[NL]; // LDS byte address of arc 0 / 1 / 2 at EVEN columns (prev = colA, cur = colB)
    int f_d0[NL], f_d1[NL], f_d2[NL];      // what to add at odd columns
    double f_c0[NL], f_c1[NL], f_c2[NL];
    int f_b0[NL], f_b1[NL], f_b2[NL];      // back-pointer codes (32-bit: with uint16_t the arrays end up in scratch memory)
    // wide rows (> 2 arcs): lane group g = tid / 16 owns wide row g of the level, lane j = tid % 16 its arcs j, j+16, ..
    int w_r[NL], w_p0[NL], w_na[NL];
    unsigned w_em[NL];
    bool w_start[NL];
#pragma unroll
    for (int lev = 0; lev < NL; ++lev) {
        const int n_narrow = level_narrow[lev];
        const int i = level_ptr[lev] + tid;
        f_r[lev] = -1; f_na[lev] = 0; f_p0[lev] = 0; f_start[lev] = false; f_em[lev] = (unsigned)S * 8u;
        f_a0[lev] = 0; f_a1[lev] = 0; f_a2[lev] = 0; f_d0[lev] = 0; f_d1[lev] = 0; f_d2[lev] = 0;
        f_c0[lev] = INF; f_c1[lev] = INF; f_c2[lev] = INF;
        f_b0[lev] = BP_NONE; f_b1[lev] = BP_NONE; f_b2[lev] = BP_NONE;
        w_r[lev] = -1; w_p0[lev] = 0; w_na[lev] = 0; w_em[lev] = (unsigned)S * 8u; w_start[lev] = false;
        {
            const int iw = level_ptr[lev] + n_narrow + (tid >> 4);
            if (iw < level_ptr[lev + 1]) {
                const int r = order[iw];
                const int st = row_state[r];
                w_r[lev] = r;
                w_p0[lev] = pred_ptr[r];
                w_na[lev] = pred_ptr[r + 1] - pred_ptr[r];
                w_em[lev] = (unsigned)(st >= 0 ? st : S) * 8u;
                w_start[lev] = (row_start[r] & 1) != 0;
            }
        }
        if (tid < n_narrow) {
            const int r = order[i];
            const int st = row_state[r];
            const int p0 = pred_ptr[r], na = pred_ptr[r + 1] - p0;
            f_r[lev] = r;
            f_na[lev] = na;
            f_p0[lev] = p0;
            f_start[lev] = (row_start[r] & 1) != 0;
            f_em[lev] = (unsigned)(st >= 0 ? st : S) * 8u;
            if (na >= 1) {
                const uint32_t w = pred_row[p0];
                const unsigned o = w & GH_ARC_ROW;
                const bool same = (w & GH_ARC_SAME) != 0;
                f_c0[lev] = (w & GH_ARC_DEAD) ? INF : pred_cost[p0];
                f_a0[lev] = (same ? COLB : 0u) + o * 8u;
                f_d0[lev] = same ? -(int)COLB : (int)COLB;
                f_b0[lev] = (int)(o | (same ? 0x8000u : 0u));
            }
            if (na >= 2) {
                const uint32_t w = pred_row[p0 + 1];
                const unsigned o = w & GH_ARC_ROW;
                const bool same = (w & GH_ARC_SAME) != 0;
                f_c1[lev] = (w & GH_ARC_DEAD) ? INF : pred_cost[p0 + 1];
                f_a1[lev] = (same ? COLB : 0u) + o * 8u;
                f_d1[lev] = same ? -(int)COLB : (int)COLB;
                f_b1[lev] = (int)(o | (same ? 0x8000u : 0u));
            }
            if (na >= 3) {
                const uint32_t w = pred_row[p0 + 2];
                const unsigned o = w & GH_ARC_ROW;
                const bool same = (w & GH_ARC_SAME) != 0;
                f_c2[lev] = (w & GH_ARC_DEAD) ? INF : pred_cost[p0 + 2];
                f_a2[lev] = (same ? COLB : 0u) + o * 8u;
                f_d2[lev] = same ? -(int)COLB : (int)COLB;
                f_b2[lev] = (int)(o | (same ? 0x8000u : 0u));
            }
        }
    }

    if (T == 1) {
        // decode.py:113 with c == 0: column c-1 wraps onto column 0 itself: serial ascending sweep
        double* cur = reinterpret_cast<double*>(smem);
        __syncthreads();
        if (tid == 0) {
            for (int r = 0; r < R; ++r) {
                const int st = row_state[r];
                const double e = st >= 0 ? (double)nll[st] : 0.0;
                double c = INF;
                if (row_start[r] & 1) {
                    c = e;
                } else {
                    const int p0 = pred_ptr[r], p1 = pred_ptr[r + 1];
                    if (p0 < p1) {
                        double best = 0;
                        for (int p = p0; p < p1; ++p) {
                            const double v = pred_cost[p] + cur[pred_row[p] & GH_ARC_ROW];
                            if (p == p0 || v < best) best = v;
                        }
                        c = best + e;
                        if (c != c) c = INF;
                    }
                }
                cur[r] = c;
                if (WANT_COSTS) costs[r] = c;
            }
        }
        __syncthreads();
    } else {
        // ---- emission chunks: chunk 0 straight to LDS, then one chunk ahead through registers ----
        ET pre[NPRE];
        const int chunk_elems = CH * S;
        {
            const int n0 = (T < CH ? T : CH) * S;
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                const int i = tid + k * bd;
                if (i < n0) lds_at(smem, EM0 + (unsigned)((i / S) * S1 + i % S) * 8u) = (double)nll[i];
            }
            for (int c = tid; c < 2 * CH; c += bd) lds_at(smem, EM0 + (unsigned)(c * S1 + S) * 8u) = 0.0;
        }
        // back-pointers: 8 columns are collected in LDS ([8][R] uint16) and flushed as one contiguous block of
        // 16-byte stores (8x fewer store instructions than one 2-byte store per row and column)
        uint16_t* bpc = reinterpret_cast<uint16_t*>(smem + a.bpc_off);
        double* co_p[NL];
#pragma unroll
        for (int lev = 0; lev < NL; ++lev) {
            co_p[lev] = WANT_COSTS ? costs + (int64_t)(f_r[lev] >= 0 ? f_r[lev] : 0) * T : nullptr;
        }
        int kc = 0, ci = 0;  // chunk index, column inside the chunk
        // Columns two at a time: the inner loop is fully unrolled, so the column parity PAR (which of the two LDS
        // cost columns is "previous") is a compile-time constant.  (Written as a plain loop on purpose: as a
        // generic lambda called per parity the row-descriptor arrays stayed in scratch memory.)
        for (int t0 = 0; t0 < T; t0 += 2) {
#pragma unroll
            for (int PAR = 0; PAR < 2; ++PAR) {
                const int t = t0 + PAR;
                if (t >= T) break;
                if (ci == 0) {
                    const int64_t base = (int64_t)(kc + 1) * chunk_elems, lim = (int64_t)T * S;
    #pragma unroll
                    for (int k = 0; k < NPRE; ++k) {
                        const int i = tid + k * bd;
                        pre[k] = (i < chunk_elems && base + i < lim) ? nll[base + i] : ET(0);
                    }
                }
                lds_barrier();
                const unsigned emv = EM0 + (unsigned)(kc & 1) * EMCH + (unsigned)(ci * S1) * 8u;
    #pragma unroll
                for (int lev = 0; lev < NL; ++lev) {
                    if (f_r[lev] >= 0) {
                        const double v0 = f_c0[lev] + lds_at(smem, f_a0[lev] + (PAR ? f_d0[lev] : 0));
                        const double v1 = f_c1[lev] + lds_at(smem, f_a1[lev] + (PAR ? f_d1[lev] : 0));
                        const double v2 = f_c2[lev] + lds_at(smem, f_a2[lev] + (PAR ? f_d2[lev] : 0));
                        const bool pick1 = v1 < v0;             // strict '<' in ascending origin order: np.argmin's first minimum
                        double best = pick1 ? v1 : v0;
                        int b = pick1 ? f_b1[lev] : f_b0[lev];
                        const bool pick2 = v2 < best;
                        best = pick2 ? v2 : best;
                        b = pick2 ? f_b2[lev] : b;
                        const double e = lds_at(smem, emv + f_em[lev]);
                        double c = best + e;
                        c = (c != c) ? INF : c;                 // min(inf, nan) keeps inf (decode.py:124)
                        if (f_na[lev] == 0) { c = INF; b = BP_NONE; }      // rows without arcs stay +inf (:116-117)
                        if (t == 0 && f_start[lev]) { c = e; b = BP_NONE; }  // decode.py:99-101
                        lds_at(smem, (PAR ? 0u : COLB) + (unsigned)f_r[lev] * 8u) = c;
                        if (WANT_PATH) bpc[(t & 7) * R + f_r[lev]] = (uint16_t)b;
                        if (WANT_COSTS) { *co_p[lev] = c; co_p[lev] += 1; }
                    }
                    if (w_r[lev] >= 0) {  // wide row: 16 lanes scan its arcs, then reduce (value, lowest arc index)
                        const unsigned prevb = PAR ? COLB : 0u, curb = PAR ? 0u : COLB;
                        double best = INF;
                        int bidx = 0x7fffffff;
                        for (int p = w_p0[lev] + (tid & 15); p < w_p0[lev] + w_na[lev]; p += 16) {
                            const uint32_t w = s_w[p];
                            const unsigned o = w & GH_ARC_ROW;
                            const double v = (w & GH_ARC_DEAD) ? INF : s_cost[p] + lds_at(smem, ((w & GH_ARC_SAME) ? curb : prevb) + o * 8u);
                            if (v < best || bidx == 0x7fffffff) { best = v; bidx = p; }
                        }
                        min_step<8>(best, bidx);
                        min_step<4>(best, bidx);
                        min_step<2>(best, bidx);
                        min_step<1>(best, bidx);
                        if ((tid & 15) == 0) {
                            const uint32_t w = s_w[bidx];
                            uint16_t b = (uint16_t)((w & GH_ARC_ROW) | ((w & GH_ARC_SAME) ? 0x8000u : 0u));
                            const double e = lds_at(smem, emv + w_em[lev]);
                            double c = best + e;
                            c = (c != c) ? INF : c;
                            if (t == 0 && w_start[lev]) { c = e; b = BP_NONE; }
                            lds_at(smem, curb + (unsigned)w_r[lev] * 8u) = c;
                            if (WANT_PATH) bpc[(t & 7) * R + w_r[lev]] = b;
                            if (WANT_COSTS) costs[(int64_t)w_r[lev] * T + t] = c;
                        }
                    }
                    if (lev + 1 < NL) lds_barrier();
                }
                if (WANT_PATH && ((t & 7) == 7 || t == T - 1)) {  // flush the staged back-pointer columns
                    lds_barrier();
                    const int c0 = t & ~7;
                    const int nvec = ((t - c0 + 1) * R * 2 + 15) >> 4;
                    // 16-byte aligned: block offsets are multiples of 8 entries, c0 is a multiple of 8 columns
                    uint4* dst = reinterpret_cast<uint4*>(bp + (int64_t)c0 * R);
                    const uint4* srcv = reinterpret_cast<const uint4*>(bpc);
                    for (int k = tid; k < nvec; k += bd) dst[k] = srcv[k];
                    lds_barrier();
                }
                if (ci == CH - 1) {  // park the prefetched chunk in the other buffer
                    const unsigned dst = EM0 + (unsigned)((kc + 1) & 1) * EMCH;
    #pragma unroll
                    for (int k = 0; k < NPRE; ++k) {
                        const int i = tid + k * bd;
                        if (i < chunk_elems) lds_at(smem, dst + (unsigned)((i / S) * S1 + i % S) * 8u) = (double)pre[k];
                    }
                    ci = 0;
                    ++kc;
                } else {
                    ++ci;
                }
            }
        }
        __syncthreads();
    }
    // the last column is in colB if T is odd (even-parity column written last), else colA
    const double* last = reinterpret_cast<const double*>(smem + ((T & 1) ? (T == 1 ? 0u : COLB) : 0u));
    if (tid == 0) {
        double best = INF;
        int bi = -1;
        double* ec = a.end_cost ? a.end_cost + (a.end_off ? a.end_off[u] : u * n_end) : nullptr;
        for (int k = 0; k < n_end; ++k) {
            const double c = last[end_rows[k]];
            if (ec) ec[k] = c;
            if (best >= c) { best = c; bi = k; }  // '>=': last minimum wins (decode.py:131)
        }
        if (a.best_end) a.best_end[u] = bi;
        s_bi = bi;
    }
    __syncthreads();
    if (WANT_PATH) {
        // Back-trace (decode.py:143-145).  A serial walk over back-pointers in global memory costs one memory
        // round trip per cell (~0.7 ms for a 317-frame utterance); instead ALL lanes stage the next block of
        // back-pointer columns into LDS (coalesced) and lane 0 walks them there; the cells found are
        // buffered in LDS and flushed by all lanes.  LDS map: [pbuf 256 cells][state][bp columns ...].
        constexpr int PB = 255;
        int32_t* pbuf = reinterpret_cast<int32_t*>(smem);
        const int NB = (a.lds_bytes - 2048 - 16 - 32) / (2 * R);  // back-pointer columns per block (host: >= 8)
        int32_t* path = a.path + 2 * a.path_off[u];
        const int64_t cap = a.path_off[u + 1] - a.path_off[u];
        if (tid == 0) {
            const int bi = s_bi;
            s_state[0] = (bi >= 0) ? end_rows[bi] : 0;
            s_state[1] = T - 1;
            s_state[2] = 0;
            s_state[3] = !(T > 1 && bi >= 0);
#ifdef GH_LEAN_NOBT  // diagnostic build: forward sweep (with back-pointer stores) only
            s_state[3] = 1;
#endif
        }
        __syncthreads();
        while (!s_state[3]) {
            const int j_hi = s_state[1];                          // columns (j_lo, j_hi] are staged; column 0 is never read
            const int j_lo = (j_hi - NB > 0) ? j_hi - NB : 0;
            // 16-byte loads from the enclosing aligned window, four per lane in flight (a 2-byte load -> wait ->
            // LDS store loop costs one memory round trip per 64 back-pointers)
            const uint16_t* src = bp + (int64_t)(j_lo + 1) * R;
            const unsigned mis = (unsigned)(reinterpret_cast<uintptr_t>(src) & 15u);
            const uint4* vsrc = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(src) - mis);
            uint4* vdst = reinterpret_cast<uint4*>(smem + 2048 + 16);
            const int nvec = (int)(((unsigned)((j_hi - j_lo) * R) * 2u + mis + 15u) >> 4);
            for (int k0 = 0; k0 < nvec; k0 += 4 * bd) {
                uint4 q[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int k = k0 + e * bd + tid; q[e] = (k < nvec) ? vsrc[k] : uint4{0, 0, 0, 0}; }
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int k = k0 + e * bd + tid; if (k < nvec) vdst[k] = q[e]; }
            }
            const uint16_t* cbp = reinterpret_cast<const uint16_t*>(smem + 2048 + 16 + mis);
            __syncthreads();
            if (tid == 0) {
                int i = s_state[0], j = j_hi, n_new = 0;
                const int len = s_state[2];
                while (j > j_lo && n_new < PB) {
                    const uint16_t bq = cbp[(j - j_lo - 1) * R + i];
                    if (bq == BP_NONE) { atomicOr(a.flag, 2); j = 0; break; }
                    if (len + n_new >= cap) { atomicOr(a.flag, 4); j = 0; break; }
                    i = bq & 0x7FFF;
                    if (!(bq & 0x8000u)) --j;
                    pbuf[2 * n_new] = i;
                    pbuf[2 * n_new + 1] = j;
                    ++n_new;
                }
                s_state[0] = i; s_state[1] = j;
                s_state[3] = (j == 0);
                pbuf[2 * PB] = n_new;
            }
            __syncthreads();
            const int n_new = pbuf[2 * PB];
            const int len = s_state[2];
            for (int k = tid; k < 2 * n_new; k += bd) path[2 * (int64_t)len + k] = pbuf[k];
            __syncthreads();
            if (tid == 0) s_state[2] = len + n_new;
            __syncthreads();
        }
        if (tid == 0) a.path_len[u] = s_state[2];
    }
}

}  // namespace

int gh_launch_viterbi_lean(gh_ctx* ctx, const gh_vit_args& a, int64_t n_utts, int block, size_t lds_bytes,
                           bool f64, bool want_path, int levels) {
    if (n_utts <= 0) return GH_OK;
    dim3 grid((unsigned)n_utts), blk((unsigned)block);
    const bool wc = a.costs != nullptr;
#define GH_VL_B(ET, WP, WC, NLV, MB) hipLaunchKernelGGL((viterbi_lean_kernel<ET, WP, WC, NLV, MB>), grid, blk, lds_bytes, ctx->stream, a)
#define GH_VL(ET, WP, WC, NLV)                               \
    do {                                                     \
        if (block <= 256) GH_VL_B(ET, WP, WC, NLV, 256);     \
        else if (block <= 512) GH_VL_B(ET, WP, WC, NLV, 512); \
        else GH_VL_B(ET, WP, WC, NLV, 1024);                 \
    } while (0)
#define GH_VL_N(ET, WP, WC)                      \
    switch (levels) {                            \
        case 1: GH_VL(ET, WP, WC, 1); break;     \
        case 2: GH_VL(ET, WP, WC, 2); break;     \
        default: GH_VL(ET, WP, WC, 3); break;    \
    }
    // (full cost matrices are a test / single-utterance feature: only the path variant carries them)
    if (f64) {
        if (wc) { GH_VL_N(double, true, true) } else if (want_path) { GH_VL_N(double, true, false) } else { GH_VL_N(double, false, false) }
    } else {
        if (wc) { GH_VL_N(float, true, true) } else if (want_path) { GH_VL_N(float, true, false) } else { GH_VL_N(float, false, false) }
    }
#undef GH_VL_N
#undef GH_VL
#undef GH_VL_B
    GH_HIP(hipGetLastError());
    return GH_OK;
}
