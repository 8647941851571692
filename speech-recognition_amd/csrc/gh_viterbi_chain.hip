// Viterbi for left-to-right chains (isolated word models, stacked side by side): every arc of
// row r comes from r, r-1 or r-2, no non-emitting rows.  Same semantics as the generic kernel
// (reference: decode_hmm_states, sr/recognition/decode.py:80-146, as HMM.evaluate uses it,
// hmm.py:126-135): candidates in ascending origin order (r-2, r-1, r) with a strict '<', start
// cells only in column 0, +inf elsewhere in column 0.
//
// gfx950 mapping: ONE WAVE per (utterance, group of <= 64 consecutive rows holding whole chains).
// The cost column lives in one VGPR pair per lane; the neighbour's previous cost arrives by a
// DPP wave shift (no LDS, no barrier); the lane's emission stream nll[t, state(row)] is a
// strided walk through the resident [N,S] matrix, prefetched PF columns ahead in a register
// ring.  A column is ~15 VALU instructions, so the kernel runs at the rate HBM delivers the
// likelihood matrix.  Back-pointers are 1 byte per cell (which of the three arcs won).
#include "gh_internal.h"
#include "gh_viterbi.h"

namespace {

constexpr int PF = 8;  // emission prefetch depth (columns)

// lane i <- lane i-1 (lane 0 keeps `fill`)
__device__ __forceinline__ double wave_shr1(double v, double fill) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), 0x138, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

template <typename ET, bool WANT_BP, bool WANT_COSTS, bool SKIP>
__global__ __launch_bounds__(64) void viterbi_chain_kernel(gh_chain_args a) {
    const int lane = threadIdx.x;
    const int64_t slot = a.slot0 + blockIdx.x / a.n_groups;
    const int g = blockIdx.x % a.n_groups;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int r0 = a.group_row0[g];
    const int nrows = a.group_row0[g + 1] - r0;
    const int r = r0 + lane;
    const bool act = lane < nrows;
    const int64_t f0 = a.utt_off[u];
    const int T = (int)(a.utt_off[u + 1] - f0);
    const int S = a.S, R = a.R;
    const double INF = INFINITY;
    if (T <= 0) return;

    // per-lane row constants
    const int rr = act ? r : r0;
    const double c0 = act ? a.cost0[rr] : INF;   // self arc        (+inf = absent)
    const double c1 = act ? a.cost1[rr] : INF;   // arc from r-1
    const double c2 = (SKIP && act) ? a.cost2[rr] : INF;  // arc from r-2
    const uint8_t info = act ? a.row_info[rr] : 0x0F;     // bits 0-1 first arc code (3 = none), bit 2 start row
    const uint8_t first_code = info & 3;
    const bool is_start = (info & 4) != 0;
    const ET* ep = static_cast<const ET*>(a.nll) + f0 * S + (act ? a.row_state[rr] : 0);

    ET ring[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) ring[k] = (k < T) ? ep[(int64_t)k * S] : ET(0);
    ep += (int64_t)PF * S;

    double prev = INF;
    uint8_t* bp = WANT_BP ? a.bp + a.bp_off[slot] + rr : nullptr;
    double* co = WANT_COSTS ? a.costs + a.costs_off[u] + (int64_t)rr * T : nullptr;

    auto column = [&](int t, ET ev) {
        const double e = (double)ev;
        const double p1 = wave_shr1(prev, INF);
        double best = INF;
        uint8_t code = first_code;
        if (SKIP) {
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
    };

    int t = 0;
    for (; t + PF <= T; t += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const ET ev = ring[k];
            ring[k] = (t + k + PF < T) ? ep[(int64_t)k * S] : ET(0);
            column(t + k, ev);
        }
        ep += (int64_t)PF * S;
    }
#pragma unroll
    for (int k = 0; k < PF; ++k)
        if (t + k < T) column(t + k, ring[k]);

    // end costs: every end row knows its slot in the graph's end list
    if (act) {
        const int es = a.end_slot[rr];
        if (es >= 0) a.end_cost[u * a.n_end + es] = prev;
    }
}

// End selection ('>=': the last of equal minima, decode.py:129-134) + back-trace of one utterance
// per wave; lane 0 walks the 1-byte back-pointers, the pairs are parked in LDS and flushed by
// the wave.
__global__ __launch_bounds__(64) void chain_backtrace_kernel(gh_chain_args a, int64_t u_begin) {
    __shared__ int32_t pbuf[2 * 256];
    __shared__ int s_n, s_i, s_j, s_len;
    const int lane = threadIdx.x;
    const int64_t slot = u_begin + blockIdx.x;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int T = (int)(a.utt_off[u + 1] - a.utt_off[u]);
    const int R = a.R;
    if (lane == 0) {
        double best = INFINITY;
        int bi = -1;
        for (int k = 0; k < a.n_end; ++k) {
            const double c = a.end_cost[u * a.n_end + k];
            if (best >= c) { best = c; bi = k; }
        }
        if (T <= 0) bi = -1;
        a.best_end[u] = bi;
        s_i = bi >= 0 ? a.end_rows[bi] : 0;
        s_j = T - 1;
        s_len = 0;
        s_n = 0;
    }
    __syncthreads();
    if (!a.path || T <= 1 || a.best_end[u] < 0) {
        if (lane == 0 && a.path_len) a.path_len[u] = 0;
        return;
    }
    const uint8_t* bp = a.bp + a.bp_off[slot];
    int32_t* path = a.path + 2 * a.path_off[u];
    while (s_j != 0) {
        if (lane == 0) {
            int i = s_i, j = s_j, n = 0;
            while (j != 0 && n < 256) {
                const int code = bp[(int64_t)j * R + i];
                if (code == 3) { atomicOr(a.flag, 2); j = 0; break; }
                i -= code;
                --j;
                pbuf[2 * n] = i;
                pbuf[2 * n + 1] = j;
                ++n;
            }
            s_i = i; s_j = j; s_n = n;
        }
        __syncthreads();
        const int n = s_n, len = s_len;
        for (int k = lane; k < 2 * n; k += 64) path[2 * (int64_t)len + k] = pbuf[k];
        __syncthreads();
        if (lane == 0) s_len = len + n;
        __syncthreads();
    }
    if (lane == 0) a.path_len[u] = s_len;
}

}  // namespace

int gh_launch_viterbi_chain(gh_ctx* ctx, const gh_chain_args& a, int64_t u_begin, int64_t n_utts, bool f64,
                            bool want_bp, bool want_costs, bool skip) {
    if (n_utts <= 0) return GH_OK;
    gh_chain_args b = a;
    dim3 grid((unsigned)(n_utts * a.n_groups)), blk(64);
    b.slot0 = u_begin;  // perm[] and bp_off[] are indexed by absolute launch slot
#define GH_VC(ET, BP, CO, SK) hipLaunchKernelGGL((viterbi_chain_kernel<ET, BP, CO, SK>), grid, blk, 0, ctx->stream, b)
#define GH_VC_S(ET, BP, CO) do { if (skip) GH_VC(ET, BP, CO, true); else GH_VC(ET, BP, CO, false); } while (0)
#define GH_VC_T(ET) do { if (want_costs) GH_VC_S(ET, true, true); else if (want_bp) GH_VC_S(ET, true, false); else GH_VC_S(ET, false, false); } while (0)
    if (f64) GH_VC_T(double); else GH_VC_T(float);
#undef GH_VC_T
#undef GH_VC_S
#undef GH_VC
    GH_HIP(hipGetLastError());
    return GH_OK;
}

// No path wanted: the end selection alone, one LANE per utterance (chain_backtrace_kernel spends a workgroup per utterance on
// it: 45 us per 10 000 utterances of the headline step, all of it workgroup starts)
__global__ void chain_end_select_kernel(gh_chain_args a, int64_t u_begin, int64_t n_utts) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_utts) return;
    const int64_t slot = u_begin + i;
    const int64_t u = a.perm ? a.perm[slot] : slot;
    const int T = (int)(a.utt_off[u + 1] - a.utt_off[u]);
    double best = INFINITY;
    int bi = -1;
    for (int k = 0; k < a.n_end; ++k) {
        const double c = a.end_cost[u * a.n_end + k];
        if (best >= c) { best = c; bi = k; }                  // '>=': the last of equal minima (decode.py:129-134)
    }
    a.best_end[u] = T <= 0 ? -1 : bi;
    if (a.path_len) a.path_len[u] = 0;
}

int gh_launch_chain_backtrace(gh_ctx* ctx, const gh_chain_args& a, int64_t u_begin, int64_t n_utts) {
    if (n_utts <= 0) return GH_OK;
    if (!a.path) {
        hipLaunchKernelGGL(chain_end_select_kernel, dim3((unsigned)((n_utts + 255) / 256)), dim3(256), 0, ctx->stream, a, u_begin, n_utts);
        GH_HIP(hipGetLastError());
        return GH_OK;
    }
    hipLaunchKernelGGL(chain_backtrace_kernel, dim3((unsigned)n_utts), dim3(64), 0, ctx->stream, a, u_begin);
    GH_HIP(hipGetLastError());
    return GH_OK;
}
