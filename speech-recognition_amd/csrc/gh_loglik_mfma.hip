// Batched GMM negative log-likelihood on the CDNA4 matrix cores (reference: GMM.evaluate,
// sr/recognition/hmm_state.py:114-120, log domain).
//
// The frame x component contraction is a true dense GEMM:
//     ll[g, n] = C[g] + sum_k P[g,k] Z[n,k],   Z[n] = [x_n^2 | x_n]  (K = 2*KP),
//     P[g]  = [-0.5/var | mean/var],  C[g] = log w - 0.5(D log 2pi + sum log var + sum mean^2/var)
// so it runs on v_mfma_f64_16x16x4_f64 (fp64, the reference's arithmetic) or
// v_mfma_f32_16x16x4_f32 (exact fp32 fma chain); the log-sum-exp over the mixture
// components is the epilogue.
//
// gfx950 mapping (one wave per workgroup, no inter-wave traffic):
//   * a wave owns 32 frames.  Their Z operand (B fragments of both 16-frame column tiles,
//     all K) is loaded ONCE -- coalesced global -> LDS tile -> registers -- and stays in
//     VGPRs for the whole kernel;
//   * the Gaussians stream past as 16-row tiles.  The host packs P so that every A
//     fragment is 64 consecutive elements in lane order (one coalesced 512-byte /
//     256-byte load per k-step); all waves read the same 256 KB, which lives in L2/L1.
//     The next tile's fragments are fetched into a second register set while the
//     current tile's MFMAs issue;
//   * accumulators start at C[g], so they finish as component log-densities.  The row
//     order inside a tile is chosen per dtype (host side) such that lane group q = lane>>4
//     holds components 4q..4q+3 of the tile in its 4 accumulator registers -- for M = 8
//     the log-sum-exp is 4 in-register terms + one xor-16 exchange;
//   * results are staged in LDS as a [32 frames, states] tile and written back as one
//     contiguous block (the [N,S] matrix is row-major), 16 bytes per lane.
#include "gh_internal.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <typename T> struct Acc;
template <> struct Acc<double> {
    typedef v4d type;
    static __device__ __forceinline__ v4d mfma(double a, double b, v4d c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
};
template <> struct Acc<float> {
    typedef v4f type;
    static __device__ __forceinline__ v4f mfma(float a, float b, v4f c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
};

// ---- fp64 exp / log for the log-sum-exp epilogue -------------------------------------
// On gfx950 fp64 MFMA and fp64 VALU share the same FMA pipeline (measured with
// tools/mfma_coissue.hip: their times ADD), so every fp64 VALU instruction of the epilogue
// is paid in full next to the MFMAs.  Both functions are therefore table driven (128-entry
// tables in LDS, filled once per workgroup) with short polynomials, branch-free:
//   exp: x = n ln2/128 + r, |r| <= ln2/256;  2^(n/128) = 2^(n>>7) * T[n&127];
//        exp(r) - 1 by a degree-5 polynomial (|error| < 6e-19)           ~11 fp64 ops
//   log: s = m 2^e, m in [.5,1); m * INV[j] = 1 + rho, |rho| <= 2^-8, j = top 7 mantissa
//        bits; log s = e ln2 + L[j] + log1p(rho), degree-6 polynomial      ~12 fp64 ops
// Accuracy ~1e-16 absolute on exp (arguments are <= 0) and on log of sums in [1, M].
struct Fp64Tables {
    double exp2[128];   // 2^(j/128)
    double inv[128];    // 1 / centre of mantissa bin j, centre = 0.5 + (j + 0.5)/256
    double nlog[128];   // -log(inv[j])
};

__device__ __forceinline__ double exp_nonpos(double x, const double* __restrict__ tab) {
    x = fmax(x, -745.0);  // one v_max_f64; NaN inputs are caught by the caller's poison check instead
    const double n = __builtin_rint(x * 184.66496523378731);         // 128 / ln 2
    double r = fma(n, -0.00541521234663378, x);                      // ln2/128, high part (32 bits)
    r = fma(n, -1.4907929134926466e-12, r);                          // low part
    const int ni = (int)n;
    const double t = tab[ni & 127];
    double p = fma(r, 8.3333333333333332177e-03, 4.1666666666666664354e-02);  // 1/120, 1/24
    p = fma(p, r, 1.6666666666666665741e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = p * r;                                                        // exp(r) - 1
    return __builtin_ldexp(fma(t, p, t), ni >> 7);
}

__device__ __forceinline__ double log_pos(double s, const double* __restrict__ inv,
                                          const double* __restrict__ nlog) {
    const double m = __builtin_amdgcn_frexp_mant(s);   // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_exp(s);
    const int j = (__double2hiint(m) >> 13) & 127;
    const double rho = fma(m, inv[j], -1.0);
    double p = fma(rho, -1.6666666666666665741e-01, 0.2);
    p = fma(p, rho, -0.25);
    p = fma(p, rho, 3.3333333333333331483e-01);
    p = fma(p, rho, -0.5);
    p = fma(p, rho, 1.0);
    const double r = fma((double)e, 6.93147180559945286e-01, fma(p, rho, nlog[j]));
    return (s == 0.0) ? -INFINITY : r;  // NaN stays NaN
}

// `tab` = Fp64Tables image in LDS (unused by the fp32 path, which has v_exp_f32 / v_log_f32)
template <typename T> __device__ __forceinline__ T t_exp(T x, const double* tab);
template <> __device__ __forceinline__ float t_exp<float>(float x, const double*) { return __expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x, const double* tab) { return exp_nonpos(x, tab); }
template <typename T> __device__ __forceinline__ T t_log(T x, const double* tab);
template <> __device__ __forceinline__ float t_log<float>(float x, const double*) { return __logf(x); }
template <> __device__ __forceinline__ double t_log<double>(double x, const double* tab) {
    return log_pos(x, tab + 128, tab + 256);
}

// Cross-lane pair exchange without LDS: gfx950's v_permlane16_swap / v_permlane32_swap swap
// 16-lane rows (resp. 32-lane halves) between two registers; fed the same value twice they
// leave (own row, partner row) pairs in the two results, so op(a, b) is the xor-16 (xor-32)
// butterfly -- 1 VALU op per dword instead of a ds_bpermute round trip.
template <int W> __device__ __forceinline__ void swap_rows(unsigned v, unsigned& a, unsigned& b) {
    if (W == 16) {
        auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        a = r[0]; b = r[1];
    } else {
        auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        a = r[0]; b = r[1];
    }
}
template <int W> __device__ __forceinline__ void pair_of(float v, float& a, float& b) {
    unsigned x, y;
    swap_rows<W>(__float_as_uint(v), x, y);
    a = __uint_as_float(x); b = __uint_as_float(y);
}
template <int W> __device__ __forceinline__ void pair_of(double v, double& a, double& b) {
    unsigned xl, yl, xh, yh;
    swap_rows<W>((unsigned)__double2loint(v), xl, yl);
    swap_rows<W>((unsigned)__double2hiint(v), xh, yh);
    a = __hiloint2double((int)xh, (int)xl); b = __hiloint2double((int)yh, (int)yl);
}
template <typename T, int W> __device__ __forceinline__ T pair_max(T v) { T a, b; pair_of<W>(v, a, b); return fmax(a, b); }
template <typename T, int W> __device__ __forceinline__ T pair_sum(T v) { T a, b; pair_of<W>(v, a, b); return a + b; }

// -(mx + log sm), branch-free: sm == 0 (every component off) gives +inf, NaN stays NaN
template <typename T> __device__ __forceinline__ T nll_of(T mx, T sm, const double* tab) {
    return -(mx + t_log<T>(sm, tab));
}

// max and sum-of-exp over the 4 registers of one lane, then over `width` lane groups (1, 2 or 4)
template <typename T, typename V>
__device__ __forceinline__ void tile_lse(const V& a, int width, T& mx, T& sm, const double* tab) {
    T m = fmax(fmax(a[0], a[1]), fmax(a[2], a[3]));
    if (width >= 2) m = pair_max<T, 16>(m);
    if (width >= 4) m = pair_max<T, 32>(m);
    const T ms = fmax(m, T(-1e300));  // all components off (-inf): exp(-inf + 1e300) = 0, no inf - inf
    T e = t_exp<T>(a[0] - ms, tab) + t_exp<T>(a[1] - ms, tab) + t_exp<T>(a[2] - ms, tab) + t_exp<T>(a[3] - ms, tab);
    // NaN parameters / features must poison the state (the reference's linear-domain sum does), but
    // fmax() drops NaNs: the plain sum of the four values is NaN exactly when one of them is
    const T poison = (a[0] + a[1]) + (a[2] + a[3]);
    e = (poison != poison) ? poison : e;
    if (width >= 2) e = pair_sum<T, 16>(e);
    if (width >= 4) e = pair_sum<T, 32>(e);
    mx = m;
    sm = e;
}

// One tile's epilogue: log-sum-exp over the mixture components held in the accumulators of
// both column tiles, written into the LDS output tile.  MP = padded mixture size (compile
// time); MP == 32 stands for "several tiles per state" (M_pad = 16 * tiles_per_state, run
// time), merged through (run_mx, run_sm).  Straight-line code: lanes with nothing to store
// write to a per-lane dummy slot behind the tile, so the whole epilogue can be scheduled
// under the next tile's MFMAs.
template <typename T, typename V, int MP>
__device__ __forceinline__ void tile_epilogue(const V& acc0, const V& acc1, int t, int f, int q, int S, int RS,
                                              int chunk_s0, int tiles_per_state, T* lds, T* dummy,
                                              const double* tab, T (&run_mx)[2], T (&run_sm)[2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const V& acc = c ? acc1 : acc0;
        T* orow = lds + (16 * c + f) * RS - chunk_s0;
        if (MP == 1) {
            const int s = 16 * t + 4 * q;
#pragma unroll
            for (int r = 0; r < 4; ++r) *((s + r < S) ? orow + s + r : dummy) = -acc[r];
        } else if (MP == 2) {
            const int s = 8 * t + 2 * q;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const T x0 = acc[2 * h], x1 = acc[2 * h + 1];
                const T m = fmax(x0, x1);
                const T ms = fmax(m, T(-1e300));
                T e = t_exp<T>(x0 - ms, tab) + t_exp<T>(x1 - ms, tab);
                const T poison = x0 + x1;  // NaN iff one of them is (fmax drops NaNs)
                e = (poison != poison) ? poison : e;
                *((s + h < S) ? orow + s + h : dummy) = nll_of<T>(m, e, tab);
            }
        } else if (MP <= 16) {
            constexpr int width = MP / 4;  // lane groups per state: 1, 2 or 4
            T mx, sm;
            tile_lse<T, V>(acc, width, mx, sm, tab);
            const int s = (16 / MP) * t + q / width;
            *(((q & (width - 1)) == 0 && s < S) ? orow + s : dummy) = nll_of<T>(mx, sm, tab);
        } else {
            T mx, sm;
            tile_lse<T, V>(acc, 4, mx, sm, tab);
            const T m = fmax(run_mx[c], mx);
            const T ms = fmax(m, T(-1e300));
            run_sm[c] = run_sm[c] * t_exp<T>(run_mx[c] - ms, tab) + sm * t_exp<T>(mx - ms, tab);
            run_mx[c] = m;
            const bool last = (t + 1) % tiles_per_state == 0;
            const int s = t / tiles_per_state;
            *((last && q == 0 && s < S) ? orow + s : dummy) = nll_of<T>(run_mx[c], run_sm[c], tab);
            run_mx[c] = last ? T(-INFINITY) : run_mx[c];
            run_sm[c] = last ? T(0) : run_sm[c];
        }
    }
}

// KS = number of k-steps (K = 4*KS = 2*KP).  A fragments travel through a ring of R = KS/2
// registers: slot j serves k-step j, is refilled with k-step j+R of the same tile, serves it,
// and is refilled with k-step j of the NEXT tile -- every load has half a tile of MFMAs
// (R*2 instructions, >= 1280 cycles in fp64) to return from L2.
// Measured on MI355X (tools/variant_bench.sh): forcing an MFMA/VALU interleave with
// sched_group_barrier is 6-10 % SLOWER than letting the MFMAs issue back to back (fp32/fp64
// MFMA and VALU share one pipeline: nothing overlaps, the interleave only adds bubbles), and a
// 3-waves/SIMD register cap helps fp32 (+4 %) but costs fp64 (-4 %).
#ifndef GH_MF_SGB
#define GH_MF_SGB 0
#endif
template <typename T, int KS, int MP>
__global__ __launch_bounds__(64, (sizeof(T) == 4 ? 3 : 1)) void loglik_mfma_kernel(const T* __restrict__ X, int64_t N, int D,
                                                         const T* __restrict__ Apk, const T* __restrict__ Cpk,
                                                         int n_tiles, int S, int M_pad, int chunk_tiles,
                                                         const double* __restrict__ tables, int tab_off,
                                                         T* __restrict__ out) {
    typedef typename Acc<T>::type V;
    constexpr int R = KS / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int lane = threadIdx.x;
    const int f = lane & 15, q = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * 32;
    const int nrows = (int)((N - n0 < 32) ? (N - n0) : 32);
    double* tab = reinterpret_cast<double*>(smem_raw + tab_off);  // exp / log tables (fp64 path)
    if (sizeof(T) == 8) {
        for (int i = lane; i < 384; i += 64) tab[i] = tables[i];
    }

    // ---- frames: global -> LDS (coalesced) -> B fragments in registers ----------------
    {
        const int64_t nelem = (int64_t)nrows * D;
        const T* src = X + n0 * D;
        for (int i = lane; i < nelem; i += 64) lds[i] = src[i];
        __syncthreads();
    }
    constexpr int KQ = KS / 2;  // k-steps of the x^2 half == of the x half (KP = 4*KQ)
    T b[2][KS];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int row = 16 * c + f;
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const int d = 4 * j + q;
            const T v = (row < nrows && d < D) ? lds[row * D + d] : T(0);
            b[c][j] = v * v;
            b[c][KQ + j] = v;
        }
    }
    __syncthreads();

    const int tiles_per_state = (MP <= 16) ? 1 : M_pad / 16;
    const int SC = (MP <= 16) ? chunk_tiles * (16 / (MP <= 16 ? MP : 16)) : chunk_tiles / tiles_per_state;
    const int RS = (S <= SC) ? S : SC;  // LDS row stride: whole matrix rows when they fit one chunk
    T* dummy = lds + 32 * RS + lane;    // per-lane slot behind the output tile
    T run_mx[2] = {-INFINITY, -INFINITY}, run_sm[2] = {T(0), T(0)};
    int chunk_s0 = 0;  // first state held in the LDS output tile

    // ---- stream the Gaussian tiles; the epilogue of tile t-1 runs under the MFMAs of tile t ----
    // (the host pads Apk / Cpk with one all-zero tile, so the run-ahead loads stay in bounds)
    T ring[R];
    const T* ap = Apk + lane;  // next fragment to fetch
#pragma unroll
    for (int j = 0; j < R; ++j) ring[j] = ap[j * 64];
    ap += R * 64;
    V c_nxt, p0, p1;
#pragma unroll
    for (int r = 0; r < 4; ++r) c_nxt[r] = Cpk[4 * q + r];

    // one tile of MFMAs: accumulators start at C (prefetched), ring refilled as it is consumed
    auto mfma_tile = [&](int t, V& acc0, V& acc1) {
        acc0 = c_nxt;
        acc1 = c_nxt;
        const T* cp = Cpk + (t + 1) * 16 + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) c_nxt[r] = cp[r];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const T a = ring[ks % R];
            acc0 = Acc<T>::mfma(a, b[0][ks], acc0);
            acc1 = Acc<T>::mfma(a, b[1][ks], acc1);
            ring[ks % R] = ap[ks * 64];
        }
        ap += KS * 64;
    };
    auto flush = [&]() {
        const int cnt = ((chunk_s0 + SC < S) ? chunk_s0 + SC : S) - chunk_s0;  // states in this chunk
        __syncthreads();
        if (cnt == S) {  // whole rows: the [nrows, S] block is contiguous in memory
            T* dst = out + n0 * S;
            const int total = nrows * S;
            for (int i = lane; i < total; i += 64) dst[i] = lds[i];
        } else if (cnt > 0) {
            for (int r = 0; r < nrows; ++r) {
                T* dst = out + (n0 + r) * S + chunk_s0;
                for (int j = lane; j < cnt; j += 64) dst[j] = lds[r * RS + j];
            }
        }
        __syncthreads();
        chunk_s0 += SC;
    };

#ifndef GH_MF_PIPE
#define GH_MF_PIPE 1
#endif
#if GH_MF_PIPE
    mfma_tile(0, p0, p1);
    for (int t = 1; t < n_tiles; ++t) {
        V acc0, acc1;
        mfma_tile(t, acc0, acc1);
        tile_epilogue<T, V, MP>(p0, p1, t - 1, f, q, S, RS, chunk_s0, tiles_per_state, lds, dummy, tab, run_mx, run_sm);
        // schedule: one MFMA, then a slice of the previous tile's epilogue VALU work
#pragma unroll
        for (int i = 0; i < (GH_MF_SGB ? 2 * KS : 0); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, sizeof(T) == 8 ? 5 : 3, 0);  // VALU
        }
        if (t % chunk_tiles == 0) flush();
        p0 = acc0;
        p1 = acc1;
    }
    tile_epilogue<T, V, MP>(p0, p1, n_tiles - 1, f, q, S, RS, chunk_s0, tiles_per_state, lds, dummy, tab, run_mx, run_sm);
    flush();
#else
    for (int t = 0; t < n_tiles; ++t) {
        mfma_tile(t, p0, p1);
        tile_epilogue<T, V, MP>(p0, p1, t, f, q, S, RS, chunk_s0, tiles_per_state, lds, dummy, tab, run_mx, run_sm);
        if ((t + 1) % chunk_tiles == 0 && t + 1 < n_tiles) flush();
    }
    flush();
#endif
}

template <typename T>
int launch_mfma_t(gh_ctx* ctx, const gh_gmm* g, gh_batch* b, const T* Apk, const T* Cpk) {
    const int64_t N = b->N;
    if (N == 0) return GH_OK;
    const int KS = g->KP / 2;
    const int M_pad = g->M_pad, n_tiles = g->n_tiles, S = g->S;
    // states per LDS output chunk: whole matrix rows when they fit 64 states, else 64-state chunks
    int chunk_tiles;
    if (M_pad <= 16) {
        const int spt = 16 / M_pad;
        chunk_tiles = (S <= 64) ? n_tiles : std::max(1, 64 / spt);
    } else {
        const int tps = M_pad / 16;
        chunk_tiles = (S <= 64) ? n_tiles : 64 * tps;
    }
    const int SC = (M_pad <= 16) ? chunk_tiles * (16 / M_pad) : chunk_tiles / (M_pad / 16);
    size_t lds = ((size_t)32 * std::max(std::min(SC, S), g->D) + 64) * sizeof(T);
    lds = (lds + 15) & ~size_t(15);
    const int tab_off = (int)lds;
    if (sizeof(T) == 8) lds += 384 * sizeof(double);
    const double* tables = ctx->d_fp64_tables;
    const unsigned grid = (unsigned)((N + 31) / 32);
    const T* X = static_cast<const T*>(b->feats);
    T* out = static_cast<T*>(b->nll);
#define GH_MF_LAUNCH(ks, mp)                                                                                \
    hipLaunchKernelGGL((loglik_mfma_kernel<T, ks, mp>), dim3(grid), dim3(64), lds, ctx->stream, X, N, g->D, \
                       Apk, Cpk, n_tiles, S, M_pad, chunk_tiles, tables, tab_off, out)
#define GH_MF_CASE(ks)                                   \
    case ks:                                             \
        switch (M_pad) {                                 \
            case 1: GH_MF_LAUNCH(ks, 1); break;          \
            case 2: GH_MF_LAUNCH(ks, 2); break;          \
            case 4: GH_MF_LAUNCH(ks, 4); break;          \
            case 8: GH_MF_LAUNCH(ks, 8); break;          \
            case 16: GH_MF_LAUNCH(ks, 16); break;        \
            default: GH_MF_LAUNCH(ks, 32); break;        \
        }                                                \
        break;
    switch (KS) {
        GH_MF_CASE(2)
        GH_MF_CASE(4)
        GH_MF_CASE(8)
        GH_MF_CASE(12)
        GH_MF_CASE(20)
        default:
            return 1;  // not an MFMA shape: caller falls back to the vector kernel
    }
#undef GH_MF_CASE
#undef GH_MF_LAUNCH
    GH_HIP(hipGetLastError());
    return GH_OK;
}

}  // namespace

// returns 1 when the shape is not covered (caller uses the VALU kernel), <0 on error
int gh_launch_loglik_mfma(gh_ctx* ctx, const gh_gmm* g, gh_batch* b) {
    if (!g->dApk64) return 1;
    if (b->dtype == GH_F64) return launch_mfma_t<double>(ctx, g, b, g->dApk64, g->dCpk64);
    return launch_mfma_t<float>(ctx, g, b, g->dApk32, g->dCpk32);
}
