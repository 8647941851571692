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
//   * a wave owns 32 frames at a time.  Their Z operand (B fragments of both 16-frame column
//     tiles, all K) is loaded ONCE -- coalesced global -> LDS tile -> registers, every load in
//     flight before the first wait -- and stays in VGPRs while the Gaussians stream past;
//   * the Gaussians stream past as 16-row tiles.  The host packs P so that every A fragment is 64
//     consecutive elements in lane order (one coalesced 512-byte / 256-byte load per k-step); all
//     waves read the same 256 KB, which lives in L2 (~60 % of it is served by the CU's L1).  The
//     fragments travel through a register ring a whole tile (fp64) / half a tile (fp32) ahead;
//   * the first k-step of a tile takes C[g] as its addend, so the accumulators finish as component
//     log-densities -- in a SCALED log domain (see below) that makes the log-sum-exp cheap.  The
//     row order inside a tile is chosen per dtype (host side) such that lane group q = lane>>4
//     holds components 4q..4q+3 of the tile in its 4 accumulator registers -- for M = 8 the
//     log-sum-exp is 4 in-register terms + one xor-16 exchange;
//   * the epilogue of tile t-1 is scheduled with the MFMAs of tile t (fp64: two accumulator pairs
//     swap roles, nothing is copied);
//   * results are staged in LDS as a [32 frames, states] tile and written back as one contiguous
//     block (the [N,S] matrix is row-major), 16 bytes per lane;
//   * light models (few MFMAs per 32 frames: M = 1, or a state subset) use the MULTI instantiation:
//     a wave walks several blocks with the operand ring kept primed (closed loop over the tiles),
//     optionally driven by a block table that restricts every utterance to its own state range
//     (gh_loglik_subset).
#include "gh_internal.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// block table entry of gh_loglik_subset: frames [n0, n0 + nrows) (one utterance), Gaussian tiles [t0, t1)
struct gh_loglik_blk { int64_t n0; int32_t nrows, t0, t1, pad; };

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

// ---- the log-sum-exp epilogue works in a SCALED log domain ---------------------------------
// On gfx950 fp64 MFMA and fp64 VALU share the same FMA pipeline (measured with
// tools/mfma_coissue.hip: their times ADD), so every fp64 VALU instruction of the epilogue is
// paid in full next to the MFMAs.  The host therefore packs P and C multiplied by
//     K = 128/ln 2 (fp64)   or   K = 1/ln 2 (fp32),
// so the accumulators hold K * log-density and
//   * fp32: exp(a - m) is ONE v_exp_f32 (2^y), log is ONE v_log_f32;
//   * fp64: y = a - m is already in units of 1/128 octave: n = rint(y), r = y - n EXACTLY,
//     2^(y/128) = 2^(n>>7) * T[n&127] * 2^(r/128), the last factor by a degree-4 polynomial
//     (|r| <= 1/2: relative error < 1.3e-15, absolute error of the result ~1e-17 of |nll|);
//     no argument clamp is needed because weight-0 / padding components carry the FINITE
//     constant GH_LSE_OFF (y = -1e300 -> n = y, r = 0, ldexp underflows to an exact 0) and
//     NaNs propagate through the arithmetic by themselves;
//   * log: s = m 2^e, m in [.5,1); m * INV[j] = 1 + rho, |rho| <= 2^-8 (j = top 7 mantissa
//     bits); K ln s = 128 e + L[j] + K log1p(rho), degree-5 polynomial (abs. error < 2e-13 / K).
// Tables (128 entries each, built in long double by gh_ctx_create) live in LDS.
struct Fp64Tables {
    double exp2[128];   // 2^(j/128)
    double inv[128];    // 1 / centre of mantissa bin j, centre = 0.5 + (j + 0.5)/256
    double nlog[128];   // -K log(inv[j])
};

template <typename T> struct Dom;
template <> struct Dom<double> {
    static constexpr double inv_k = 1.0 / GH_LSE_SCALE64, off = GH_LSE_OFF64, off_test = GH_LSE_OFF64 * 0.1;
};
template <> struct Dom<float> {
    static constexpr float inv_k = (float)(1.0 / GH_LSE_SCALE32), off = GH_LSE_OFF32, off_test = GH_LSE_OFF32 * 0.1f;
};

// IEEE maxNum in ONE instruction (fmax() adds a v_max x,x canonicalisation per operand that comes
// out of an MFMA); a NaN operand loses, two NaNs give NaN
__device__ __forceinline__ double vmax(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// 2^(y/128) for finite y <= 0 (or NaN)
__device__ __forceinline__ double exp2s(double y, const double* __restrict__ tab) {
    const double n = __builtin_rint(y);
    const double r = y - n;
    const int ni = (int)n;  // v_cvt_i32_f64 saturates
    const double t = tab[ni & 127];
    double p = fma(r, 3.583032305400251285e-11, 2.6466421444330968834e-08);  // (ln2/128)^k / k!, k = 4, 3
    p = fma(p, r, 1.4662262387640424337e-05);
    p = fma(p, r, 5.4152123481245727298e-03);
    p = p * r;  // 2^(r/128) - 1
    return __builtin_ldexp(fma(t, p, t), ni >> 7);
}
__device__ __forceinline__ float exp2s(float y, const double*) { return __builtin_amdgcn_exp2f(y); }

// K ln s for s in [1, M] (or NaN)
__device__ __forceinline__ double logs(double s, const double* __restrict__ tab) {
    const double* __restrict__ inv = tab + 128;
    const double* __restrict__ nlog = tab + 256;
    const double m = __builtin_amdgcn_frexp_mant(s);  // [0.5, 1)
    const int e = __builtin_amdgcn_frexp_exp(s);
    const int j = (__double2hiint(m) >> 13) & 127;
    const double rho = fma(m, inv[j], -1.0);
    double p = fma(rho, 36.932993046757463228, -46.166241308446829036);  // K/5, -K/4
    p = fma(p, rho, 61.554988411262438714);                              // K/3
    p = fma(p, rho, -92.332482616893658071);                             // -K/2
    p = fma(p, rho, 184.66496523378731614);                              // K
    return fma(p, rho, nlog[j]) + (double)(e << 7);
}
__device__ __forceinline__ float logs(float s, const double*) { return __builtin_amdgcn_logf(s); }  // v_log_f32 = log2

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
template <typename T, int W> __device__ __forceinline__ T pair_max(T v) { T a, b; pair_of<W>(v, a, b); return vmax(a, b); }
template <typename T, int W> __device__ __forceinline__ T pair_sum(T v) { T a, b; pair_of<W>(v, a, b); return a + b; }

// scaled (max, sum of 2^..) -> negative log-likelihood; every component off -> +inf; NaN stays NaN
// inf_below: the scaled log below which the largest term counts as "off".  Default Dom<T>::off_test (only switched-off
// components); compat mode (gh_ctx_set_compat bit 0): K ln 2^-1075 -- the reference sums w pdf in the LINEAR domain
// (hmm_state.py:114-120), where every term whose logarithm is below that rounds to 0 and the state costs -log 0 = +inf
template <typename T> __device__ __forceinline__ T nll_of(T mx, T sm, const double* tab, T inf_below) {
    const T v = (mx + logs(sm, tab)) * -Dom<T>::inv_k;
    return (mx < inf_below) ? T(INFINITY) : v;
}

// max and sum-of-exp over the 4 registers of one lane, then over `width` lane groups (1, 2 or 4)
template <typename T, typename V>
__device__ __forceinline__ void tile_lse(const V& a, int width, T& mx, T& sm, const double* tab) {
    T m = vmax(vmax(a[0], a[1]), vmax(a[2], a[3]));
    if (width >= 2) m = pair_max<T, 16>(m);
    if (width >= 4) m = pair_max<T, 32>(m);
    // a NaN component makes its own term (and so the sum) NaN: the state is poisoned as in the
    // reference's linear-domain sum, although vmax() dropped the NaN
    T e = (exp2s(a[0] - m, tab) + exp2s(a[1] - m, tab)) + (exp2s(a[2] - m, tab) + exp2s(a[3] - m, tab));
    if (width >= 2) e = pair_sum<T, 16>(e);
    if (width >= 4) e = pair_sum<T, 32>(e);
    mx = m;
    sm = e;
}

// ---- opt-in epilogue with the exponentials in FP32 (gh_ctx_set_compat bit 1 / GMMHMM_LSE=f32exp; fp64 batches only).
// The maximum, `a - max` and the final `max + log(sum)` stay fp64; the terms 2^((a - max)/128) <= 1 go through
// v_cvt_f32_f64 + v_exp_f32, their sum and its v_log_f32 stay fp32: 2 instructions on the shared fp64 pipe per term
// instead of ~10.  A term carries v_exp_f32's relative error (~1.2e-7), so the log-sum-exp -- and with it the
// negative log-likelihood -- moves by at most ~2.4e-7 ABSOLUTE (on costs of ~60: 4e-9 relative, north star 1e-5).
// Off by default: bench.py reports the kernel both ways together with max |delta nll| and the path mismatch rates.
__device__ __forceinline__ float exp2s_fe(double y) { return __builtin_amdgcn_exp2f((float)y * 0.0078125f); }
template <int W> __device__ __forceinline__ float pair_sum_f(float v) { float a, b; pair_of<W>(v, a, b); return a + b; }
template <typename V>
__device__ __forceinline__ void tile_lse_fe(const V& a, int width, double& mx, float& sm) {
    double m = vmax(vmax(a[0], a[1]), vmax(a[2], a[3]));
    if (width >= 2) m = pair_max<double, 16>(m);
    if (width >= 4) m = pair_max<double, 32>(m);
    float e = (exp2s_fe(a[0] - m) + exp2s_fe(a[1] - m)) + (exp2s_fe(a[2] - m) + exp2s_fe(a[3] - m));
    if (width >= 2) e = pair_sum_f<16>(e);
    if (width >= 4) e = pair_sum_f<32>(e);
    mx = m;
    sm = e;
}
__device__ __forceinline__ double nll_of_fe(double mx, float sm, double inf_below) {
    const double v = (mx + (double)(128.0f * __builtin_amdgcn_logf(sm))) * -Dom<double>::inv_k;     // v_log_f32 = log2
    return (mx < inf_below) ? (double)INFINITY : v;
}

// One tile's epilogue: log-sum-exp over the mixture components held in the accumulators of
// both column tiles, written into the LDS output tile.  MP = padded mixture size (compile
// time); MP == 32 stands for "several tiles per state" (M_pad = 16 * tiles_per_state, run
// time), merged through (run_mx, run_sm).  Straight-line code: lanes with nothing to store
// write to a per-lane dummy slot behind the tile.  When a state spans two or four lane groups
// every group ends up with the same (max, sum) pair, so the even group finishes column tile 0
// and the odd group column tile 1: ONE logarithm per lane instead of two.
template <typename T, typename V, int MP, bool FE = false>
__device__ __forceinline__ void tile_epilogue(const V& acc0, const V& acc1, int t, int f, int q, int S, int RS,
                                              int chunk_s0, int tiles_per_state, T* lds, T* dummy,
                                              const double* tab, T (&run_mx)[2], T (&run_sm)[2], T inf_below) {
    T* orow0 = lds + f * RS - chunk_s0;
    T* orow1 = orow0 + 16 * RS;
    if (MP == 1) {
        const int s = 16 * t + 4 * q;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const V& acc = c ? acc1 : acc0;
            T* orow = c ? orow1 : orow0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *((s + r < S) ? orow + s + r : dummy) = (acc[r] < inf_below) ? T(INFINITY) : acc[r] * -Dom<T>::inv_k;
        }
    } else if (MP == 2) {
        const int s = 8 * t + 2 * q;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const V& acc = c ? acc1 : acc0;
            T* orow = c ? orow1 : orow0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const T x0 = acc[2 * h], x1 = acc[2 * h + 1];
                const T m = vmax(x0, x1);
                const T e = exp2s(x0 - m, tab) + exp2s(x1 - m, tab);
                *((s + h < S) ? orow + s + h : dummy) = nll_of<T>(m, e, tab, inf_below);
            }
        }
#ifdef GH_MF_NOEPI  // diagnostic build: MFMAs + loads only (tools/variant_bench.sh)
    } else if (MP <= 16) {
        const int s = (16 / MP) * t + q / (MP / 4);
        *(((q & (MP / 4 - 1)) == 0 && s < S) ? orow0 + s : dummy) = -((acc0[0] + acc0[1]) + (acc0[2] + acc0[3]));
        *(((q & (MP / 4 - 1)) == 0 && s < S) ? orow1 + s : dummy) = -((acc1[0] + acc1[1]) + (acc1[2] + acc1[3]));
#endif
    } else if (FE && MP == 4) {
        if constexpr (FE) {
            double mx; float sm;
            const int s = 4 * t + q;
            tile_lse_fe<V>(acc0, 1, mx, sm);
            *((s < S) ? orow0 + s : dummy) = nll_of_fe(mx, sm, inf_below);
            tile_lse_fe<V>(acc1, 1, mx, sm);
            *((s < S) ? orow1 + s : dummy) = nll_of_fe(mx, sm, inf_below);
        }
    } else if (FE && MP <= 16) {
        if constexpr (FE) {
            constexpr int width = MP / 4;
            double mx0, mx1; float sm0, sm1;
            tile_lse_fe<V>(acc0, width, mx0, sm0);
            tile_lse_fe<V>(acc1, width, mx1, sm1);
            const bool odd = q & 1;
            const int s = (16 / MP) * t + q / width;
            const double v = nll_of_fe(odd ? mx1 : mx0, odd ? sm1 : sm0, inf_below);
            *(((q & (width - 1)) < 2 && s < S) ? (odd ? orow1 : orow0) + s : dummy) = v;
        }
    } else if (FE) {
        if constexpr (FE) {
            const bool last = (t + 1) % tiles_per_state == 0;
            const int s = t / tiles_per_state;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                double mx; float smf;
                tile_lse_fe<V>(c ? acc1 : acc0, 4, mx, smf);
                const double d = run_mx[c] - mx;
                const float e = exp2s_fe((d > 0.0) ? -d : d);
                const float rs = (float)run_sm[c];              // (the running sum is an fp32 value kept in the fp64 slot)
                run_sm[c] = (double)((d > 0.0) ? fmaf(smf, e, rs) : fmaf(rs, e, smf));
                run_mx[c] = (d > 0.0) ? run_mx[c] : mx;
            }
            const bool odd = q & 1;
            const double v = nll_of_fe(odd ? run_mx[1] : run_mx[0], (float)(odd ? run_sm[1] : run_sm[0]), inf_below);
            *((last && q < 2 && s < S) ? (odd ? orow1 : orow0) + s : dummy) = v;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                run_mx[c] = last ? Dom<T>::off : run_mx[c];
                run_sm[c] = last ? T(0) : run_sm[c];
            }
        }
    } else if (MP == 4) {
        T mx, sm;
        const int s = 4 * t + q;
        tile_lse<T, V>(acc0, 1, mx, sm, tab);
        *((s < S) ? orow0 + s : dummy) = nll_of<T>(mx, sm, tab, inf_below);
        tile_lse<T, V>(acc1, 1, mx, sm, tab);
        *((s < S) ? orow1 + s : dummy) = nll_of<T>(mx, sm, tab, inf_below);
    } else if (MP <= 16) {
        constexpr int width = MP / 4;  // lane groups per state: 2 or 4
        T mx0, sm0, mx1, sm1;
        tile_lse<T, V>(acc0, width, mx0, sm0, tab);
        tile_lse<T, V>(acc1, width, mx1, sm1, tab);
        const bool odd = q & 1;
        const int s = (16 / MP) * t + q / width;
        const T v = nll_of<T>(odd ? mx1 : mx0, odd ? sm1 : sm0, tab, inf_below);
        *(((q & (width - 1)) < 2 && s < S) ? (odd ? orow1 : orow0) + s : dummy) = v;
    } else {
        const bool last = (t + 1) % tiles_per_state == 0;
        const int s = t / tiles_per_state;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            T mx, sm;
            tile_lse<T, V>(c ? acc1 : acc0, 4, mx, sm, tab);
            // merge with the running pair: one of the two rescale factors is 2^0
            const T d = run_mx[c] - mx;
            const T e = exp2s((d > T(0)) ? -d : d, tab);
            run_sm[c] = (d > T(0)) ? fma(sm, e, run_sm[c]) : fma(run_sm[c], e, sm);
            run_mx[c] = (d > T(0)) ? run_mx[c] : mx;
        }
        const bool odd = q & 1;
        const T v = nll_of<T>(odd ? run_mx[1] : run_mx[0], odd ? run_sm[1] : run_sm[0], tab, inf_below);
        *((last && q < 2 && s < S) ? (odd ? orow1 : orow0) + s : dummy) = v;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            run_mx[c] = last ? Dom<T>::off : run_mx[c];
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
// MULTI: the wave walks `bpw` blocks (light models); false = one block per wave, the block loop folds away.
template <typename T, int KS, int MP, bool MULTI, bool FE = false>
__global__ __launch_bounds__(64, (sizeof(T) == 4 ? 3 : 1)) void loglik_mfma_kernel(const T* __restrict__ X, int64_t N, int D,
                                                         const T* __restrict__ Apk, const T* __restrict__ Cpk,
                                                         int n_tiles, int S, int M_pad, int chunk_tiles,
                                                         const double* __restrict__ tables, int tab_off,
                                                         T* __restrict__ out, int bpw,
                                                         const gh_loglik_blk* __restrict__ blk_tab, int64_t n_blk,
                                                         const T* __restrict__ cen, T inf_below) {
    typedef typename Acc<T>::type V;
#ifndef GH_MF_RING32
#define GH_MF_RING32 2
#endif
#ifndef GH_MF_RING64
#define GH_MF_RING64 1
#endif
    constexpr int R = KS / (sizeof(T) == 4 ? GH_MF_RING32 : GH_MF_RING64);  // ring depth: half a tile (2) or a whole tile (1) of run-ahead
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int lane = threadIdx.x;
    const int f = lane & 15, q = lane >> 4;
    // A wave processes `bpw` consecutive 32-frame blocks (1 for models with enough work per block; several for
    // light models -- M = 1, few states -- where launching one wave per 32 frames and re-priming the operand ring
    // cost more than the block's 64 MFMAs).  The Gaussian tiles stream in a closed loop: the last tile's run-ahead
    // loads fetch tile 0 again, so the ring is primed when the wave moves to its next block.
    const int64_t n_blocks = (N + 31) / 32;
    int64_t n0 = (int64_t)blockIdx.x * (MULTI ? bpw : 1) * 32;
#ifdef GH_MF_TIMING  // diagnostic build (tools/wave_timeline.py): per-wave phase time stamps overwrite row n0 of the output
    long long tk[5], tw[2];
    tk[0] = clock64(); tw[0] = wall_clock64();
#endif
    int nrows = 0;
    // (MULTI only) a block table restricts every block to the tiles [t_lo, t_hi) of its utterance's states
    const bool subset = MULTI && blk_tab != nullptr;
    const int64_t kb0 = (int64_t)blockIdx.x * (MULTI ? bpw : 1);
    int t_lo = subset ? blk_tab[kb0].t0 : 0, t_hi = n_tiles, nxt_t0 = 0;
    double* tab = reinterpret_cast<double*>(smem_raw + tab_off);  // exp / log tables (fp64 path)
    if (sizeof(T) == 8 && MP != 1) {   // (a single-component state needs no exp / log)
        double tr[6];
#pragma unroll
        for (int it = 0; it < 6; ++it) tr[it] = tables[lane + 64 * it];
#pragma unroll
        for (int it = 0; it < 6; ++it) tab[lane + 64 * it] = tr[it];
    }
    // Every load of a block's prologue is issued before the first wait (a load -> wait -> ds_write loop
    // costs one HBM round trip per 64 elements: 20 in a row for a 32 x 39 tile).
    auto stage_frames = [&]() {
        // ---- frames: global -> LDS (coalesced) -> B fragments in registers ----------------
        const int nelem = nrows * D;  // <= 32 * KP = 64 * KS
        const T* src = X + n0 * D;
        T xr[KS];
#pragma unroll
        for (int it = 0; it < KS; ++it) {
            const int i = lane + 64 * it;
            xr[it] = (i < nelem) ? src[i] : T(0);
        }
#pragma unroll
        for (int it = 0; it < KS; ++it) {
            const int i = lane + 64 * it;
            if (i < nelem) lds[i] = xr[it];
        }
        __syncthreads();
    };
    constexpr int KQ = KS / 2;  // k-steps of the x^2 half == of the x half (KP = 4*KQ)
    T b[2][KS];
    // fp32: the operands are packed for centred features x - cen (gh_internal.h, dCen32): this lane's dimensions
    T cen_r[KQ];
#pragma unroll
    for (int j = 0; j < KQ; ++j) cen_r[j] = (sizeof(T) == 4 && cen) ? cen[4 * j + q] : T(0);
    auto build_b = [&]() {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int row = 16 * c + f;
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                const int d = 4 * j + q;
                T v = (row < nrows && d < D) ? lds[row * D + d] : T(0);
                if (sizeof(T) == 4) v = (row < nrows && d < D) ? v - cen_r[j] : T(0);
                b[c][j] = v * v;
                b[c][KQ + j] = v;
            }
        }
        __syncthreads();
    };

    const int tiles_per_state = (MP <= 16) ? 1 : M_pad / 16;
    const int SC = (MP <= 16) ? chunk_tiles * (16 / (MP <= 16 ? MP : 16)) : chunk_tiles / tiles_per_state;
    const int RS = (S <= SC) ? S : SC;  // LDS row stride: whole matrix rows when they fit one chunk
    T* dummy = lds + 32 * RS + lane;    // per-lane slot behind the output tile
    T run_mx[2] = {Dom<T>::off, Dom<T>::off}, run_sm[2] = {T(0), T(0)};
    int chunk_s0 = 0;  // first state held in the LDS output tile
    int subset_cnt = 0;  // (block table) number of states the block's tile range covers

    // ---- stream the Gaussian tiles; the epilogue of tile t-1 runs under the MFMAs of tile t ----
    // (the host pads Apk / Cpk with one all-zero tile, so the run-ahead loads stay in bounds)
    T ring[R];
#pragma unroll
    for (int j = 0; j < R; ++j) ring[j] = Apk[((int64_t)t_lo * KS + j) * 64 + lane];
    V c_a, c_b, p0, p1;   // C values of the current / the next tile (roles alternate)
#pragma unroll
    for (int r = 0; r < 4; ++r) c_a[r] = Cpk[t_lo * 16 + 4 * q + r];

    // one tile of MFMAs: the first k-step takes the (prefetched) C values as its addend -- no accumulator
    // initialisation copies --, the next tile's C travels into the other register set; ring refilled as it is consumed
    auto mfma_tile = [&](int t, V& acc0, V& acc1, const V& c_use, V& c_load) {
        const int tn = (!MULTI || t + 1 < t_hi) ? t + 1 : nxt_t0;   // one block per wave: run on into the zero pad tile
        const T* cp = Cpk + tn * 16 + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) c_load[r] = cp[r];
        // slot j serves k-step j [and j + R], refilled R k-steps ahead: from this tile while that stays inside it,
        // from the next tile (tile 0 after the last) otherwise
        const T* a_cur = Apk + (int64_t)t * (KS * 64) + lane;
        const T* a_nxt = Apk + (int64_t)tn * (KS * 64) + lane;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const T a = ring[ks % R];
#ifdef GH_MF_NOMFMA  // diagnostic builds (tools/variant_bench.sh): loads + epilogue only / operands never refilled
            if (ks == 0) { acc0 = c_use; acc1 = c_use; }
            if (ks % 8 == 0) { acc0[ks % 4] += a * b[0][ks]; acc1[ks % 4] += a * b[1][ks]; }
            else asm volatile("" :: "v"(a));
#else
            acc0 = Acc<T>::mfma(a, b[0][ks], ks == 0 ? c_use : acc0);
            acc1 = Acc<T>::mfma(a, b[1][ks], ks == 0 ? c_use : acc1);
#endif
#ifndef GH_MF_NOLOAD
            ring[ks % R] = (ks + R < KS) ? a_cur[(ks + R) * 64] : a_nxt[(ks + R - KS) * 64];
#endif
        }
    };
    auto flush = [&]() {
        const int cnt = subset ? subset_cnt : ((chunk_s0 + SC < S) ? chunk_s0 + SC : S) - chunk_s0;  // states in this chunk
        __syncthreads();
        if (cnt == S && !subset) {  // whole rows: the [nrows, S] block is contiguous in memory (16 bytes per lane)
            typedef T V16 __attribute__((ext_vector_type(16 / sizeof(T))));
            constexpr int VE = 16 / sizeof(T);
            T* dst = out + n0 * S;  // 32 * S * sizeof(T) bytes per block: always 16-byte aligned
            const int total = nrows * S;
            const int nvec = total / VE;
#pragma unroll 4
            for (int i = lane; i < nvec; i += 64)
                reinterpret_cast<V16*>(dst)[i] = reinterpret_cast<const V16*>(lds)[i];
            for (int i = nvec * VE + lane; i < total; i += 64) dst[i] = lds[i];
        } else if (cnt > 0) {   // a column range of the rows: (row, column) pairs flattened over the lanes
            const int total = nrows * cnt;
            for (int i = lane; i < total; i += 64) {
                const int r = i / cnt, j = i - r * cnt;
                out[(n0 + r) * S + chunk_s0 + j] = lds[r * RS + j];
            }
        }
        __syncthreads();
        chunk_s0 += SC;
    };

    const int nb = MULTI ? bpw : 1;
    int64_t staged_n0 = -1;
    for (int ib = 0; ib < nb; ++ib, n0 += 32) {
        if (subset) {
            const int64_t k = kb0 + ib;
            if (k >= n_blk) break;
            const gh_loglik_blk bd = blk_tab[k];
            n0 = bd.n0; nrows = bd.nrows; t_lo = bd.t0; t_hi = bd.t1;
            nxt_t0 = (ib + 1 < nb && k + 1 < n_blk) ? blk_tab[k + 1].t0 : bd.t0;
        } else {
            if (MULTI && n0 >= N) break;
            nrows = (int)((N - n0 < 32) ? (N - n0) : 32);
        }
        if (!(subset && ib > 0 && n0 == staged_n0)) {   // (block table: the previous entry may cover the same frames)
            // (Round 4, measured and NOT kept: this prologue at raised priority, s_setprio 3 ... 0.  A wave that has just
            //  started is the youngest on its SIMD and the issue arbiter -- priority, then age -- serves the older waves'
            //  MFMAs first: 45 k of an fp32 wave's 191 k cycles pass in a prologue of ~300 instructions, 16 k of 171 k in
            //  fp64.  With the priority the prologues shrink to 27 k / 11 k cycles and the tile loops grow by as much:
            //  the pipe is shared, the waves' totals do not move -- fp64 1.24 -> 1.27 ms, fp32 0.69 -> 0.77 ms per launch.
            //  -DGH_MF_PRIO brings it back.)
#ifdef GH_MF_PRIO
            __builtin_amdgcn_s_setprio(3);
#endif
            stage_frames();
            build_b();
#ifdef GH_MF_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            staged_n0 = n0;
        }
        chunk_s0 = (MP <= 16) ? t_lo * (16 / (MP <= 16 ? MP : 16)) : t_lo / tiles_per_state;   // 0 without a block table
        if (subset) {
            const int s_end = (MP <= 16) ? t_hi * (16 / (MP <= 16 ? MP : 16)) : t_hi / tiles_per_state;
            subset_cnt = ((s_end < S) ? s_end : S) - chunk_s0;
        }
#ifdef GH_MF_TIMING
        tk[1] = clock64();
        tk[2] = tk[1];
#endif
        // Two tiles per iteration with the accumulator pairs swapping roles: the epilogue of tile t-1 is scheduled
        // with the MFMAs of tile t without copying accumulators (16 v_mov per tile otherwise).
        auto epi = [&](const V& e0, const V& e1, int t) {
            tile_epilogue<T, V, MP, FE>(e0, e1, t, f, q, S, RS, chunk_s0, tiles_per_state, lds, dummy, tab, run_mx, run_sm, inf_below);
#pragma unroll
            for (int i = 0; i < (GH_MF_SGB ? 2 * KS : 0); ++i) {   // (forced MFMA / VALU interleave: measured slower)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, sizeof(T) == 8 ? 5 : 3, 0);
            }
            if (!subset && (t + 1) % chunk_tiles == 0 && t + 1 < n_tiles) flush();
        };
        V r0, r1;
        // (every block starts with c_a holding tile 0's C: n_tiles tiles alternate c_a / c_b, so the wrap-around load
        //  of a multi-block wave lands in c_a only when n_tiles is even -- it is reloaded below when it is odd)
        const int tb = MULTI ? t_lo : 0, te = MULTI ? t_hi : n_tiles;   // tile range of this block
        mfma_tile(tb, p0, p1, c_a, c_b);
        if (sizeof(T) == 8) {
            int t = tb + 1;
            for (; t + 1 < te; t += 2) {
                mfma_tile(t, r0, r1, c_b, c_a);
                epi(p0, p1, t - 1);
                mfma_tile(t + 1, p0, p1, c_a, c_b);
                epi(r0, r1, t);
            }
            if (t < te) {
                mfma_tile(t, r0, r1, c_b, c_a);
                epi(p0, p1, t - 1);
                epi(r0, r1, t);
            } else {
                epi(p0, p1, te - 1);
                if (MULTI) c_a = c_b;
            }
        } else {   // fp32: the copying loop schedules better (measured 0.65 vs 0.67 ms)
            for (int t = tb + 1; t < te; ++t) {
                mfma_tile(t, r0, r1, c_b, c_a);
                epi(p0, p1, t - 1);
                p0 = r0;
                p1 = r1;
                c_b = c_a;
            }
            epi(p0, p1, te - 1);
            if (MULTI) c_a = c_b;
        }
#ifdef GH_MF_TIMING
        tk[3] = clock64();
#endif
        flush();
#ifdef GH_MF_TIMING
        tk[4] = clock64(); tw[1] = wall_clock64();
        if (lane == 0 && S >= 8) {
            T* o = out + n0 * S;
            for (int i = 0; i < 5; ++i) o[i] = (T)(double)(tk[i] - tk[0]);
            o[5] = (T)(double)(tw[0] & 0xffffffffffll); o[6] = (T)(double)(tw[1] - tw[0]);
            o[7] = (T)(double)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID
        }
#endif
    }
}

// states per LDS output chunk of the kernel: whole matrix rows when they fit 64 states, else 64-state chunks
static void chunk_shape(const gh_gmm* g, int* chunk_tiles_out, int* SC_out) {
    const int M_pad = g->M_pad, n_tiles = g->n_tiles, S = g->S;
    int chunk_tiles;
    if (M_pad <= 16) {
        const int spt = 16 / M_pad;
        chunk_tiles = (S <= 64) ? n_tiles : std::max(1, 64 / spt);
    } else {
        const int tps = M_pad / 16;
        chunk_tiles = (S <= 64) ? n_tiles : 64 * tps;
    }
    *chunk_tiles_out = chunk_tiles;
    *SC_out = (M_pad <= 16) ? chunk_tiles * (16 / M_pad) : chunk_tiles / (M_pad / 16);
}

// Block table of gh_loglik_subset / gh_loglik_sets.  rng_off == null: one state range per utterance; else utterance u
// owns the ranges [rng_off[u], rng_off[u + 1]) -- the states of its transcript's words.  Ranges become runs of Gaussian
// tiles; runs that touch are merged (two words may share a tile), and every run gets its own table entries for the
// utterance's 32-frame blocks.  Returns 1 when a run is wider than one LDS chunk (the caller computes the full matrix).
static int build_blk_table(const gh_gmm* g, const gh_batch* b, int SC, const int32_t* st_lo, const int32_t* st_hi,
                           const int64_t* rng_off, std::vector<gh_loglik_blk>& tabv, int* max_tiles_out,
                           const uint8_t* include = nullptr) {
    const int M_pad = g->M_pad, S = g->S;
    int max_tiles = 1;
    std::vector<std::pair<int, int>> runs;
    for (int64_t u = 0; u < b->U; ++u) {
        if (include && !include[u]) continue;      // (a table over part of the batch: the two halves of gh_em_iteration)
        runs.clear();
        const int64_t r0 = rng_off ? rng_off[u] : u, r1 = rng_off ? rng_off[u + 1] : u + 1;
        for (int64_t r = r0; r < r1; ++r) {
            const int lo = st_lo[r], hi = st_hi[r];
            if (lo < 0 || hi > S || lo >= hi) { gh_set_error("gh_loglik_subset: utterance %lld has state range [%d, %d)", (long long)u, lo, hi); return GH_ERR_INVALID; }
            int t0, t1;
            if (M_pad <= 16) { const int spt = 16 / M_pad; t0 = lo / spt; t1 = (hi + spt - 1) / spt; }
            else { const int tps = M_pad / 16; t0 = lo * tps; t1 = hi * tps; }
            runs.push_back({t0, t1});
        }
        std::sort(runs.begin(), runs.end());
        size_t w = 0;
        for (size_t i = 0; i < runs.size(); ++i) {
            if (w > 0 && runs[i].first <= runs[w - 1].second) runs[w - 1].second = std::max(runs[w - 1].second, runs[i].second);
            else runs[w++] = runs[i];
        }
        runs.resize(w);
        for (const auto& tr : runs) {
            const int t0 = tr.first, t1 = tr.second;
            const int span = (M_pad <= 16) ? (t1 - t0) * (16 / M_pad) : (t1 - t0) / (M_pad / 16);
            if (span > SC) return 1;
            max_tiles = std::max(max_tiles, t1 - t0);
        }
        // block-major: the entries of one 32-frame block (one per run) follow each other, so a wave that takes
        // several of them stages the frames once
        for (int64_t f = b->offsets[u]; f < b->offsets[u + 1]; f += 32)
            for (const auto& tr : runs)
                tabv.push_back(gh_loglik_blk{f, (int32_t)std::min<int64_t>(32, b->offsets[u + 1] - f), tr.first, tr.second, 0});
    }
    *max_tiles_out = max_tiles;
    return GH_OK;
}

template <typename T>
int launch_mfma_t(gh_ctx* ctx, const gh_gmm* g, gh_batch* b, const T* Apk, const T* Cpk, const int32_t* st_lo,
                  const int32_t* st_hi, const T* cen, const int64_t* rng_off, const gh_loglik_plan* plan) {
    const int64_t N = b->N;
    if (N == 0) return GH_OK;
    const int KS = g->KP / 2;
    const int M_pad = g->M_pad, n_tiles = g->n_tiles, S = g->S;
    int chunk_tiles, SC;
    chunk_shape(g, &chunk_tiles, &SC);
    size_t lds = ((size_t)32 * std::max(std::min(SC, S), g->D) + 64) * sizeof(T);
    lds = (lds + 15) & ~size_t(15);
    const int tab_off = (int)lds;
    if (sizeof(T) == 8 && M_pad != 1) lds += 384 * sizeof(double);   // exp / log tables (not needed for M = 1)
    const double* tables = ctx->d_fp64_tables;
    // ln 2^-1075 = -745.13...: exp() of anything below rounds to +0 in fp64 (the reference's arithmetic, whatever T is)
    const T inf_below = (ctx->compat & 1) ? (T)(-745.1332191019412 * (sizeof(T) == 8 ? GH_LSE_SCALE64 : GH_LSE_SCALE32)) : Dom<T>::off_test;
    // ---- optional block table: every utterance in blocks of <= 32 frames, only the tiles of its state range ----
    gh_loglik_blk* d_blk = nullptr;
    int64_t n_blk = 0;
    int max_tiles = n_tiles;
    if (plan) {   // built once by gh_loglik_plan_build (a trainer's transcripts never change)
        d_blk = static_cast<gh_loglik_blk*>(plan->d_blk);
        n_blk = plan->n_blk;
        max_tiles = plan->max_tiles;
        if (n_blk == 0) return GH_OK;
    } else if (st_lo) {
        std::vector<gh_loglik_blk> tabv;
        int rc = build_blk_table(g, b, SC, st_lo, st_hi, rng_off, tabv, &max_tiles);
        if (rc) return rc;
        n_blk = (int64_t)tabv.size();
        if (n_blk == 0) return GH_OK;
        void* base;
        rc = gh_scratch(ctx, tabv.size() * sizeof(gh_loglik_blk), &base);
        if (rc) return rc;
        d_blk = static_cast<gh_loglik_blk*>(base);
        GH_HIP(hipMemcpyAsync(d_blk, tabv.data(), tabv.size() * sizeof(gh_loglik_blk), hipMemcpyHostToDevice, ctx->stream));
        GH_HIP(hipStreamSynchronize(ctx->stream));   // tabv goes out of scope
    }
    const bool subset = plan || st_lo;
    // blocks per wave: enough MFMAs per wave (>= ~512) to amortise its launch and the ring priming
    const int64_t n_blocks = subset ? n_blk : (N + 31) / 32;
    const int per_block = std::max(1, max_tiles * KS * 2);
    int bpw = (int)std::max<int64_t>(1, std::min<int64_t>(8, (512 + per_block - 1) / per_block));
    if (const char* e = getenv("GMMHMM_LOGLIK_BPW")) bpw = std::max(1, std::min(16, atoi(e)));   // tuning knob (blocks of 32 frames per wave)
    if (subset) bpw = std::max(bpw, 2);   // the block table lives in the multi-block instantiation
    const unsigned grid = (unsigned)((n_blocks + bpw - 1) / bpw);
    const T* X = static_cast<const T*>(b->feats);
    T* out = static_cast<T*>(b->nll);
    // fp32 exponentials in the fp64 epilogue (opt-in, see tile_lse_fe): mixtures of >= 4 components
    const bool fe = sizeof(T) == 8 && (ctx->compat & 2) && M_pad >= 4;
#define GH_MF_LAUNCH(ks, mp)                                                                                             \
    do {                                                                                                                 \
        if (fe && sizeof(T) == 8 && mp >= 4) {                                                                           \
            if (bpw > 1)                                                                                                 \
                hipLaunchKernelGGL((loglik_mfma_kernel<T, ks, mp, true, (sizeof(T) == 8 && mp >= 4)>), dim3(grid), dim3(64), lds, ctx->stream, X, N, g->D, \
                                   Apk, Cpk, n_tiles, S, M_pad, chunk_tiles, tables, tab_off, out, bpw, d_blk, n_blk, cen, inf_below); \
            else                                                                                                         \
                hipLaunchKernelGGL((loglik_mfma_kernel<T, ks, mp, false, (sizeof(T) == 8 && mp >= 4)>), dim3(grid), dim3(64), lds, ctx->stream, X, N, g->D, \
                                   Apk, Cpk, n_tiles, S, M_pad, chunk_tiles, tables, tab_off, out, bpw, d_blk, n_blk, cen, inf_below); \
        } else if (bpw > 1)                                                                                                     \
            hipLaunchKernelGGL((loglik_mfma_kernel<T, ks, mp, true>), dim3(grid), dim3(64), lds, ctx->stream, X, N, g->D, \
                               Apk, Cpk, n_tiles, S, M_pad, chunk_tiles, tables, tab_off, out, bpw, d_blk, n_blk, cen, inf_below); \
        else                                                                                                             \
            hipLaunchKernelGGL((loglik_mfma_kernel<T, ks, mp, false>), dim3(grid), dim3(64), lds, ctx->stream, X, N, g->D, \
                               Apk, Cpk, n_tiles, S, M_pad, chunk_tiles, tables, tab_off, out, bpw, d_blk, n_blk, cen, inf_below); \
    } while (0)
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
        GH_MF_CASE(24)
        GH_MF_CASE(32)
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
int gh_launch_loglik_mfma(gh_ctx* ctx, const gh_gmm* g, gh_batch* b, const int32_t* st_lo, const int32_t* st_hi,
                          const int64_t* rng_off, const gh_loglik_plan* plan) {
    if (!g->dApk64) return 1;
    // whose likelihoods the batch's matrix holds from here on (a subset launch overwrites the rows of its ranges; a
    // caller that mixes models across ranges loses the stamp's meaning -- the consumer, gh_bw_accumulate with M > 8,
    // documents that the E-step's likelihoods must come from the model it is given)
    b->nll_serial = g->serial;
    if (b->dtype == GH_F64) return launch_mfma_t<double>(ctx, g, b, g->dApk64, g->dCpk64, st_lo, st_hi, nullptr, rng_off, plan);
    return launch_mfma_t<float>(ctx, g, b, g->dApk32, g->dCpk32, st_lo, st_hi, g->dCen32, rng_off, plan);
}

// a block table that outlives the call (own allocation): 1 = shape / ranges not covered by the subset kernel
int gh_loglik_plan_build(gh_ctx* ctx, const gh_gmm* g, const gh_batch* b, const int32_t* st_lo, const int32_t* st_hi,
                         const int64_t* rng_off, gh_loglik_plan* out, const uint8_t* include) {
    out->d_blk = nullptr; out->n_blk = 0; out->max_tiles = 1;
    if (!g->dApk64) return 1;
    const int KS = g->KP / 2;
    if (KS != 2 && KS != 4 && KS != 8 && KS != 12 && KS != 20 && KS != 24 && KS != 32) return 1;
    int chunk_tiles, SC;
    chunk_shape(g, &chunk_tiles, &SC);
    std::vector<gh_loglik_blk> tabv;
    int rc = build_blk_table(g, b, SC, st_lo, st_hi, rng_off, tabv, &out->max_tiles, include);
    if (rc) return rc;
    out->n_blk = (int64_t)tabv.size();
    if (tabv.empty()) return GH_OK;
    GH_HIP(hipMalloc(&out->d_blk, tabv.size() * sizeof(gh_loglik_blk)));
    GH_HIP(hipMemcpy(out->d_blk, tabv.data(), tabv.size() * sizeof(gh_loglik_blk), hipMemcpyHostToDevice));
    return GH_OK;
}

void gh_loglik_plan_free(gh_loglik_plan* p) {
    if (p && p->d_blk) hipFree(p->d_blk);
    if (p) { p->d_blk = nullptr; p->n_blk = 0; }
}
