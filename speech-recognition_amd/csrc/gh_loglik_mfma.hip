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

template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp<float>(float x) { return __expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T t_log(T x);
template <> __device__ __forceinline__ float t_log<float>(float x) { return __logf(x); }
template <> __device__ __forceinline__ double t_log<double>(double x) { return log(x); }

template <typename T> __device__ __forceinline__ T xor_lane(T v, int mask);
template <> __device__ __forceinline__ float xor_lane<float>(float v, int mask) { return __shfl_xor(v, mask, 64); }
template <> __device__ __forceinline__ double xor_lane<double>(double v, int mask) { return __shfl_xor(v, mask, 64); }

template <typename T> __device__ __forceinline__ T nll_of(T mx, T sm) {
    // -(mx + log sm); NaN parameters poison the state, an all-(-inf) state costs +inf
    if (sm != sm) return T(NAN);
    return (sm > T(0)) ? -(mx + t_log<T>(sm)) : T(INFINITY);
}

// max and sum-of-exp over the 4 registers of one lane, then over `width` lane groups (1, 2 or 4)
template <typename T, typename V>
__device__ __forceinline__ void tile_lse(const V& a, int width, T& mx, T& sm) {
    T m = fmax(fmax(a[0], a[1]), fmax(a[2], a[3]));
    if (width >= 2) m = fmax(m, xor_lane<T>(m, 16));
    if (width >= 4) m = fmax(m, xor_lane<T>(m, 32));
    const T ms = (m == -INFINITY) ? T(0) : m;  // all components off: exp(-inf - 0) = 0
    T e = t_exp<T>(a[0] - ms) + t_exp<T>(a[1] - ms) + t_exp<T>(a[2] - ms) + t_exp<T>(a[3] - ms);
    if (width >= 2) e += xor_lane<T>(e, 16);
    if (width >= 4) e += xor_lane<T>(e, 32);
    mx = m;
    sm = e;
}

// KS = number of k-steps (K = 4*KS = 2*KP).
template <typename T, int KS>
__global__ __launch_bounds__(64) void loglik_mfma_kernel(const T* __restrict__ X, int64_t N, int D,
                                                         const T* __restrict__ Apk, const T* __restrict__ Cpk,
                                                         int n_tiles, int S, int M_pad, int chunk_tiles,
                                                         T* __restrict__ out) {
    typedef typename Acc<T>::type V;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    const int lane = threadIdx.x;
    const int f = lane & 15, q = lane >> 4;
    const int64_t n0 = (int64_t)blockIdx.x * 32;
    const int nrows = (int)((N - n0 < 32) ? (N - n0) : 32);

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

    // ---- stream the Gaussian tiles -----------------------------------------------------
    T a_cur[KS], a_nxt[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a_cur[ks] = Apk[(int64_t)ks * 64 + lane];

    const int states_per_tile = (M_pad <= 16) ? 16 / M_pad : 0;
    const int tiles_per_state = (M_pad <= 16) ? 1 : M_pad / 16;
    T run_mx[2] = {-INFINITY, -INFINITY}, run_sm[2] = {T(0), T(0)};  // M_pad > 16: across tiles
    int chunk_s0 = 0;  // first state held in the LDS output tile
    const int SC = (M_pad <= 16) ? chunk_tiles * states_per_tile : chunk_tiles / tiles_per_state;
    const int RS = (S <= SC) ? S : SC;  // LDS row stride: whole matrix rows when they fit one chunk

    for (int t = 0; t < n_tiles; ++t) {
        if (t + 1 < n_tiles) {
            const T* nx = Apk + (int64_t)(t + 1) * KS * 64 + lane;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a_nxt[ks] = nx[ks * 64];
        }
        V acc0, acc1;
        {
            const T* cp = Cpk + t * 16 + 4 * q;
#pragma unroll
            for (int r = 0; r < 4; ++r) { acc0[r] = cp[r]; acc1[r] = cp[r]; }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            acc0 = Acc<T>::mfma(a_cur[ks], b[0][ks], acc0);
            acc1 = Acc<T>::mfma(a_cur[ks], b[1][ks], acc1);
        }
        // ---- epilogue: log-sum-exp over the mixture, into the LDS output tile ----------
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const V& acc = c ? acc1 : acc0;
            T* orow = lds + (16 * c + f) * RS;
            if (M_pad == 1) {
                const int s = 16 * t + 4 * q - chunk_s0;
#pragma unroll
                for (int r = 0; r < 4; ++r) if (s + r + chunk_s0 < S) orow[s + r] = -acc[r];
            } else if (M_pad == 2) {
                const int s = 8 * t + 2 * q - chunk_s0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const T x0 = acc[2 * h], x1 = acc[2 * h + 1];
                    const T m = fmax(x0, x1);
                    const T ms = (m == -INFINITY) ? T(0) : m;
                    const T e = t_exp<T>(x0 - ms) + t_exp<T>(x1 - ms);
                    if (s + h + chunk_s0 < S) orow[s + h] = nll_of<T>(m, e);
                }
            } else if (M_pad <= 16) {
                const int width = M_pad / 4;  // lane groups per state: 1, 2 or 4
                T mx, sm;
                tile_lse<T, V>(acc, width, mx, sm);
                const int s = states_per_tile * t + q / width;
                if ((q & (width - 1)) == 0 && s < S) orow[s - chunk_s0] = nll_of<T>(mx, sm);
            } else {
                T mx, sm;
                tile_lse<T, V>(acc, 4, mx, sm);
                // merge this tile into the running (max, sum) of the state
                const T m = fmax(run_mx[c], mx);
                const T ms = (m == -INFINITY) ? T(0) : m;
                run_sm[c] = run_sm[c] * t_exp<T>(run_mx[c] - ms) + sm * t_exp<T>(mx - ms);
                run_mx[c] = m;
                if ((t + 1) % tiles_per_state == 0) {
                    const int s = t / tiles_per_state;
                    if (q == 0 && s < S) orow[s - chunk_s0] = nll_of<T>(run_mx[c], run_sm[c]);
                    run_mx[c] = -INFINITY;
                    run_sm[c] = T(0);
                }
            }
        }
        // ---- flush the LDS tile when its state chunk is complete -----------------------
        if ((t + 1) % chunk_tiles == 0 || t + 1 == n_tiles) {
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
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a_cur[ks] = a_nxt[ks];
    }
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
    const size_t lds = (size_t)32 * std::max(std::min(SC, S), g->D) * sizeof(T);
    const unsigned grid = (unsigned)((N + 31) / 32);
    const T* X = static_cast<const T*>(b->feats);
    T* out = static_cast<T*>(b->nll);
#define GH_MF_CASE(ks)                                                                                  \
    case ks:                                                                                            \
        hipLaunchKernelGGL((loglik_mfma_kernel<T, ks>), dim3(grid), dim3(64), lds, ctx->stream, X, N, g->D, \
                           Apk, Cpk, n_tiles, S, M_pad, chunk_tiles, out);                              \
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
