// Batched diagonal-covariance GMM negative log-likelihood (reference: GMM.evaluate,
// sr/recognition/hmm_state.py:114-120, evaluated in the log domain).
//
//   nll[n,s] = -logsumexp_m ( C[s,m] + sum_d ( A[s,m,d] x_nd^2 + B[s,m,d] x_nd ) )
//
// gfx950 design notes
//   * one frame per lane, one wave per workgroup: the frame (x and x^2) lives in
//     VGPRs for the whole kernel, staged once through LDS with coalesced 16-byte
//     global loads;
//   * Gaussian parameters are wave-uniform, so they are fetched with scalar loads
//     (s_load_dwordx8/x16 through the scalar cache, served by L2) and feed the
//     v_fma straight from SGPRs -- no LDS traffic and no VGPRs for parameters;
//   * four Gaussians are accumulated at once to give four independent fma chains;
//   * online log-sum-exp per state, one exp per component.
#include "gh_internal.h"

namespace {

template <typename T> __device__ __forceinline__ T gh_exp(T x);
template <> __device__ __forceinline__ float gh_exp<float>(float x) { return __expf(x); }
template <> __device__ __forceinline__ double gh_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T gh_log(T x);
template <> __device__ __forceinline__ float gh_log<float>(float x) { return __logf(x); }
template <> __device__ __forceinline__ double gh_log<double>(double x) { return log(x); }

template <typename T>
__device__ __forceinline__ void lse_push(T ll, T& mx, T& sm) {
    // running (max, sum of exp(. - max)).  A NaN component (NaN parameters) poisons the state,
    // like the reference's linear-domain sum does; a -inf component (w == 0) adds nothing.
    if (ll != ll) { sm = T(NAN); return; }
    if (ll == -INFINITY) return;
    T d = ll - mx;
    if (d > T(0)) {
        sm = sm * gh_exp<T>(-d) + T(1);
        mx = ll;
    } else {
        sm += gh_exp<T>(d);
    }
}

// inf_below: compat mode (gh_ctx_set_compat bit 0) -- once the largest term's logarithm is below ln 2^-1075 the
// reference's linear-domain sum is 0 and the state costs +inf (hmm_state.py:114-120); -inf otherwise
template <typename T>
__device__ __forceinline__ T lse_finish(T mx, T sm, T inf_below) {
    if (sm != sm) return T(NAN);
    if (mx < inf_below) return T(INFINITY);
    return (sm > T(0)) ? -(mx + gh_log<T>(sm)) : T(INFINITY);
}

// KP = padded feature length (multiple of 4), compile-time so x/x^2 stay in registers.
template <typename T, int KP>
__global__ __launch_bounds__(64) void loglik_kernel(const T* __restrict__ X, int64_t N, int D,
                                                    const T* __restrict__ A, const T* __restrict__ B,
                                                    const T* __restrict__ C, int S, int M,
                                                    T* __restrict__ out, const T* __restrict__ cen, T inf_below) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);  // [64][D] frames of this wave, row-major
    const int lane = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int64_t nrows = (N - n0 < 64) ? (N - n0) : 64;
    const int64_t nelem = nrows * D;
    const T* src = X + n0 * D;
    // coalesced stage: consecutive lanes read consecutive elements
    for (int64_t i = lane; i < nelem; i += 64) tile[i] = src[i];
    __syncthreads();

    T x[KP], x2[KP];
    const int row = (lane < nrows) ? lane : 0;
#pragma unroll
    for (int d = 0; d < KP; ++d) {
        T v = (d < D) ? tile[row * D + d] : T(0);
        if (cen) v -= cen[d];   // fp32: operands are packed for centred features (gh_internal.h, dCen32)
        x[d] = v;
        x2[d] = v * v;
    }

    const int64_t n = n0 + lane;
    for (int s = 0; s < S; ++s) {
        T mx = -INFINITY, sm = T(0);
        int m = 0;
        for (; m + 4 <= M; m += 4) {
            const int g = s * M + m;
            const T* a0 = A + (int64_t)g * KP;
            const T* b0 = B + (int64_t)g * KP;
            T acc0 = C[g], acc1 = C[g + 1], acc2 = C[g + 2], acc3 = C[g + 3];
#pragma unroll
            for (int d = 0; d < KP; ++d) {
                acc0 = fma(a0[d], x2[d], acc0);
                acc1 = fma(a0[KP + d], x2[d], acc1);
                acc2 = fma(a0[2 * KP + d], x2[d], acc2);
                acc3 = fma(a0[3 * KP + d], x2[d], acc3);
                acc0 = fma(b0[d], x[d], acc0);
                acc1 = fma(b0[KP + d], x[d], acc1);
                acc2 = fma(b0[2 * KP + d], x[d], acc2);
                acc3 = fma(b0[3 * KP + d], x[d], acc3);
            }
            lse_push(acc0, mx, sm);
            lse_push(acc1, mx, sm);
            lse_push(acc2, mx, sm);
            lse_push(acc3, mx, sm);
        }
        for (; m < M; ++m) {
            const int g = s * M + m;
            const T* a0 = A + (int64_t)g * KP;
            const T* b0 = B + (int64_t)g * KP;
            T acc0 = C[g];
#pragma unroll
            for (int d = 0; d < KP; ++d) {
                acc0 = fma(a0[d], x2[d], acc0);
                acc0 = fma(b0[d], x[d], acc0);
            }
            lse_push(acc0, mx, sm);
        }
        T nll = lse_finish<T>(mx, sm, inf_below);
        if (lane < nrows) out[n * S + s] = nll;
    }
}

// Fallback for any feature length: x re-read from the LDS tile for every Gaussian.
template <typename T>
__global__ __launch_bounds__(64) void loglik_kernel_any(const T* __restrict__ X, int64_t N, int D, int KP,
                                                        const T* __restrict__ A, const T* __restrict__ B,
                                                        const T* __restrict__ C, int S, int M,
                                                        T* __restrict__ out, const T* __restrict__ cen, T inf_below) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* tile = reinterpret_cast<T*>(smem_raw);
    const int lane = threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * 64;
    const int64_t nrows = (N - n0 < 64) ? (N - n0) : 64;
    const int64_t nelem = nrows * D;
    const T* src = X + n0 * D;
    for (int64_t i = lane; i < nelem; i += 64) tile[i] = src[i] - (cen ? cen[i % D] : T(0));
    __syncthreads();
    const int row = (lane < nrows) ? lane : 0;
    const T* xr = tile + row * D;
    const int64_t n = n0 + lane;
    for (int s = 0; s < S; ++s) {
        T mx = -INFINITY, sm = T(0);
        for (int m = 0; m < M; ++m) {
            const int g = s * M + m;
            const T* a0 = A + (int64_t)g * KP;
            const T* b0 = B + (int64_t)g * KP;
            T acc = C[g];
            for (int d = 0; d < D; ++d) {
                T v = xr[d];
                acc = fma(a0[d], v * v, acc);
                acc = fma(b0[d], v, acc);
            }
            lse_push(acc, mx, sm);
        }
        T nll = lse_finish<T>(mx, sm, inf_below);
        if (lane < nrows) out[n * S + s] = nll;
    }
}

template <typename T>
int launch_t(gh_ctx* ctx, const gh_gmm* g, gh_batch* b, const T* A, const T* B, const T* C, const T* cen) {
    const int64_t N = b->N;
    if (N == 0) return GH_OK;
    const unsigned grid = (unsigned)((N + 63) / 64);
    const size_t lds = (size_t)64 * g->D * sizeof(T);
    const T* X = static_cast<const T*>(b->feats);
    T* out = static_cast<T*>(b->nll);
    const T inf_below = (ctx->compat & 1) ? (T)-745.1332191019412 : (T)-INFINITY;   // (natural-log domain here)
#define GH_LL_CASE(kp)                                                                              \
    case kp:                                                                                        \
        hipLaunchKernelGGL((loglik_kernel<T, kp>), dim3(grid), dim3(64), lds, ctx->stream, X, N, g->D, \
                           A, B, C, g->S, g->M, out, cen, inf_below);                               \
        break;
    switch (g->KP) {
        GH_LL_CASE(4)
        GH_LL_CASE(8)
        GH_LL_CASE(16)
        GH_LL_CASE(24)
        GH_LL_CASE(40)
        default:
            hipLaunchKernelGGL((loglik_kernel_any<T>), dim3(grid), dim3(64), lds, ctx->stream, X, N, g->D,
                               g->KP, A, B, C, g->S, g->M, out, cen, inf_below);
    }
#undef GH_LL_CASE
    GH_HIP(hipGetLastError());
    return GH_OK;
}

}  // namespace

int gh_launch_loglik(gh_ctx* ctx, const gh_gmm* g, gh_batch* b) {
    b->nll_serial = g->serial;
    if (b->dtype == GH_F64) return launch_t<double>(ctx, g, b, g->dA64, g->dB64, g->dC64, nullptr);
    return launch_t<float>(ctx, g, b, g->dA32, g->dB32, g->dC32, g->dCen32);
}
