// Small fp64 helper kernels on host-resident inputs (training-side building blocks):
//   * per-component weighted log densities of one state (hmm_state.py:114-116)
//   * frame x template distance matrices: Euclidean norm (default dist_fun,
//     kmeans.py:111,167) and diagonal-Gaussian negative log-likelihood
//     `mahalanobis` (hmm_state.py:48-58)
// One output element per lane, frames along lanes so the [K,N] / [N,M] stores coalesce.
#include "gh_internal.h"
#include <cmath>

namespace {

__global__ void component_loglik_kernel(const double* __restrict__ X, int64_t N, int D, int M,
                                        const double* __restrict__ mean, const double* __restrict__ ivar,
                                        const double* __restrict__ logc, double* __restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double* x = X + n * D;
    for (int m = 0; m < M; ++m) {
        double q = 0;
        for (int d = 0; d < D; ++d) {
            const double t = x[d] - mean[m * D + d];
            q = fma(t * ivar[m * D + d], t, q);
        }
        out[n * M + m] = logc[m] - 0.5 * q;
    }
}

// var_rows: 0 -> Euclidean norm, 1 -> shared variance, K -> per-template variance
__global__ void distance_kernel(const double* __restrict__ X, int64_t N, int K, int D,
                                const double* __restrict__ Y, const double* __restrict__ var, int var_rows,
                                const double* __restrict__ logdet /*[var_rows]*/, double* __restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double* x = X + n * D;
    for (int k = 0; k < K; ++k) {
        const double* y = Y + (int64_t)k * D;
        double q = 0;
        if (var_rows == 0) {
            for (int d = 0; d < D; ++d) {
                const double t = x[d] - y[d];
                q = fma(t, t, q);
            }
            out[(int64_t)k * N + n] = sqrt(q);
        } else {
            const double* v = var + (var_rows == 1 ? 0 : (int64_t)k * D);
            for (int d = 0; d < D; ++d) {
                const double t = x[d] - y[d];
                q += t / v[d] * t;  // m / variance * m (hmm_state.py:58)
            }
            out[(int64_t)k * N + n] = logdet[var_rows == 1 ? 0 : k] + 0.5 * q;
        }
    }
}

}  // namespace

extern "C" int gh_component_loglik(gh_ctx* ctx, const gh_gmm* g, int state, int64_t N, const double* x_host,
                                   double* out_host) {
    GH_REQUIRE(ctx && g && x_host && out_host, "gh_component_loglik: NULL argument");
    GH_REQUIRE(state >= 0 && state < g->S, "gh_component_loglik: state %d out of range", state);
    if (N <= 0) return GH_OK;
    GH_HIP(hipSetDevice(ctx->device));
    const int D = g->D, M = g->M;
    void* base;
    const size_t xb = ((size_t)N * D * 8 + 255) & ~size_t(255);
    int rc = gh_scratch(ctx, xb + (size_t)N * M * 8, &base);
    if (rc) return rc;
    double* dx = (double*)base;
    double* dout = (double*)((char*)base + xb);
    hipStream_t st = ctx->stream;
    GH_HIP(hipMemcpyAsync(dx, x_host, (size_t)N * D * 8, hipMemcpyHostToDevice, st));
    const size_t go = (size_t)state * M;
    hipLaunchKernelGGL(component_loglik_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dx, N, D, M,
                       g->dMean + go * D, g->dIvar + go * D, g->dLogc + go, dout);
    GH_HIP(hipGetLastError());
    GH_HIP(hipMemcpyAsync(out_host, dout, (size_t)N * M * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}

extern "C" int gh_distance_matrix(gh_ctx* ctx, int64_t N, int K, int D, const double* x_host,
                                  const double* y_host, const double* var_host, int var_rows,
                                  double* out_host) {
    GH_REQUIRE(ctx && x_host && y_host && out_host, "gh_distance_matrix: NULL argument");
    GH_REQUIRE(K > 0 && D > 0, "gh_distance_matrix: K=%d D=%d", K, D);
    GH_REQUIRE((var_host == nullptr) == (var_rows == 0), "gh_distance_matrix: var / var_rows mismatch");
    GH_REQUIRE(var_rows == 0 || var_rows == 1 || var_rows == K, "gh_distance_matrix: var_rows=%d", var_rows);
    if (N <= 0) return GH_OK;
    GH_HIP(hipSetDevice(ctx->device));
    // 0.5*log((2 pi)^D * prod(var)) exactly as hmm_state.py:58 evaluates it (linear-domain product)
    std::vector<double> logdet(var_rows > 0 ? var_rows : 1, 0.0);
    for (int r = 0; r < var_rows; ++r) {
        double prod = 1.0;
        for (int d = 0; d < D; ++d) prod *= var_host[(size_t)r * D + d];
        logdet[r] = 0.5 * std::log(std::pow(2.0 * M_PI, D) * prod);
    }
    auto al = [](size_t b) { return (b + 255) & ~size_t(255); };
    const size_t bx = al((size_t)N * D * 8), by = al((size_t)K * D * 8), bv = al((size_t)std::max(var_rows, 1) * D * 8),
                 bl = al(logdet.size() * 8);
    void* base;
    int rc = gh_scratch(ctx, bx + by + bv + bl + (size_t)K * N * 8, &base);
    if (rc) return rc;
    char* p = (char*)base;
    double *dx = (double*)p, *dy = (double*)(p + bx), *dv = (double*)(p + bx + by), *dl = (double*)(p + bx + by + bv),
           *dout = (double*)(p + bx + by + bv + bl);
    hipStream_t st = ctx->stream;
    GH_HIP(hipMemcpyAsync(dx, x_host, (size_t)N * D * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(dy, y_host, (size_t)K * D * 8, hipMemcpyHostToDevice, st));
    if (var_rows) GH_HIP(hipMemcpyAsync(dv, var_host, (size_t)var_rows * D * 8, hipMemcpyHostToDevice, st));
    GH_HIP(hipMemcpyAsync(dl, logdet.data(), logdet.size() * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(distance_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dx, N, K, D, dy, dv,
                       var_rows, dl, dout);
    GH_HIP(hipGetLastError());
    GH_HIP(hipMemcpyAsync(out_host, dout, (size_t)K * N * 8, hipMemcpyDeviceToHost, st));
    GH_HIP(hipStreamSynchronize(st));
    return GH_OK;
}
