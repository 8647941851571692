// Micro-benchmark: which ingredient of the likelihood kernel's tile loop costs MFMA issue rate?
// V0 same operands; V1 40 distinct B fragments + 10-slot A ring (no loads); V2 = V1 + ring refills from
// an L2-resident operand stream; V3 = V2 + accumulator re-initialisation per tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v4d __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)
template <int VAR>
__global__ __launch_bounds__(64) void k(const double* __restrict__ A, const double* __restrict__ X, double* out, int tiles, int n_tiles_mod) {
    constexpr int KS = 20, R = 10;
    const int lane = threadIdx.x;
    double b[2][KS];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < KS; ++j) b[c][j] = X[(c * KS + j) * 64 + lane];
    double ring[R];
#pragma unroll
    for (int j = 0; j < R; ++j) ring[j] = A[j * 64 + lane];
    v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, sink = {0, 0, 0, 0};
    const long long c0 = clock64(), w0 = wall_clock64();
    int t = 0;
    for (int it = 0; it < tiles; ++it) {
        const double* a_cur = A + (size_t)t * KS * 64 + lane;
        const int tn = (t + 1 < n_tiles_mod) ? t + 1 : 0;
        const double* a_nxt = A + (size_t)tn * KS * 64 + lane;
        if (VAR >= 3) { sink += acc0 + acc1; acc0 = (v4d){1, 2, 3, 4}; acc1 = acc0; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (VAR == 0) {
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[0], b[0][0], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[0], b[1][0], acc1, 0, 0, 0);
            } else {
                const double a = ring[ks % R];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[0][ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[1][ks], acc1, 0, 0, 0);
                if (VAR >= 2) ring[ks % R] = (ks < R) ? a_cur[(ks + R) * 64] : a_nxt[(ks - R) * 64];
            }
        }
        t = tn;
    }
    sink += acc0 + acc1;
    out[blockIdx.x * 64 + lane] = sink[0] + sink[1] + sink[2] + sink[3];
    if (blockIdx.x == 0 && lane == 0) { out[256 * 16 * 64] = (double)(clock64() - c0); out[256 * 16 * 64 + 1] = (double)(wall_clock64() - w0); }
}
template <int VAR> void run(const double* A, const double* X, double* out, int wpc, int tiles) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * wpc;
    hipLaunchKernelGGL((k<VAR>), dim3(grid), dim3(64), 0, 0, A, X, out, tiles, 25);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<VAR>), dim3(grid), dim3(64), 0, 0, A, X, out, tiles, 25);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double tf = 2048.0 * 40 * tiles * grid / (ms / 3 * 1e-3) / 1e12;
    double h[2];
    CK(hipMemcpy(h, out + 256 * 16 * 64, 16, hipMemcpyDeviceToHost));
    printf("variant %d waves/SIMD %.0f: %.1f TF (%.3f ms)  clock64/wall_clock64 -> %.3f GHz\n", VAR, wpc / 4.0, tf, ms / 3, h[0] / h[1] * 0.1);
}
int main(int argc, char** argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 2000;
    double *A, *X, *out;
    CK(hipMalloc(&A, 26 * 20 * 64 * 8)); CK(hipMalloc(&X, 40 * 64 * 8)); CK(hipMalloc(&out, 256 * 16 * 64 * 8 + 16));
    // operand data: zeros (minimal switching activity) vs random values (what a real workload feeds the
    // matrix cores) -- the sustained clock, and so the achievable rate, depends on it
    for (int random : {0, 1}) {
        const size_t na = 26 * 20 * 64, nx = 40 * 64;
        double* h = (double*)malloc((na + nx) * 8);
        for (size_t i = 0; i < na + nx; ++i) h[i] = random ? (rand() / (double)RAND_MAX - 0.5) * 4.0 : 0.0;
        CK(hipMemcpy(A, h, na * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(X, h + na, nx * 8, hipMemcpyHostToDevice));
        free(h);
        printf("-- operands: %s\n", random ? "random in [-2,2]" : "all zero");
        for (int wpc : {4, 8}) {
            run<0>(A, X, out, wpc, tiles); run<1>(A, X, out, wpc, tiles); run<2>(A, X, out, wpc, tiles); run<3>(A, X, out, wpc, tiles);
        }
    }
    return 0;
}
