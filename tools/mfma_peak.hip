// Micro-benchmark: sustained rate of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 on this chip,
// to price the likelihood kernel against a measured ceiling rather than the datasheet only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k64(double* out, int iters) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k32(float* out, int iters) {
    v4f acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f){0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double timeit(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5 * 1e-3;
}
int main(int argc, char** argv) {
    void* buf; hipMalloc(&buf, 64 << 20);
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    for (int wpb : {64, 128, 256}) {       // waves per CU = blocks/CU * wpb/64
        for (int bpc : {4, 8}) {
            const int grid = 256 * bpc;
            double t2 = timeit([&] { hipLaunchKernelGGL((k64<2>), dim3(grid), dim3(wpb), 0, 0, (double*)buf, iters); });
            double t4 = timeit([&] { hipLaunchKernelGGL((k64<4>), dim3(grid), dim3(wpb), 0, 0, (double*)buf, iters); });
            double f2 = 2048.0 * 2 * iters * grid * (wpb / 64) / t2 / 1e12, f4 = 2048.0 * 4 * iters * grid * (wpb / 64) / t4 / 1e12;
            double s4 = timeit([&] { hipLaunchKernelGGL((k32<4>), dim3(grid), dim3(wpb), 0, 0, (float*)buf, iters); });
            double g4 = 2048.0 * 4 * iters * grid * (wpb / 64) / s4 / 1e12;
            printf("threads/block %3d blocks/CU %d (waves/SIMD %.1f): f64 2acc %.1f TF  4acc %.1f TF | f32 4acc %.1f TF\n", wpb, bpc,
                   bpc * wpb / 64 / 4.0, f2, f4, g4);
        }
    }
    return 0;
}
