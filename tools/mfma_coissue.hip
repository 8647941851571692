// Micro-benchmark: do fp64 (fp32) VALU FMAs overlap with fp64 (fp32) MFMAs on one SIMD, or do
// they compete for the same pipeline?  One kernel, NV independent VALU fma per MFMA, in one wave
// (program-order interleave) at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NM, int NV>
__global__ void k64(double* out, int iters) {
    v4d acc[2] = {(v4d){0, 0, 0, 0}, (v4d){0, 0, 0, 0}};
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3 + i;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (NM) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j % 8] = fma(v[j % 8], 0.999999, 1e-9);
        }
    }
    double s = 0;
    for (int i = 0; i < 2; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NM, int NV>
__global__ void k32(float* out, int iters) {
    v4f acc[2] = {(v4f){0, 0, 0, 0}, (v4f){0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (NM) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j % 8] = fmaf(v[j % 8], 0.999999f, 1e-9f);
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double timeit(F launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 3;
}
#define RUN64(NM, NV) timeit([&] { hipLaunchKernelGGL((k64<NM, NV>), dim3(grid), dim3(64), 0, 0, (double*)buf, iters); })
#define RUN32(NM, NV) timeit([&] { hipLaunchKernelGGL((k32<NM, NV>), dim3(grid), dim3(64), 0, 0, (float*)buf, iters); })
int main() {
    void* buf; (void)hipMalloc(&buf, 64 << 20);
    const int iters = 10000;
    for (int wps : {1, 2}) {
        const int grid = 256 * 4 * wps;
        printf("waves/SIMD %d   (ms; per iteration: 2 MFMA + 2*NV VALU fma)\n", wps);
        printf("  f64: mfma only %.3f | valu only NV=4 %.3f NV=8 %.3f NV=16 %.3f | both NV=4 %.3f NV=8 %.3f NV=16 %.3f\n",
               RUN64(1, 0), RUN64(0, 4), RUN64(0, 8), RUN64(0, 16), RUN64(1, 4), RUN64(1, 8), RUN64(1, 16));
        printf("  f32: mfma only %.3f | valu only NV=4 %.3f NV=8 %.3f NV=16 %.3f | both NV=4 %.3f NV=8 %.3f NV=16 %.3f\n",
               RUN32(1, 0), RUN32(0, 4), RUN32(0, 8), RUN32(0, 16), RUN32(1, 4), RUN32(1, 8), RUN32(1, 16));
    }
    return 0;
}
