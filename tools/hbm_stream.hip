// Micro-benchmark: what HBM delivers to plain streaming kernels on this chip -- read only, write only, and the 1 : 4
// read : write mix of the C1 likelihood kernel (104 B of features in, 400 B of likelihoods out per frame) -- to price
// the HBM-bound kernels against a measured ceiling per traffic mix rather than the 8 TB/s datasheet number only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v2d __attribute__((ext_vector_type(2)));
__global__ void k_read(const v2d* __restrict__ in, size_t n, double* sink) {
    v2d acc = {0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc.x + acc.y == 12345.678) sink[0] = acc.x;
}
__global__ void k_write(v2d* __restrict__ out, size_t n) {
    const v2d v = {1.0, 2.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = v;
}
// every thread block reads R 16-byte vectors and writes 4 R derived from them
__global__ void k_mix(const v2d* __restrict__ in, v2d* __restrict__ out, size_t n_in) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += (size_t)gridDim.x * blockDim.x) {
        const v2d v = in[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k * n_in + i] = v * (double)(k + 1);   // four coalesced output streams
    }
}
template <typename F> double timeit(F launch, int reps = 20) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps * 1e-3;
}
int main() {
    const size_t GB = (size_t)1 << 30, bytes = 4 * GB, n = bytes / 16;
    v2d *a, *b;
    double* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 8);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    for (int wg : {2048, 8192, 32768}) {
        const double tr = timeit([&] { hipLaunchKernelGGL(k_read, dim3(wg), dim3(256), 0, 0, a, n, sink); });
        const double tw = timeit([&] { hipLaunchKernelGGL(k_write, dim3(wg), dim3(256), 0, 0, b, n); });
        const double tm = timeit([&] { hipLaunchKernelGGL(k_mix, dim3(wg), dim3(256), 0, 0, a, b, n / 4); });
        printf("%6d workgroups: read %.2f TB/s   write %.2f TB/s   1:4 read:write mix %.2f TB/s (4 GiB each, 16 B per lane)\n", wg,
               bytes / tr / 1e12, bytes / tw / 1e12, (bytes / 4 + bytes) / tm / 1e12);
    }
    const double tms = timeit([&] { hipMemsetAsync(b, 0, bytes, 0); });
    printf("hipMemsetAsync: %.2f TB/s\n", bytes / tms / 1e12);
    return 0;
}
