// Micro-benchmark: what HBM delivers to plain streaming kernels on this chip -- read only, write only, and the 1 : 4
// read : write mix of the C1 likelihood kernel (104 B of features in, 400 B of likelihoods out per frame) -- to price
// the HBM-bound kernels against a measured ceiling per traffic mix rather than the 8 TB/s datasheet number only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
// calibration kernels for the PMC byte counters (FETCH_SIZE is only documented for 16 B per lane): the access shapes of the
// kernels whose traffic is quoted -- 8 B per lane, contiguous (the fused sweep's and the refit's slab loads), one pass
__global__ void k_read8(const double* __restrict__ in, size_t n, double* sink) {
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += in[i];
    if (acc == 12345.678) sink[0] = acc;
}
// one wave reads one contiguous piece of `len` doubles with 64 x 8 B loads (a tile of frames), pieces back to back
__global__ void k_read8_tiles(const double* __restrict__ in, size_t n_tiles, int len, double* sink) {
    double acc = 0;
    const int lane = threadIdx.x & 63;
    for (size_t t = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); t < n_tiles; t += (size_t)gridDim.x * (blockDim.x >> 6))
        for (int e = lane; e < len; e += 64) acc += in[t * len + e];
    if (acc == 12345.678) sink[0] = acc;
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
int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "calib")) {
        // ONE launch of every calibration kernel over 2 GiB (beyond the 256 MiB Infinity Cache), for rocprofv3 --pmc:
        //   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o c -- tools/bin/hbm_stream calib
        const size_t bytes = (size_t)2 << 30;
        double *a, *sink;
        hipMalloc(&a, bytes); hipMalloc(&sink, 8);
        hipMemset(a, 0, bytes);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const v2d*)a, bytes / 16, sink);
        hipLaunchKernelGGL(k_read8, dim3(8192), dim3(256), 0, 0, (const double*)a, bytes / 8, sink);
        hipLaunchKernelGGL(k_read8_tiles, dim3(8192), dim3(256), 0, 0, (const double*)a, bytes / 8 / 416, 416, sink);   // 32 frames x 13
        hipLaunchKernelGGL(k_read8_tiles, dim3(8192), dim3(256), 0, 0, (const double*)a, bytes / 8 / 624, 624, sink);   // 16 frames x 39
        hipDeviceSynchronize();
        printf("calib: k_read (16 B per lane) %zu bytes, k_read8 (8 B per lane) %zu bytes, k_read8_tiles len 416: %zu bytes, len 624: %zu bytes\n",
               bytes, bytes, bytes / 8 / 416 * 416 * 8, bytes / 8 / 624 * 624 * 8);
        return 0;
    }
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
