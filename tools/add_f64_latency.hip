// Latency of a dependent v_add_f64 on gfx950: the floor per row of numpy-ordered row sums (fit_segsum_wide_kernel,
// kmeans_rowsum_kernel: one chain per dimension).   hipcc --offload-arch=gfx950 -O3 tools/add_f64_latency.hip -o /tmp/addlat
// MI355X: 8.3 cycles per dependent add.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void chain(const double* __restrict__ in, double* __restrict__ out, long long* cyc, int n) {
    double x = in[threadIdx.x], acc = 0.0;
    long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) asm volatile("v_add_f64 %0, %0, %1" : "+v"(acc) : "v"(x));
    }
    long long t1 = clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    double *in, *out;
    long long* cyc;
    if (hipMalloc(&in, 1024 * 8) != hipSuccess || hipMalloc(&out, 64 * 8) != hipSuccess || hipMalloc(&cyc, 8) != hipSuccess) return 1;
    (void)hipMemset(in, 0, 1024 * 8);
    const int n = 200;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, in, out, cyc, n);
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    long long h = 0;
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%lld cycles for %d dependent v_add_f64 = %.2f cycles per add\n", h, n * 64, (double)h / (n * 64));
    return 0;
}
