// Micro-benchmark + layout probe for v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4) on gfx950:
//  (1) which lane holds which (block, i, k) of A, (block, k, j) of B and (block, i, j) of D -- found by one-hot inputs;
//  (2) its sustained rate next to v_mfma_f64_16x16x4_f64 and a plain v_fma_f64 chain with an SGPR operand.
// The lock-step refit kernels (gh_lockstep_mfma.hip) are built on (1); their roofline is priced against (2).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_4x4.hip -o tools/bin/mfma_f64_4x4 && tools/bin/mfma_f64_4x4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void probe(double* out) {    // block = (la, lb): A one-hot at lane la, B one-hot at lane lb
    const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[(size_t)blockIdx.x * 64 + lane] = d;
}

// CBSZ = 2: the A values of block ABID are used by all four blocks (one-hot probe as above)
template <int ABID>
__global__ void probe_bcast(double* out) {
    const int la = blockIdx.x >> 6, lb = blockIdx.x & 63, lane = threadIdx.x;
    const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 2, ABID, 0);
    out[(size_t)blockIdx.x * 64 + lane] = d;
}

template <int NACC>
__global__ void k4(double* out, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k16(double* out, int iters) {
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
// NACC independent fma chains, the multiplier wave-uniform (an SGPR pair from a scalar load)
template <int NACC>
__global__ void kv(double* out, const double* __restrict__ p, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
        const double s0 = p[(it & 7)];      // uniform address -> s_load
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], s0, 1e-9);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
double timeit(F launch) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 5 * 1e-3;
}
int main(int argc, char** argv) {
    void* buf; (void)hipMalloc(&buf, 64 << 20);
    double* dp; (void)hipMalloc((void**)&dp, 64);
    double hp[8] = {0.999999, 0.999998, 0.999997, 0.999996, 0.999995, 0.999994, 0.999993, 0.999992};
    (void)hipMemcpy(dp, hp, 64, hipMemcpyHostToDevice);
    // ---- (1) layout ----
    hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, (double*)buf);
    std::vector<double> h(4096 * 64);
    (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
    // for every lane la: the set of lanes lb that pair with it (same block, same k) and the output lane
    printf("pairs (la, lb) -> ld with D[ld] = 1:\n");
    int npairs = 0;
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            int nz = 0, ld = -1;
            for (int l = 0; l < 64; ++l) if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) { ++nz; ld = l; }
            if (nz) { printf(" B%d->D%d%s", lb, ld, nz > 1 ? "(multi)" : ""); ++npairs; }
        }
        printf("\n");
    }
    printf("non-zero pairs: %d (expected 4 blocks x 4 i x 4 j x 4 k = 256)\n", npairs);
    // ---- (1b) cbsz:2 abid:t -- expected: A lane 16k + 4t + i pairs with B lanes 16k + 4b + j of EVERY block b -> D lane 16i + 4b + j
    for (int t = 0; t < 4; ++t) {
        if (t == 0) hipLaunchKernelGGL(probe_bcast<0>, dim3(4096), dim3(64), 0, 0, (double*)buf);
        if (t == 1) hipLaunchKernelGGL(probe_bcast<1>, dim3(4096), dim3(64), 0, 0, (double*)buf);
        if (t == 2) hipLaunchKernelGGL(probe_bcast<2>, dim3(4096), dim3(64), 0, 0, (double*)buf);
        if (t == 3) hipLaunchKernelGGL(probe_bcast<3>, dim3(4096), dim3(64), 0, 0, (double*)buf);
        (void)hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost);
        int good = 0, bad = 0;
        for (int la = 0; la < 64; ++la)
            for (int lb = 0; lb < 64; ++lb) {
                const int ka = la >> 4, ba = (la >> 2) & 3, ia = la & 3, kb = lb >> 4, bb = (lb >> 2) & 3, jb = lb & 3;
                const bool expect = ba == t && ka == kb;
                const int ld_exp = 16 * ia + 4 * bb + jb;
                for (int l = 0; l < 64; ++l) {
                    const bool nz = h[((size_t)la * 64 + lb) * 64 + l] != 0.0;
                    if (nz == (expect && l == ld_exp)) ++good; else ++bad;
                }
            }
        printf("cbsz:2 abid:%d -> %d entries as expected, %d not\n", t, good, bad);
        for (int la = 0; la < 64; ++la) {
            printf("  cbsz2 abid%d A lane %2d:", t, la);
            for (int lb = 0; lb < 64; ++lb)
                for (int l = 0; l < 64; ++l) if (h[((size_t)la * 64 + lb) * 64 + l] != 0.0) printf(" B%d->D%d", lb, l);
            printf("\n");
        }
    }
    // ---- (2) rates ----
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    for (int wpb : {64, 128, 256}) {
        for (int bpc : {4, 8}) {
            const int grid = 256 * bpc;
            const double waves = (double)grid * (wpb / 64);
            double t4 = timeit([&] { hipLaunchKernelGGL((k4<8>), dim3(grid), dim3(wpb), 0, 0, (double*)buf, iters); });
            double t16 = timeit([&] { hipLaunchKernelGGL((k16<4>), dim3(grid), dim3(wpb), 0, 0, (double*)buf, iters); });
            double tv = timeit([&] { hipLaunchKernelGGL((kv<8>), dim3(grid), dim3(wpb), 0, 0, (double*)buf, (const double*)dp, iters); });
            printf("threads/block %3d blocks/CU %d (waves/SIMD %.1f): 4x4x4_4b %.1f TF (%.1f cycles/instr at 2.4 GHz) | 16x16x4 %.1f TF | v_fma_f64 sgpr %.1f TF\n",
                   wpb, bpc, bpc * wpb / 64 / 4.0, 512.0 * 8 * iters * waves / t4 / 1e12,
                   t4 * 2.4e9 / (8.0 * iters * (bpc * wpb / 64 / 4.0)), 2048.0 * 4 * iters * waves / t16 / 1e12,
                   128.0 * 8 * iters * waves / tv / 1e12);
        }
    }
    return 0;
}
