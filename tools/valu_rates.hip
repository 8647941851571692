// Micro-benchmark: issue cost (cycles per wave64 instruction) of the fp64 VALU instructions the likelihood epilogue
// uses, to decide which ones are worth replacing.  One wave per SIMD would under-fill the pipe, so 4 waves per SIMD
// run 8 independent dependency chains each; cycles come from clock64.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %d line %d\n", (int)e_, __LINE__); exit(1); } } while (0)
constexpr int ITERS = 2000, CH = 8;
#define KERNEL(NAME, ...)                                                                   \
    __global__ void NAME(double* out, long long* cyc, double seed) {                         \
        double v[CH];                                                                        \
        for (int i = 0; i < CH; ++i) v[i] = seed + threadIdx.x * 1e-3 + i;                   \
        const long long t0 = clock64();                                                      \
        for (int it = 0; it < ITERS; ++it) {                                                 \
            _Pragma("unroll") for (int i = 0; i < CH; ++i) { double& x = v[i]; __VA_ARGS__; }       \
        }                                                                                    \
        const long long t1 = clock64();                                                      \
        double s = 0; for (int i = 0; i < CH; ++i) s += v[i];                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                      \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                     \
    }
KERNEL(k_fma, x = __builtin_fma(x, 0.999999, 1e-7))
KERNEL(k_add, x = x + 1e-7)
KERNEL(k_mul, x = x * 0.9999999)
KERNEL(k_max, asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(seed)))
KERNEL(k_rndne, asm volatile("v_rndne_f64 %0, %0" : "+v"(x)))
KERNEL(k_ldexp, asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x)); x = x * 0.5)   /* ldexp + mul: subtract k_mul */
KERNEL(k_cvt, { int n; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(x)); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x) : "v"(n)); })
KERNEL(k_frexp, asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(x)))
KERNEL(k_cmp_cnd, x = (x < seed) ? x + 1e-7 : x)        /* cmp + 2 cndmask + add */
KERNEL(k_mov, { unsigned lo = __double2loint(x); unsigned hi = __double2hiint(x); asm volatile("v_add_u32 %0, %0, 1" : "+v"(lo)); x = __hiloint2double(hi, lo); })
template <typename K> void run(const char* name, K kern, double* out, long long* cyc, int per_inst) {
    const int blocks = 256 * 16;   // 16 waves per CU = 4 per SIMD
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0);
    CK(hipDeviceSynchronize());
    long long h[64];
    CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < 64; ++i) avg += h[i]; avg /= 64;
    // 4 waves share a SIMD: per-wave wall cycles / (ITERS*CH*instr) * (1/4) = issue cycles per instruction
    printf("%-10s %7.2f cycles per wave64 instruction (x%d instr per statement)\n", name, avg / (double)(ITERS * CH) / 4.0 / per_inst, per_inst);
}
int main() {
    double* out; long long* cyc;
    CK(hipMalloc(&out, 256 * 16 * 64 * 8)); CK(hipMalloc(&cyc, 256 * 16 * 8));
    run("fma_f64", k_fma, out, cyc, 1); run("add_f64", k_add, out, cyc, 1); run("mul_f64", k_mul, out, cyc, 1);
    run("max_f64", k_max, out, cyc, 1); run("rndne_f64", k_rndne, out, cyc, 1); run("ldexp+mul", k_ldexp, out, cyc, 2);
    run("cvt i32<->f64", k_cvt, out, cyc, 2); run("frexp_mant", k_frexp, out, cyc, 1);
    run("cmp+cnd+add", k_cmp_cnd, out, cyc, 4); run("u32 add", k_mov, out, cyc, 1);
    return 0;
}
