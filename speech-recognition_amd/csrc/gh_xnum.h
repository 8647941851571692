#pragma once
// Probabilities with an extended exponent for the sum-product kernels: value = f * 2^e, f a double, e an int32.
// A forward-backward in the log domain pays one exp per term and one log per sum (fp64: ~50 and ~70 instructions); in
// this form a term is a multiply and an integer add, a sum aligns the exponents with v_ldexp_f64 and adds, and the ONLY
// transcendental left is the split exponential of the emission cost -- one per state and column.  Unlike the textbook
// per-column scaling nothing is ever flushed relative to a column maximum: every value carries its own exponent, so a
// state 2^-5000 below its neighbour keeps all 53 bits, exactly as it would as a logarithm.
#include <hip/hip_runtime.h>
#include <cmath>

struct xnum {
    double f;
    int e;
};
#define XN_ZERO_E (-(1 << 28))   // exponent of 0 (f = 0): loses every max(), survives a few multiplications without wrapping

__device__ __forceinline__ xnum xn_zero() { return xnum{0.0, XN_ZERO_E}; }
__device__ __forceinline__ xnum xn_one() { return xnum{1.0, 0}; }

// exp(-c) as (f in [0.70, 1.42], e): c = +inf (and anything beyond 1e9) gives 0, NaN stays NaN
__device__ __forceinline__ xnum xn_exp_neg(double c) {
    const double x = -c;
    const bool tiny = !(x >= -1.0e9);                         // also catches NaN (restored below)
    const double xc = tiny ? 0.0 : fmin(x, 1.0e9);
    const double n = rint(xc * 1.4426950408889634);           // log2(e)
    double r = fma(-n, 0x1.62e42fefa39efp-1, xc);             // ln 2, high and low part
    r = fma(-n, 0x1.abc9e3b39803fp-56, r);
    // exp(r), |r| <= ln2 / 2: Taylor to r^13 (truncation 4e-18)
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    xnum o;
    o.f = tiny ? ((x != x) ? x : 0.0) : p;
    o.e = tiny ? XN_ZERO_E : (int)n;
    return o;
}

// The same function with the polynomial written as v_fma_f64 with the coefficient in an SGPR pair: left to itself the
// compiler turns every step into v_mov_b64 (coefficient -> accumulator) + v_fmac_f64 -- 12 extra issue slots per
// exponential, a tenth of a lane-per-cell forward-backward column.  Same operations in the same order: same bits.
__device__ __forceinline__ double xn_fma_sc(double p, double r, double c) {
    double o;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(p), "v"(r), "s"(c));
    return o;
}
__device__ __forceinline__ xnum xn_exp_neg_sc(double c) {
    const double x = -c;
    const bool tiny = !(x >= -1.0e9);
    const double xc = tiny ? 0.0 : fmin(x, 1.0e9);
    const double n = rint(xc * 1.4426950408889634);
    double r = fma(-n, 0x1.62e42fefa39efp-1, xc);
    r = fma(-n, 0x1.abc9e3b39803fp-56, r);
    double p = 1.0 / 6227020800.0;
    p = xn_fma_sc(p, r, 1.0 / 479001600.0);
    p = xn_fma_sc(p, r, 1.0 / 39916800.0);
    p = xn_fma_sc(p, r, 1.0 / 3628800.0);
    p = xn_fma_sc(p, r, 1.0 / 362880.0);
    p = xn_fma_sc(p, r, 1.0 / 40320.0);
    p = xn_fma_sc(p, r, 1.0 / 5040.0);
    p = xn_fma_sc(p, r, 1.0 / 720.0);
    p = xn_fma_sc(p, r, 1.0 / 120.0);
    p = xn_fma_sc(p, r, 1.0 / 24.0);
    p = xn_fma_sc(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    xnum o;
    o.f = tiny ? ((x != x) ? x : 0.0) : p;
    o.e = tiny ? XN_ZERO_E : (int)n;
    return o;
}

__device__ __forceinline__ xnum xn_mul(xnum a, xnum b) { return xnum{a.f * b.f, a.e + b.e}; }

// sums are left UNNORMALISED (f may be anything >= 0): normalise once per cell with xn_norm
__device__ __forceinline__ xnum xn_add(xnum a, xnum b) {
    const int m = max(a.e, b.e);
    return xnum{__builtin_amdgcn_ldexp(a.f, a.e - m) + __builtin_amdgcn_ldexp(b.f, b.e - m), m};
}
__device__ __forceinline__ xnum xn_add3(xnum a, xnum b, xnum c) {
    const int m = max(a.e, max(b.e, c.e));
    return xnum{__builtin_amdgcn_ldexp(a.f, a.e - m) + __builtin_amdgcn_ldexp(b.f, b.e - m) + __builtin_amdgcn_ldexp(c.f, c.e - m), m};
}
// f back into [0.5, 1); 0 gets the canonical zero exponent
__device__ __forceinline__ xnum xn_norm(xnum a) {
    xnum o;
    o.f = __builtin_amdgcn_frexp_mant(a.f);
    o.e = (a.f == 0.0) ? XN_ZERO_E : a.e + __builtin_amdgcn_frexp_exp(a.f);
    return o;
}
// natural logarithm (-inf for 0)
__device__ __forceinline__ double xn_log(xnum a) {
    return (a.f == 0.0) ? -INFINITY : fma((double)a.e, 0x1.62e42fefa39efp-1, log(a.f)) + (double)a.e * 0x1.abc9e3b39803fp-56;
}
// a * b / p as a plain double, inv_pf = 1 / p.f (a quantity in [0, 1] up to rounding: plain double range is enough)
__device__ __forceinline__ double xn_ratio(xnum a, xnum b, double inv_pf, int pe) {
    return __builtin_amdgcn_ldexp(a.f * b.f * inv_pf, max(a.e + b.e - pe, -4000));
}
