// gpk_kmat_eval.hpp -- the scalar functions of the kernel-matrix kernels (exp of a non-positive argument, sqrt of a squared
// distance), shared by gpk_kmat.hip and by the GEMM tile that GENERATES its C tile from the inputs (gpk_gemm_tile.hpp, CGEN: the
// first trailing update of a batched factorisation never reads the kernel matrix it updates -- it evaluates it).
#pragma once
#include "gpk_common.hpp"

namespace {

// exp(a), a <= 0, through v_exp_f32 (2^x) at libm accuracy: the product a * log2(e) is formed with its rounding error
// (two FMAs), split into an integer and a fraction in [-0.5, 0.5], and only the fraction (plus the error) goes through the
// hardware 2^x; the integer part is applied exactly by v_ldexp_f32.  (The bare exp2(a * log2e) is 3+ ulp off for |a| ~ 10 --
// enough to cost the fp32 posterior mean of cfg3 its 1e-3: the solve amplifies kernel-matrix errors by kappa ~ 1e5.)
__device__ __forceinline__ float gpk_exp_neg(float a) {
    a = (a < -104.f) ? -104.f : a;       // exp(-104) already underflows to 0 in fp32; keeps -inf (huge distances) from turning into inf - inf below; NaN stays NaN
    const float L = 1.44269502162933349609375f, Ll = 1.925963033500011e-8f;     // log2(e) = L + Ll
    const float t = a * L;
    float e = fmaf(a, L, -t);
    e = fmaf(a, Ll, e);
    const float n = rintf(t);
    const float f = (t - n) + e;
    return ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}
// fp64: Cody-Waite reduction a = n ln2 + r (|r| <= ln2 / 2, two-part ln2, one FMA each), the Taylor polynomial of degree 13 in
// Horner form (truncation 4e-18 relative on that interval) and v_ldexp_f64: 13 + 4 FMA-class operations and no branch, against
// ~3x that with branches for the library exp, which was what bounded the fp64 EQ build (2.3 of 8 TB/s).  Measured against
// expl() on 2e7 arguments in [-700, 0]: <= 0.87 ulp.  Arguments below -750 (the result is 0 from -745.2 on) are clamped so that
// -inf gives 0 rather than inf - inf; NaN stays NaN.
__device__ __forceinline__ double gpk_exp_neg(double a) {
    a = (a < -750.0) ? -750.0 : a;
    const double n = rint(a * 1.4426950408889634);
    double r = fma(-n, 6.93147180369123816490e-01, a);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;          // 1/13!
    p = fma(p, r, 2.08767569878681e-09);        // 1/12!
    p = fma(p, r, 2.505210838544172e-08);       // 1/11!
    p = fma(p, r, 2.755731922398589e-07);       // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);      // 1/9!
    p = fma(p, r, 2.48015873015873e-05);        // 1/8!
    p = fma(p, r, 1.984126984126984e-04);       // 1/7!
    p = fma(p, r, 1.388888888888889e-03);       // 1/6!
    p = fma(p, r, 8.333333333333333e-03);       // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);      // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);      // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}


template <typename T>
__device__ __forceinline__ T gpk_exp(T x);
template <>
__device__ __forceinline__ double gpk_exp<double>(double x) { return gpk_exp_neg(x); }   // every argument on this path is <= 0
template <>
__device__ __forceinline__ float gpk_exp<float>(float x) { return expf(x); }
template <typename T>
__device__ __forceinline__ T gpk_sqrtk(T x);
// sqrt of a squared distance times a positive constant (x >= 0, often exactly 0 on the diagonal): the hardware rsq estimate, two
// coupled Goldschmidt steps and one residual correction (the scheme of sqrt_rsqrt in gpk_potrf.hip: ~1 ulp) -- 10 FMA-class
// operations, no range scaling, no branch -- instead of the library sqrt.  Arguments below 1e-280 (0 included: rsq would overflow)
// give 0: the kernel value moves by < 1e-140.  NaN stays NaN.
template <>
__device__ __forceinline__ double gpk_sqrtk<double>(double x) {
    const bool tiny = x < 1e-280;
    const double xc = tiny ? 1.0 : x;
    const double y = __builtin_amdgcn_rsq(xc);
    double g = xc * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    g = fma(fma(-g, g, xc), h, g);
    return tiny ? 0.0 : g;
}
template <>
__device__ __forceinline__ float gpk_sqrtk<float>(float x) { return sqrtf(x); }

}  // namespace
