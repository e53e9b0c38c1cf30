// gpk_kmat.hip -- fused pairwise-distance + kernel evaluation.
//
//   out[i][j] (+)= sum_t variance_t * kappa_t( x_i, y_j ; inv_ls_t )  [+ diag_add + diag_vec[i] if i == j]
//
// Replaces: mlkernels `pairwise` for EQ / Matern12 (Exp) / Matern32 / Matern52 /
// Linear and Sum / Scaled / Stretched combinations of them, plus the
// `B.add(k(x), noise)` of stheno/model/fdd.py:79 and observations.py:139,286
// (the noise / jitter diagonal is fused into the same pass), and `elwise`
// (fdd.py:66, observations.py:304) as gpk_kdiag.  Formula evidence for EQ:
// exp(-0.5 * pw_dists2), tests/model/test_model.py:342-345.
//
// HBM-write-bound.  One workgroup (4 waves) produces a 32 x (64*VEC) tile:
// every wave-store is one full, contiguous 1 KiB row segment (16 B per lane);
// the X rows of the tile sit in LDS and are read as wave-uniform broadcasts, the
// Y rows of a lane live in registers.  Squared distances are accumulated as
// direct differences (no |a|^2 + |b|^2 - 2ab cancellation).
#include "gpk_common.hpp"

enum { GPK_K_EQ = 0, GPK_K_MATERN12 = 1, GPK_K_MATERN32 = 2, GPK_K_MATERN52 = 3, GPK_K_LINEAR = 4, GPK_K_CONST = 5 };

GPK_KNOB(int, g_kmat_compact, 1);   // tuning knob (gpk_tune(34, v)): 1-D compact grid for the lower triangle of a square matrix
GPK_KNOB(int, g_kmat_band, 1);      // tuning knob (gpk_tune(12, v)): 1 = row-band kernel, 0 = the one-tile-per-workgroup kernel
GPK_KNOB(int, g_kmat_band_f64_sqrt, 1);   // tuning knob (gpk_tune(51, v)): fp64 kernels with a square root (Matern) take the row-band kernel too
void gpk_tune_kmat(int key, int64_t value) {
    if (key == 51) GPK_KNOB_SET(g_kmat_band_f64_sqrt = (int)value;);
    if (key == 12) GPK_KNOB_SET(g_kmat_band = (int)value;);
    if (key == 34) GPK_KNOB_SET(g_kmat_compact = (int)value;);
}

namespace {

constexpr int TM = 32;   // tile rows
constexpr int RW = 8;    // rows per wave
// input dimensions per staged chunk: template parameter DC in {1, 2, 4, 8} -- the smallest that holds d
// (the distance loop runs over the whole chunk; with D = 1 or 3 a fixed chunk of 8 is mostly zero padding)

template <typename T>
struct KTermT {
    int kind;
    T variance;
    T ils2;   // squared inverse length scale
};

template <typename T>
struct KmatArgs {
    const T* X;
    const T* Y;
    T* out;
    const T* diag_vec;
    int64_t ldx, ldy, sX, sY, ld, sO, sDiag;
    int n, m, d;
    int nterms;
    KTermT<T> terms[GPK_MAX_TERMS];
    T diag_add;
    int symmetric, lower_only, accumulate, need_dot, vec_ok;
    int ct;           // row-band kernel: column tiles per workgroup
    int nbands;       // row-band kernel: number of row bands (TM rows each)
    int compact;      // row-band kernel, lower triangle of one square matrix: a 1-D grid of exactly the (row band, column chunk) pairs on or
                      // below the diagonal -- `compact` = row bands per column chunk; 0 = the plain 2-D grid
};

__device__ __forceinline__ double gpk_exp_neg(double a);     // (below) branch-free fp64 exp of a non-positive argument
template <typename T>
__device__ __forceinline__ T gpk_exp(T x);
template <>
__device__ __forceinline__ double gpk_exp<double>(double x) { return gpk_exp_neg(x); }   // every argument on this path is <= 0
template <>
__device__ __forceinline__ float gpk_exp<float>(float x) { return expf(x); }
#ifndef GPK_KMAT_R4_MATH
#define GPK_KMAT_R4_MATH 0       // 1: round 4's fp64 exp (degree-13 polynomial) and sqrt (two Goldschmidt steps), for `make ab`
#endif
template <typename T>
__device__ __forceinline__ T gpk_sqrtk(T x);
// sqrt of a squared distance times a positive constant (x >= 0, often exactly 0 on the diagonal): the hardware rsq estimate
// (~2^-23 relative), ONE coupled Goldschmidt step (-> ~2^-45 on both the root and the half reciprocal root) and one residual
// correction, which is a Newton step of its own (-> rounding level; the second Goldschmidt step of sqrt_rsqrt in gpk_potrf.hip
// bought nothing measurable here) -- 8 FMA-class operations, no range scaling, no branch, no select -- instead of the library
// sqrt.  The argument is shifted by 1e-280 (rsq(0) would overflow): invisible from 1e-264 up, and sqrt(0) comes out as 1e-140,
// which moves a kernel value by < 1e-140.  NaN stays NaN.
template <>
__device__ __forceinline__ double gpk_sqrtk<double>(double x) {
    const double xc = x + 1e-280;
    const double y = __builtin_amdgcn_rsq(xc);
    double g = xc * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
#if GPK_KMAT_R4_MATH
    e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
#endif
    return fma(fma(-g, g, xc), h, g);
}
template <>
__device__ __forceinline__ float gpk_sqrtk<float>(float x) { return sqrtf(x); }

template <typename T>
__device__ __forceinline__ T eval_terms(const KmatArgs<T>& p, T r2, T dot) {
    T val = T(0);
    for (int t = 0; t < p.nterms; ++t) {
        const int kind = p.terms[t].kind;
        const T q = r2 * p.terms[t].ils2;
        T k;
        if (kind == GPK_K_EQ) {
            k = gpk_exp<T>(T(-0.5) * q);
        } else if (kind == GPK_K_MATERN12) {
            k = gpk_exp<T>(-gpk_sqrtk<T>(q));
        } else if (kind == GPK_K_MATERN32) {
            const T s = gpk_sqrtk<T>(T(3) * q);
            k = (T(1) + s) * gpk_exp<T>(-s);
        } else if (kind == GPK_K_MATERN52) {
            const T s = gpk_sqrtk<T>(T(5) * q);
            k = (T(1) + s + s * s * T(1.0 / 3.0)) * gpk_exp<T>(-s);
        } else if (kind == GPK_K_LINEAR) {
            k = dot * p.terms[t].ils2;
        } else {
            k = T(1);
        }
        val += p.terms[t].variance * k;
    }
    return val;
}

template <typename T, bool DOT, int DC>
__global__ __launch_bounds__(256) void kmat_kernel(KmatArgs<T> p) {
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    constexpr int TN = 64 * VEC;
    __shared__ T xs[TM * DC];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.y * TM;
    const int col0 = blockIdx.x * TN;
    if (p.lower_only && col0 > row0 + TM - 1) return;

    const int64_t b = blockIdx.z;
    const T* __restrict__ X = p.X + b * p.sX;
    const T* __restrict__ Y = p.Y + b * p.sY;
    T* __restrict__ out = p.out + b * p.sO;

    const int colb = col0 + lane * VEC;   // first of this lane's VEC columns

    T r2[RW][VEC];
    T dt[RW][VEC];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            r2[r][v] = T(0);
            dt[r][v] = T(0);
        }

    for (int dc = 0; dc < p.d; dc += DC) {
        __syncthreads();
        if (tid < TM * DC) {   // stage X chunk: TM*DC <= 256 elements, one per thread
            const int r = tid / DC, j = tid % DC;
            const int row = row0 + r;
            xs[tid] = (row < p.n && dc + j < p.d) ? X[(int64_t)row * p.ldx + dc + j] : T(0);
        }
        T yv[VEC][DC];
#pragma unroll
        for (int v = 0; v < VEC; ++v)
#pragma unroll
            for (int j = 0; j < DC; ++j) {
                const int col = colb + v;
                yv[v][j] = (col < p.m && dc + j < p.d) ? Y[(int64_t)col * p.ldy + dc + j] : T(0);
            }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int rl = wave * RW + r;
#pragma unroll
            for (int j = 0; j < DC; ++j) {
                const T xv = xs[rl * DC + j];
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const T df = xv - yv[v][j];
                    r2[r][v] += df * df;
                    if (DOT) dt[r][v] += xv * yv[v][j];
                }
            }
        }
    }

#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int row = row0 + wave * RW + r;
        if (row >= p.n) continue;
        T vals[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int col = colb + v;
            T val = eval_terms<T>(p, r2[r][v], dt[r][v]);
            if (p.symmetric && col == row) {
                val += p.diag_add;
                if (p.diag_vec != nullptr) val += p.diag_vec[b * p.sDiag + row];
            }
            vals[v] = val;
        }
        T* o = out + (int64_t)row * p.ld + colb;
        if (p.vec_ok && colb + VEC <= p.m) {
            vec_t w;
            if (p.accumulate) {
                w = *reinterpret_cast<const vec_t*>(o);
#pragma unroll
                for (int v = 0; v < VEC; ++v) w[v] += vals[v];
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) w[v] = vals[v];
            }
            *reinterpret_cast<vec_t*>(o) = w;
        } else {
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                if (colb + v < p.m) o[v] = p.accumulate ? o[v] + vals[v] : vals[v];
        }
    }
}


// ---------------------------------------------------------------------------
// Row-band kernel (the default): a workgroup owns a band of TM rows and walks CT consecutive column tiles of it.
//  * the X rows of the band are staged in LDS ONCE (input dimension <= 8: one chunk) and the walk has no
//    barrier: each wave loads the Y values of the NEXT column tile while it evaluates the current one;
//  * every row of the band is written as CT back-to-back 1 KiB wave-stores -- 8 KiB contiguous per row and
//    workgroup instead of isolated 1 KiB pieces 64-128 KiB apart (HBM write locality);
//  * the kernel "program" is resolved at compile time for the common cases -- one EQ / Matern12 / 32 / 52 term,
//    or EQ + Linear -- instead of a per-element loop over a term table with a kind switch; fp32 EQ is one
//    multiply + v_exp_f32 (exp2 of a pre-scaled argument, <= 2 ulp) instead of the library expf.
// ---------------------------------------------------------------------------
enum { PROG_GENERIC = -1, PROG_EQ = 0, PROG_M12 = 1, PROG_M32 = 2, PROG_M52 = 3, PROG_EQ_LINEAR = 6 };

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
// Two values at a time (round 5): the fp32 kernels are bound by the vector ALU (43 operations per element at D = 8, issued at the
// chip's full rate), and gfx950 multiplies / adds / FMAs two fp32 values per lane and instruction (v_pk_*_f32).  Same arithmetic as
// above -- the error term through explicit FMAs, never left to contraction -- with the product, both FMAs and the final sum packed;
// clamp, rint, 2^x and ldexp have no packed form.
typedef float gpk_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ gpk_f2 gpk_exp_neg_pk(gpk_f2 a) {
    a.x = (a.x < -104.f) ? -104.f : a.x;
    a.y = (a.y < -104.f) ? -104.f : a.y;
    const gpk_f2 L = {1.44269502162933349609375f, 1.44269502162933349609375f}, Ll = {1.925963033500011e-8f, 1.925963033500011e-8f};
    const gpk_f2 t = a * L;
    gpk_f2 e = __builtin_elementwise_fma(a, L, -t);
    e = __builtin_elementwise_fma(a, Ll, e);
    const gpk_f2 n = {rintf(t.x), rintf(t.y)};
    const gpk_f2 f = (t - n) + e;
    return gpk_f2{ldexpf(__builtin_amdgcn_exp2f(f.x), (int)n.x), ldexpf(__builtin_amdgcn_exp2f(f.y), (int)n.y)};
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

// The same with a table of 2^(j/128) (in LDS: `tab`): a = (128 m + j) ln2/128 + r, |r| <= ln2/256 = 0.0027, so the polynomial stops at
// degree 5 (truncation r^6/720 < 6e-19) and exp(a) = 2^m (T_j + T_j (e^r - 1)): 12 FMA-class operations + 3 integer ones + one
// 8-byte LDS read instead of 19.  Error: the table entry's rounding, one FMA rounding and ~0.003 ulp from the polynomial.
__device__ const double gpk_exp2_128[128] = {
    0x1.0000000000000p+0, 0x1.0163da9fb3335p+0, 0x1.02c9a3e778061p+0, 0x1.04315e86e7f85p+0,
    0x1.059b0d3158574p+0, 0x1.0706b29ddf6dep+0, 0x1.0874518759bc8p+0, 0x1.09e3ecac6f383p+0,
    0x1.0b5586cf9890fp+0, 0x1.0cc922b7247f7p+0, 0x1.0e3ec32d3d1a2p+0, 0x1.0fb66affed31bp+0,
    0x1.11301d0125b51p+0, 0x1.12abdc06c31ccp+0, 0x1.1429aaea92de0p+0, 0x1.15a98c8a58e51p+0,
    0x1.172b83c7d517bp+0, 0x1.18af9388c8deap+0, 0x1.1a35beb6fcb75p+0, 0x1.1bbe084045cd4p+0,
    0x1.1d4873168b9aap+0, 0x1.1ed5022fcd91dp+0, 0x1.2063b88628cd6p+0, 0x1.21f49917ddc96p+0,
    0x1.2387a6e756238p+0, 0x1.251ce4fb2a63fp+0, 0x1.26b4565e27cddp+0, 0x1.284dfe1f56381p+0,
    0x1.29e9df51fdee1p+0, 0x1.2b87fd0dad990p+0, 0x1.2d285a6e4030bp+0, 0x1.2ecafa93e2f56p+0,
    0x1.306fe0a31b715p+0, 0x1.32170fc4cd831p+0, 0x1.33c08b26416ffp+0, 0x1.356c55f929ff1p+0,
    0x1.371a7373aa9cbp+0, 0x1.38cae6d05d866p+0, 0x1.3a7db34e59ff7p+0, 0x1.3c32dc313a8e5p+0,
    0x1.3dea64c123422p+0, 0x1.3fa4504ac801cp+0, 0x1.4160a21f72e2ap+0, 0x1.431f5d950a897p+0,
    0x1.44e086061892dp+0, 0x1.46a41ed1d0057p+0, 0x1.486a2b5c13cd0p+0, 0x1.4a32af0d7d3dep+0,
    0x1.4bfdad5362a27p+0, 0x1.4dcb299fddd0dp+0, 0x1.4f9b2769d2ca7p+0, 0x1.516daa2cf6642p+0,
    0x1.5342b569d4f82p+0, 0x1.551a4ca5d920fp+0, 0x1.56f4736b527dap+0, 0x1.58d12d497c7fdp+0,
    0x1.5ab07dd485429p+0, 0x1.5c9268a5946b7p+0, 0x1.5e76f15ad2148p+0, 0x1.605e1b976dc09p+0,
    0x1.6247eb03a5585p+0, 0x1.6434634ccc320p+0, 0x1.6623882552225p+0, 0x1.68155d44ca973p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6c012750bdabfp+0, 0x1.6dfb23c651a2fp+0, 0x1.6ff7df9519484p+0,
    0x1.71f75e8ec5f74p+0, 0x1.73f9a48a58174p+0, 0x1.75feb564267c9p+0, 0x1.780694fde5d3fp+0,
    0x1.7a11473eb0187p+0, 0x1.7c1ed0130c132p+0, 0x1.7e2f336cf4e62p+0, 0x1.80427543e1a12p+0,
    0x1.82589994cce13p+0, 0x1.8471a4623c7adp+0, 0x1.868d99b4492edp+0, 0x1.88ac7d98a6699p+0,
    0x1.8ace5422aa0dbp+0, 0x1.8cf3216b5448cp+0, 0x1.8f1ae99157736p+0, 0x1.9145b0b91ffc6p+0,
    0x1.93737b0cdc5e5p+0, 0x1.95a44cbc8520fp+0, 0x1.97d829fde4e50p+0, 0x1.9a0f170ca07bap+0,
    0x1.9c49182a3f090p+0, 0x1.9e86319e32323p+0, 0x1.a0c667b5de565p+0, 0x1.a309bec4a2d33p+0,
    0x1.a5503b23e255dp+0, 0x1.a799e1330b358p+0, 0x1.a9e6b5579fdbfp+0, 0x1.ac36bbfd3f37ap+0,
    0x1.ae89f995ad3adp+0, 0x1.b0e07298db666p+0, 0x1.b33a2b84f15fbp+0, 0x1.b59728de5593ap+0,
    0x1.b7f76f2fb5e47p+0, 0x1.ba5b030a1064ap+0, 0x1.bcc1e904bc1d2p+0, 0x1.bf2c25bd71e09p+0,
    0x1.c199bdd85529cp+0, 0x1.c40ab5fffd07ap+0, 0x1.c67f12e57d14bp+0, 0x1.c8f6d9406e7b5p+0,
    0x1.cb720dcef9069p+0, 0x1.cdf0b555dc3fap+0, 0x1.d072d4a07897cp+0, 0x1.d2f87080d89f2p+0,
    0x1.d5818dcfba487p+0, 0x1.d80e316c98398p+0, 0x1.da9e603db3285p+0, 0x1.dd321f301b460p+0,
    0x1.dfc97337b9b5fp+0, 0x1.e264614f5a129p+0, 0x1.e502ee78b3ff6p+0, 0x1.e7a51fbc74c83p+0,
    0x1.ea4afa2a490dap+0, 0x1.ecf482d8e67f1p+0, 0x1.efa1bee615a27p+0, 0x1.f252b376bba97p+0,
    0x1.f50765b6e4540p+0, 0x1.f7bfdad9cbe14p+0, 0x1.fa7c1819e90d8p+0, 0x1.fd3c22b8f71f1p+0,
};
__device__ __forceinline__ double gpk_exp_neg_tab(double a, const double* tab) {
    a = (a < -750.0) ? -750.0 : a;
    const double n = rint(a * 184.6649652337873);                  // 128 / ln 2
    double r = fma(-n, 6.93147180369123816490e-01 / 128, a);      // ln2_hi / 128 (32 significant bits: exact times n < 2^17)
    r = fma(-n, 1.90821492927058770002e-10 / 128, r);             // ln2_lo / 128
    const int ni = (int)n;
    const double t = tab[ni & 127];
    double u = 8.333333333333333e-03;        // 1/5!
    u = fma(u, r, 4.1666666666666664e-02);
    u = fma(u, r, 1.6666666666666666e-01);
    u = fma(u, r, 0.5);
    u = fma(u, r, 1.0);
    return ldexp(fma(t, u * r, t), ni >> 7);
}
__device__ __forceinline__ double gpk_exp_neg_t(double a, const double* tab) { return GPK_KMAT_R4_MATH ? gpk_exp_neg(a) : gpk_exp_neg_tab(a, tab); }
__device__ __forceinline__ float gpk_exp_neg_t(float a, const float*) { return gpk_exp_neg(a); }

template <typename T, int PROG>
__device__ __forceinline__ T eval_prog(const KmatArgs<T>& p, T r2, T dot, const T* etab) {
    if (PROG == PROG_GENERIC) return eval_terms<T>(p, r2, dot);
    const T v0 = p.terms[0].variance, c0 = p.terms[0].ils2;
    if (PROG == PROG_EQ) return v0 * gpk_exp_neg_t(T(-0.5) * c0 * r2, etab);
    if (PROG == PROG_EQ_LINEAR) return v0 * gpk_exp_neg_t(T(-0.5) * c0 * r2, etab) + p.terms[1].variance * p.terms[1].ils2 * dot;
    // (the constants below are uniform: formed once per workgroup.  3 c0 r2 for (3 c0) r2 and v0 + v0 s + (v0 / 3) s^2 in Horner form
    // move a value by an ulp or two against the literal formula -- the price of 1 + 2 operations less per element)
    if (PROG == PROG_M12) return v0 * gpk_exp_neg_t(-gpk_sqrtk<T>(r2 * c0), etab);
    if (PROG == PROG_M32) {
        const T sd = gpk_sqrtk<T>(r2 * (T(3) * c0));
        return fma(v0, sd, v0) * gpk_exp_neg_t(-sd, etab);
    }
    const T sd = gpk_sqrtk<T>(r2 * (T(5) * c0));
    return fma(fma(v0 * T(1.0 / 3.0), sd, v0), sd, v0) * gpk_exp_neg_t(-sd, etab);
}

constexpr int CT_MAX = 8;    // column tiles per workgroup (fewer when the grid would not fill the chip a few times over)

template <typename T, int PROG, bool DOT, int DC>
__global__ __launch_bounds__(256) void kmat_band_kernel(KmatArgs<T> p) {
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    constexpr int TN = 64 * VEC;
    constexpr int TNP = TN + 4;                 // LDS row pitch of the transposed Y tile (bank spread of the staging writes)
    constexpr int YL = (TN * DC + 255) / 256;   // Y elements each thread stages per column tile
    __shared__ T xs[TM * DC];
    __shared__ __attribute__((aligned(16))) T ys[2][DC * TNP];   // Y tile, dimension-major: ys[j][c]
    constexpr bool ETAB = sizeof(T) == 8 && PROG != PROG_GENERIC;  // fp64: exp through the 2^(j/128) table (gpk_exp_neg_tab)
    constexpr bool PK32 = sizeof(T) == 4 && PROG != PROG_GENERIC && !GPK_KMAT_R4_MATH;   // fp32: two values per instruction
    __shared__ T etab[ETAB ? 128 : 1];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int by = blockIdx.y, bx = blockIdx.x;
    if (p.compact) {
        // 1-D grid over the (row band, column chunk) pairs on or below the diagonal: the bands of chunk-group g (G = p.compact
        // consecutive row bands) have g + 1 chunks each.  Groups are laid out from the bottom of the matrix up (full chunks first).
        const int G = p.compact;
        int bid = (int)blockIdx.x;
        int g = (p.nbands + G - 1) / G - 1;
        for (; g > 0; --g) {
            const int cnt = min(G, p.nbands - g * G) * (g + 1);
            if (bid < cnt) break;
            bid -= cnt;
        }
        by = g * G + bid / (g + 1);
        bx = bid % (g + 1);
        if (by >= p.nbands) return;
    }
    const int row0 = by * TM;
    const int CT = p.ct;
    const int ct0 = bx * CT;
    if (p.lower_only && ct0 * TN > row0 + TM - 1) return;

    const int64_t b = blockIdx.z;
    const T* __restrict__ X = p.X + b * p.sX;
    const T* __restrict__ Y = p.Y + b * p.sY;
    T* __restrict__ out = p.out + b * p.sO;

    if (tid < TM * DC) {      // the band's X rows: TM * DC <= 256 elements, one per thread; d <= DC (launcher)
        const int r = tid / DC, j = tid % DC;
        const int row = row0 + r;
        xs[tid] = (row < p.n && j < p.d) ? X[(int64_t)row * p.ldx + j] : T(0);
    }
    // A column tile of Y is TN points x d values, contiguous in memory when ldy == d: consecutive lanes fetch
    // consecutive elements (two cache lines per wave-load); a lane fetching "its own" columns straight from global
    // memory touches 64 different lines per load instruction, and the L1 then sets the pace of the whole kernel
    // (measured: same time in fp32 and fp64, 10x below the ALU and 4x below the HBM bound).
    T stage[YL];
    auto fetch_y = [&](int col0) {
#pragma unroll
        for (int k = 0; k < YL; ++k) {
            const int idx = tid + 256 * k;
            const int c = idx / DC, j = idx % DC;
            const int col = col0 + c;
            stage[k] = (idx < TN * DC && col < p.m && j < p.d) ? Y[(int64_t)col * p.ldy + j] : T(0);
        }
    };
    auto commit_y = [&](int buf) {
#pragma unroll
        for (int k = 0; k < YL; ++k) {
            const int idx = tid + 256 * k;
            if (idx < TN * DC) ys[buf][(idx % DC) * TNP + idx / DC] = stage[k];
        }
    };
    fetch_y(ct0 * TN);
    commit_y(0);
    if (ETAB && tid < 128) etab[tid] = (T)gpk_exp2_128[tid];
    __syncthreads();

    for (int c = 0; c < CT; ++c) {
        const int col0 = (ct0 + c) * TN;
        if (col0 >= p.m || (p.lower_only && col0 > row0 + TM - 1)) break;       // (uniform over the workgroup)
        const bool more = (c + 1 < CT) && (col0 + TN < p.m) && !(p.lower_only && col0 + TN > row0 + TM - 1);
        if (more) fetch_y(col0 + TN);                 // next tile's Y: in flight under this tile's arithmetic
        T ya[DC][VEC];
#pragma unroll
        for (int j = 0; j < DC; ++j) {
            const vec_t w = *reinterpret_cast<const vec_t*>(&ys[c & 1][j * TNP + lane * VEC]);
#pragma unroll
            for (int v = 0; v < VEC; ++v) ya[j][v] = w[v];
        }
        const int colb = col0 + lane * VEC;
        const bool has_diag = p.symmetric && col0 <= row0 + TM - 1 && col0 + TN > row0;   // (uniform) only such tiles touch the diagonal
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int row = row0 + wave * RW + r;
            T xr[DC];              // this row of the band: wave-uniform LDS broadcast reads
#pragma unroll
            for (int j = 0; j < DC; ++j) xr[j] = xs[(wave * RW + r) * DC + j];
            T vals[VEC];
            if constexpr (PK32) {
                // fp32, every compile-time kernel program: the row's four values as two pairs on the packed fp32 instructions (see gpk_exp_neg_pk)
                gpk_f2 acc[2] = {gpk_f2{0.f, 0.f}, gpk_f2{0.f, 0.f}}, dot[2] = {gpk_f2{0.f, 0.f}, gpk_f2{0.f, 0.f}};
#pragma unroll
                for (int j = 0; j < DC; ++j) {
                    const gpk_f2 xx = {(float)xr[j], (float)xr[j]};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const gpk_f2 yy = {(float)ya[j][2 * h], (float)ya[j][2 * h + 1]};
                        const gpk_f2 df = xx - yy;
                        acc[h] = __builtin_elementwise_fma(df, df, acc[h]);
                        if (DOT) dot[h] = __builtin_elementwise_fma(xx, yy, dot[h]);
                    }
                }
                const float c0 = (float)p.terms[0].ils2, v0 = (float)p.terms[0].variance;
                const float vl = (PROG == PROG_EQ_LINEAR) ? (float)(p.terms[1].variance * p.terms[1].ils2) : 0.f;
                const gpk_f2 vv = {v0, v0};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    gpk_f2 k;
                    if (PROG == PROG_EQ || PROG == PROG_EQ_LINEAR) {
                        const float ca = -0.5f * c0;
                        k = gpk_exp_neg_pk(acc[h] * gpk_f2{ca, ca}) * vv;
                        if (PROG == PROG_EQ_LINEAR) k = __builtin_elementwise_fma(dot[h], gpk_f2{vl, vl}, k);
                    } else {
                        // Matern: sqrt straight from v_sqrt_f32 (1 ulp; the library sqrtf spends ~10 more instructions on rescaling
                        // denormal arguments, i.e. distances below 1e-19, and on the last half ulp)
                        const float cs = (PROG == PROG_M12 ? 1.f : PROG == PROG_M32 ? 3.f : 5.f) * c0;
                        const gpk_f2 q = acc[h] * gpk_f2{cs, cs};
                        const gpk_f2 sd = {__builtin_amdgcn_sqrtf(q.x), __builtin_amdgcn_sqrtf(q.y)};
                        const gpk_f2 e = gpk_exp_neg_pk(-sd);
                        if (PROG == PROG_M12) k = e * vv;
                        else if (PROG == PROG_M32) k = __builtin_elementwise_fma(sd, vv, vv) * e;
                        else k = __builtin_elementwise_fma(__builtin_elementwise_fma(sd, gpk_f2{v0 * (1.f / 3.f), v0 * (1.f / 3.f)}, vv), sd, vv) * e;
                    }
                    vals[2 * h] = (T)k.x;
                    vals[2 * h + 1] = (T)k.y;
                }
                if (has_diag) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        if (colb + v == row) {
                            vals[v] += p.diag_add;
                            if (p.diag_vec != nullptr) vals[v] += p.diag_vec[b * p.sDiag + row];
                        }
                }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    T r2 = T(0), dt = T(0);
#pragma unroll
                    for (int j = 0; j < DC; ++j) {
                        const T df = xr[j] - ya[j][v];
                        r2 += df * df;
                        if (DOT) dt += xr[j] * ya[j][v];
                    }
                    T val = eval_prog<T, PROG>(p, r2, dt, etab);
                    if (has_diag && colb + v == row) {
                        val += p.diag_add;
                        if (p.diag_vec != nullptr) val += p.diag_vec[b * p.sDiag + row];
                    }
                    vals[v] = val;
                }
            }
            if (row >= p.n) continue;
            T* o = out + (int64_t)row * p.ld + colb;
            if (p.vec_ok && colb + VEC <= p.m) {
                vec_t w;
                if (p.accumulate) {
                    w = *reinterpret_cast<const vec_t*>(o);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) w[v] += vals[v];
                    *reinterpret_cast<vec_t*>(o) = w;
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) w[v] = vals[v];
                    __builtin_nontemporal_store(w, reinterpret_cast<vec_t*>(o));     // written once, read by a later kernel
                }
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v)
                    if (colb + v < p.m) o[v] = p.accumulate ? o[v] + vals[v] : vals[v];
            }
        }
        if (more) commit_y((c + 1) & 1);     // the other buffer: nobody reads it during this iteration
        __syncthreads();
    }
}

template <typename T>
struct KdiagArgs {
    const T* X;
    T* out;
    int64_t ldx, sX, sO;
    int n, d, nterms;
    KTermT<T> terms[GPK_MAX_TERMS];
};

template <typename T>
__global__ __launch_bounds__(256) void kdiag_kernel(KdiagArgs<T> p) {
    const int64_t b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const T* x = p.X + b * p.sX + i * p.ldx;
    T nrm = T(0);
    for (int j = 0; j < p.d; ++j) nrm += x[j] * x[j];
    T val = T(0);
    for (int t = 0; t < p.nterms; ++t)
        val += p.terms[t].variance * (p.terms[t].kind == GPK_K_LINEAR ? nrm * p.terms[t].ils2 : T(1));
    p.out[b * p.sO + i] = val;
}

}  // namespace

// terms: host arrays of length nterms
template <typename T>
int gpk_kmat_launch(const int* kinds, const double* variances, const double* inv_ls, int nterms,
                    const T* X, int64_t n, int64_t ldx, int64_t sX, const T* Y, int64_t m, int64_t ldy,
                    int64_t sY, int d, T* out, int64_t ld, int64_t sO, int64_t batch, int lower_only,
                    int symmetric, double diag_add, const T* diag_vec, int64_t sDiag, int accumulate,
                    hipStream_t stream) {
    if (n <= 0 || m <= 0 || batch <= 0) return GPK_OK;
    if (nterms < 0 || nterms > GPK_MAX_TERMS) return GPK_ERR_ARG(4);
    if (n > INT32_MAX || m > INT32_MAX || batch > 65535) return GPK_ERR_ARG(6);
    if (d < 0) return GPK_ERR_ARG(13);
    constexpr int VEC = Traits<T>::VEC;
    KmatArgs<T> a;
    a.X = X; a.Y = Y; a.out = out; a.diag_vec = diag_vec;
    a.ldx = ldx; a.ldy = ldy; a.sX = sX; a.sY = sY; a.ld = ld; a.sO = sO; a.sDiag = sDiag;
    a.n = (int)n; a.m = (int)m; a.d = d;
    a.nterms = nterms;
    a.need_dot = 0;
    for (int t = 0; t < nterms; ++t) {
        if (kinds[t] < GPK_K_EQ || kinds[t] > GPK_K_CONST) return GPK_ERR_ARG(1);
        a.terms[t].kind = kinds[t];
        a.terms[t].variance = (T)variances[t];
        a.terms[t].ils2 = (T)(inv_ls[t] * inv_ls[t]);
        if (kinds[t] == GPK_K_LINEAR) a.need_dot = 1;
    }
    a.diag_add = (T)diag_add;
    a.symmetric = symmetric; a.lower_only = lower_only; a.accumulate = accumulate;
    a.vec_ok = ((uintptr_t)out % 16 == 0) && (ld % VEC == 0) && (sO % VEC == 0);
    const int64_t gy = gpk_cdiv(n, TM);
    if (gy > 65535) return GPK_ERR_ARG(6);
    // program resolved at compile time where it is one of the common ones
    int prog = PROG_GENERIC;
    if (nterms == 1 && kinds[0] >= GPK_K_EQ && kinds[0] <= GPK_K_MATERN52) prog = kinds[0];
    if (nterms == 2 && kinds[0] == GPK_K_EQ && kinds[1] == GPK_K_LINEAR) prog = PROG_EQ_LINEAR;
    // (round 2, with the library sqrt + exp: the one-tile kernel was 15 % faster for fp64 kernels with a square root -- 0.53 vs 0.62 ms at
    // N = 16384 -- and kept them; round 5: with the branch-free exp of round 3 and the rsq-based sqrt above, knob 51 decides)
    const bool band_ok = sizeof(T) == 4 || prog == PROG_EQ || prog == PROG_EQ_LINEAR || g_kmat_band_f64_sqrt;
    if (d <= 8 && g_kmat_band && band_ok) {
        const int64_t tiles_x = gpk_cdiv(m, 64 * VEC);
        int ct = CT_MAX;
        while (ct > 1 && gpk_cdiv(tiles_x, ct) * gy * batch / (lower_only ? 2 : 1) < 6144) ct >>= 1;
        a.ct = ct;
        a.nbands = (int)gy;
        a.compact = 0;
        dim3 bgrid((unsigned)gpk_cdiv(tiles_x, ct), (unsigned)gy, (unsigned)batch);
        // Lower triangle of a square matrix with several column chunks per row band: half of the 2-D grid would be workgroups
        // that exit at once, and the dispatcher works through them in index order in front of the real ones (measured at
        // N = 16384: the lower triangle took 0.46 ms against 0.52 ms for the FULL matrix).  A 1-D grid of exactly the pairs needed.
        const int64_t chunk_cols = (int64_t)ct * 64 * VEC;
        if (g_kmat_compact && lower_only && symmetric && n == m && bgrid.x > 1 && chunk_cols % TM == 0) {
            const int64_t G = chunk_cols / TM;
            int64_t total = 0;
            for (int64_t g = 0; g * G < gy; ++g) total += ((gy - g * G < G) ? gy - g * G : G) * (g + 1);
            a.compact = (int)G;
            bgrid = dim3((unsigned)total, 1u, (unsigned)batch);
        }
#define GPK_BAND(PROGV, DOTV, DCV) hipLaunchKernelGGL((kmat_band_kernel<T, PROGV, DOTV, DCV>), bgrid, dim3(256), 0, stream, a)
#define GPK_BAND_DC(PROGV, DOTV)                      \
    do {                                              \
        if (d <= 1) GPK_BAND(PROGV, DOTV, 1);         \
        else if (d <= 2) GPK_BAND(PROGV, DOTV, 2);    \
        else if (d <= 4) GPK_BAND(PROGV, DOTV, 4);    \
        else GPK_BAND(PROGV, DOTV, 8);                \
    } while (0)
        switch (prog) {
            case PROG_EQ: GPK_BAND_DC(PROG_EQ, false); break;
            case PROG_M12: GPK_BAND_DC(PROG_M12, false); break;
            case PROG_M32: GPK_BAND_DC(PROG_M32, false); break;
            case PROG_M52: GPK_BAND_DC(PROG_M52, false); break;
            case PROG_EQ_LINEAR: GPK_BAND_DC(PROG_EQ_LINEAR, true); break;
            default:
                if (a.need_dot) GPK_BAND_DC(PROG_GENERIC, true);
                else GPK_BAND_DC(PROG_GENERIC, false);
        }
#undef GPK_BAND_DC
#undef GPK_BAND
        GPK_CHECK_LAUNCH();
        return GPK_OK;
    }
    dim3 grid((unsigned)gpk_cdiv(m, 64 * VEC), (unsigned)gy, (unsigned)batch);
#define GPK_KMAT_LAUNCH(DCV)                                                                     \
    do {                                                                                         \
        if (a.need_dot)                                                                          \
            hipLaunchKernelGGL((kmat_kernel<T, true, DCV>), grid, dim3(256), 0, stream, a);      \
        else                                                                                     \
            hipLaunchKernelGGL((kmat_kernel<T, false, DCV>), grid, dim3(256), 0, stream, a);     \
    } while (0)
    if (d <= 1)
        GPK_KMAT_LAUNCH(1);
    else if (d <= 2)
        GPK_KMAT_LAUNCH(2);
    else if (d <= 4)
        GPK_KMAT_LAUNCH(4);
    else
        GPK_KMAT_LAUNCH(8);
#undef GPK_KMAT_LAUNCH
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

template <typename T>
int gpk_kdiag_launch(const int* kinds, const double* variances, const double* inv_ls, int nterms,
                     const T* X, int64_t n, int64_t ldx, int64_t sX, int d, T* out, int64_t sO,
                     int64_t batch, hipStream_t stream) {
    if (n <= 0 || batch <= 0) return GPK_OK;
    if (nterms < 0 || nterms > GPK_MAX_TERMS) return GPK_ERR_ARG(4);
    KdiagArgs<T> a;
    a.X = X; a.out = out; a.ldx = ldx; a.sX = sX; a.sO = sO;
    a.n = (int)n; a.d = d; a.nterms = nterms;
    for (int t = 0; t < nterms; ++t) {
        a.terms[t].kind = kinds[t];
        a.terms[t].variance = (T)variances[t];
        a.terms[t].ils2 = (T)(inv_ls[t] * inv_ls[t]);
    }
    dim3 grid((unsigned)gpk_cdiv(n, 256), (unsigned)batch);
    hipLaunchKernelGGL((kdiag_kernel<T>), grid, dim3(256), 0, stream, a);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

#define GPK_INST(T)                                                                                   \
    template int gpk_kmat_launch<T>(const int*, const double*, const double*, int, const T*, int64_t, \
                                    int64_t, int64_t, const T*, int64_t, int64_t, int64_t, int, T*,   \
                                    int64_t, int64_t, int64_t, int, int, double, const T*, int64_t,   \
                                    int, hipStream_t);                                                \
    template int gpk_kdiag_launch<T>(const int*, const double*, const double*, int, const T*, int64_t, \
                                     int64_t, int64_t, int, T*, int64_t, int64_t, hipStream_t);
GPK_INST(double)
GPK_INST(float)
