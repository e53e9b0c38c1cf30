// selftest.cpp -- native GPU self-test + micro-benchmarks for libgpk.so.
// Test infrastructure only (host references are plain C++ loops).  Built by the
// Makefile into gpk_selftest; run on the MI355X box:
//     ./gpk_selftest            correctness of every entry point vs host loops
//     ./gpk_selftest --perf     + timings (HIP events) of the hot kernels
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <random>
#include <string>
#include <limits>
#include <vector>
#include <atomic>
#include <functional>
#include <thread>
#include "../../include/gpk.h"

#define HIPCHK(x)                                                                       \
    do {                                                                                \
        hipError_t e = (x);                                                             \
        if (e != hipSuccess) {                                                          \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(3);                                                                    \
        }                                                                               \
    } while (0)

static int g_fail = 0, g_pass = 0;
static void report(const std::string& name, double err, double tol) {
    const bool ok = (err <= tol) && std::isfinite(err);
    printf("%s %-58s err=%.3e tol=%.1e\n", ok ? "PASS" : "FAIL", name.c_str(), err, tol);
    fflush(stdout);
    if (ok) ++g_pass; else ++g_fail;
}

template <typename T> struct DT;
template <> struct DT<double> { static constexpr int v = GPK_F64; static constexpr double eps = 1e-12; static const char* name() { return "f64"; } };
template <> struct DT<float> { static constexpr int v = GPK_F32; static constexpr double eps = 2e-4; static const char* name() { return "f32"; } };

template <typename T>
struct Dev {
    T* p = nullptr;
    size_t n = 0;
    explicit Dev(size_t n_) : n(n_) { HIPCHK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T))); }
    ~Dev() { hipFree(p); }
    void up(const std::vector<T>& h) { HIPCHK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); }
    std::vector<T> down() const {
        std::vector<T> h(n);
        HIPCHK(hipMemcpy(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost));
        return h;
    }
    void zero() { HIPCHK(hipMemset(p, 0, n * sizeof(T))); }
};

// The host references are O(n^3) scalar loops; on one core they were 300 of the GPU suite's 510 seconds (round 6: 72 factorisations of
// order 2048 for ONE batched case).  They run on every host core now -- over the members of a batch, the rows of a column block, the
// rows under a matrix -- with every entry's summation order unchanged (same bits as the serial loops).
static void parallel_for(int n, const std::function<void(int)>& f, int grain = 1) {
    unsigned hw = std::thread::hardware_concurrency();
    const int nt = (int)std::min<unsigned>((unsigned)std::max((n + grain - 1) / grain, 1), std::min(hw ? hw : 1u, 64u));
    if (nt <= 1) {
        for (int i = 0; i < n; ++i) f(i);
        return;
    }
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    th.reserve(nt);
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&] {
            for (int i0 = next.fetch_add(grain); i0 < n; i0 = next.fetch_add(grain))
                for (int i = i0; i < std::min(n, i0 + grain); ++i) f(i);
        });
    for (auto& t : th) t.join();
}

static std::mt19937_64 rng(1234);
template <typename T>
static std::vector<T> randv(size_t n, double scale = 1.0) {
    std::normal_distribution<double> d(0.0, 1.0);
    std::vector<T> v(n);
    for (auto& x : v) x = (T)(scale * d(rng));
    return v;
}

template <typename T>
static double relerr(const std::vector<T>& got, const std::vector<double>& ref) {
    double num = 0, den = 0;
    for (size_t i = 0; i < ref.size(); ++i) {
        const double d = (double)got[i] - ref[i];
        if (!std::isfinite((double)got[i])) return INFINITY;
        num = std::max(num, std::fabs(d));
        den = std::max(den, std::fabs(ref[i]));
    }
    return num / std::max(den, 1e-300);
}

// ----------------------------------------------------------------------------
// MFMA layout probe (documents the lane maps the kernels rely on)
// ----------------------------------------------------------------------------
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe64(const double* A, const double* B, double* C) {   // A 16x4, B 4x16 row-major
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) C[((l >> 4) + 4 * i) * 16 + (l & 15)] = acc[i];
}
__global__ void probe32(const float* A, const float* B, float* C) {
    const int l = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) C[((l >> 4) * 4 + i) * 16 + (l & 15)] = acc[i];
}
template <typename T>
static void test_probe() {
    auto A = randv<T>(64), B = randv<T>(64);
    Dev<T> dA(64), dB(64), dC(256);
    dA.up(A); dB.up(B);
    if (sizeof(T) == 8) hipLaunchKernelGGL(probe64, dim3(1), dim3(64), 0, 0, (double*)dA.p, (double*)dB.p, (double*)dC.p);
    else hipLaunchKernelGGL(probe32, dim3(1), dim3(64), 0, 0, (float*)dA.p, (float*)dB.p, (float*)dC.p);
    HIPCHK(hipDeviceSynchronize());
    std::vector<double> ref(256, 0.0);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) ref[i * 16 + j] += (double)A[i * 4 + k] * (double)B[k * 16 + j];
    report(std::string("mfma_layout_") + DT<T>::name(), relerr(dC.down(), ref), DT<T>::eps);
}

// ----------------------------------------------------------------------------
// GEMM
// ----------------------------------------------------------------------------
template <typename T>
static void test_gemm_case(bool ak, bool bk, int M, int N, int K, double alpha, double beta, int batch, bool lower, int pad) {
    const int64_t lda = (ak ? K : M) + pad, ldb = (bk ? K : N) + pad, ldc = N + pad;
    const int64_t ra = ak ? M : K, rb = bk ? N : K;
    const int64_t sA = ra * lda + 2 * pad, sB = rb * ldb + 2 * pad, sC = M * ldc + 2 * pad;
    auto A = randv<T>(sA * batch), B = randv<T>(sB * batch), C = randv<T>(sC * batch);
    Dev<T> dA(A.size()), dB(B.size()), dC(C.size());
    dA.up(A); dB.up(B); dC.up(C);
    int st = gpk_gemm(DT<T>::v, ak, bk, M, N, K, alpha, dA.p, lda, sA, dB.p, ldb, sB, beta, dC.p, ldc, sC, batch, lower, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto got = dC.down();
    std::vector<double> ref(C.begin(), C.end());
    parallel_for(batch * M, [&](int bm) {          // (one output row each)
        const int b = bm / M, m = bm % M;
        for (int n = 0; n < N; ++n) {
            if (lower && n > m) { ref[b * sC + m * ldc + n] = (double)got[b * sC + m * ldc + n]; continue; }   // above the diagonal: unspecified
            double s = 0;
            for (int k = 0; k < K; ++k) {
                const double a = ak ? A[b * sA + m * lda + k] : A[b * sA + k * lda + m];
                const double bb = bk ? B[b * sB + n * ldb + k] : B[b * sB + k * ldb + n];
                s += a * bb;
            }
            const double c0 = (beta != 0.0) ? beta * (double)C[b * sC + m * ldc + n] : 0.0;
            ref[b * sC + m * ldc + n] = alpha * s + c0;
        }
    }, 8);
    char nm[160];
    snprintf(nm, sizeof nm, "gemm_%s %c%c M%d N%d K%d a%.0f b%.0f batch%d low%d pad%d st%d", DT<T>::name(), ak ? 'k' : 'r', bk ? 'k' : 'r', M, N, K, alpha, beta, batch, (int)lower, pad, st);
    report(nm, st ? INFINITY : relerr(got, ref), DT<T>::eps * std::sqrt((double)K + 1));
}
// GPK_GEMM_TRI_K_LOWER: A (M x K, k-major) lower triangular -> same result as the full product
template <typename T>
static void test_gemm_trilow_case(bool bk, int M, int N) {
    const int K = M;
    auto A = randv<T>((size_t)M * K), B = randv<T>((size_t)(bk ? N : K) * (bk ? K : N));
    for (int m = 0; m < M; ++m) for (int k = m + 1; k < K; ++k) A[(size_t)m * K + k] = T(0);
    Dev<T> dA(A.size()), dB(B.size()), dC((size_t)M * N);
    dA.up(A); dB.up(B);
    int st = gpk_gemm(DT<T>::v, 1, bk, M, N, K, 1.0, dA.p, K, 0, dB.p, bk ? K : N, 0, 0.0, dC.p, N, 0, 1, GPK_GEMM_TRI_K_LOWER, nullptr);
    HIPCHK(hipDeviceSynchronize());
    std::vector<double> ref((size_t)M * N, 0.0);
    parallel_for(M, [&](int m) {
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k <= m; ++k) s += (double)A[(size_t)m * K + k] * (double)(bk ? B[(size_t)n * K + k] : B[(size_t)k * N + n]);
            ref[(size_t)m * N + n] = s;
        }
    }, 4);
    char nm[160];
    snprintf(nm, sizeof nm, "gemm_trilow_%s k%c M%d N%d st%d", DT<T>::name(), bk ? 'k' : 'r', M, N, st);
    report(nm, st ? INFINITY : relerr(dC.down(), ref), DT<T>::eps * std::sqrt((double)K + 1));
}
// in-place use (C aliases A): P <- P W^T, the panel TRSM of the Cholesky; one workgroup per row tile
template <typename T>
static void test_gemm_inplace_case(int M, int64_t lda, int batch) {
    const int N = 128, K = 128;
    const int64_t sA = (int64_t)M * lda;
    auto A = randv<T>((size_t)sA * batch), W = randv<T>((size_t)N * K);
    Dev<T> dA(A.size()), dW(W.size());
    dA.up(A); dW.up(W);
    int st = gpk_gemm(DT<T>::v, 1, 1, M, N, K, 1.0, dA.p, lda, sA, dW.p, K, 0, 0.0, dA.p, lda, sA, batch, 0, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto got = dA.down();
    std::vector<double> ref(A.begin(), A.end());
    parallel_for(batch * M, [&](int bm) {
        const int b = bm / M, m = bm % M;
        for (int n = 0; n < N; ++n) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)A[b * sA + m * lda + k] * (double)W[(size_t)n * K + k];
            ref[b * sA + m * lda + n] = s;
        }
    }, 64);
    char nm[160];
    snprintf(nm, sizeof nm, "gemm_inplace_%s M%d lda%d batch%d st%d", DT<T>::name(), M, (int)lda, batch, st);
    report(nm, st ? INFINITY : relerr(got, ref), DT<T>::eps * 12);
}
template <typename T>
static void test_gemm() {
    test_gemm_inplace_case<T>(15360, 256, 1);       // 64 x 128 per workgroup (two column tiles)
    test_gemm_inplace_case<T>(3001, 131, 1);        // ragged, unaligned
    test_gemm_inplace_case<T>(1920, 130, 80);       // 128-tile path (>= 1024 tiles)
    test_gemm_trilow_case<T>(false, 512, 2048);     // 64x64 tiles
    test_gemm_trilow_case<T>(false, 300, 70);       // ragged
    test_gemm_trilow_case<T>(true, 1024, 256);
    test_gemm_trilow_case<T>(false, 1024, 8192 + 64);   // 128x128 tiles (>= 1024 of them), ragged edge
    for (int ak = 0; ak < 2; ++ak)
        for (int bk = 0; bk < 2; ++bk) {
            test_gemm_case<T>(ak, bk, 128, 128, 128, 1, 0, 1, false, 0);
            test_gemm_case<T>(ak, bk, 256, 384, 64, -1, 1, 2, false, 0);
            test_gemm_case<T>(ak, bk, 100, 70, 37, 1, 1, 1, false, 0);
            test_gemm_case<T>(ak, bk, 300, 200, 150, -1, 0, 3, false, 1);
        }
    // an empty contraction (K = 0): C = beta C, and no spinning k loop -- whole tiles on purpose (the kernels without
    // bounds checks run at least one chunk), lda = ldb = pad keeps the leading dimensions legal
    test_gemm_case<T>(true, true, 128, 128, 0, 1, 1, 1, false, 4);
    test_gemm_case<T>(true, false, 256, 128, 0, 1, 0, 2, false, 4);
    test_gemm_case<T>(true, true, 1024, 1024, 0, -1, 1, 1, false, 4);
    test_gemm_case<T>(true, true, 1024, 1024, 8, -1, 1, 1, false, 0);
    test_gemm_case<T>(true, true, 512, 512, 256, -1, 1, 1, true, 0);
    test_gemm_case<T>(true, true, 5 * 128 + 17, 5 * 128 + 17, 128, -1, 1, 2, true, 0);
    test_gemm_case<T>(true, true, 4352, 4352, 32, -1, 1, 1, true, 0);    // 34x34 tiles -> XCD super-tile path
    test_gemm_case<T>(true, false, 4224, 4224, 32, 1, 0, 1, false, 0);   // 33x33 tiles, rectangular super-tiles
    test_gemm_case<T>(true, false, 300, 3000, 40, 1, 1, 1, false, 1);    // many more tile columns than rows: column-major tile order
    test_gemm_case<T>(true, true, 260, 2100, 33, -1, 0, 2, false, 0);
    // more than one round of 128-tiles with a short last round: its tiles run as 64 x 64 quarters (ragged edges, both operand layouts)
    test_gemm_case<T>(true, true, 4100, 4100, 70, -1, 1, 1, true, 1);
    test_gemm_case<T>(false, false, 3000, 3100, 50, 1, 1, 1, false, 1);
    test_gemm_case<T>(true, false, 4100, 4100, 40, -1, 1, 1, true, 0);
    test_gemm_case<T>(false, true, 3072, 3328, 48, 1, 0, 1, false, 0);
    // ragged M, N AND K with aligned operands: the edge tiles take the pipelined loop on clamped rows, the partial last k-chunk goes
    // through the bounds-checked loads behind it (round 4); an operand stored K x M whose M is not a whole number of vectors does not
    test_gemm_case<T>(true, true, 1000, 900, 1000, -1, 1, 1, false, 0);
    test_gemm_case<T>(true, false, 1000, 904, 336, 1, 0, 1, false, 0);
    test_gemm_case<T>(false, true, 1000, 904, 336, -1, 1, 1, false, 0);
    test_gemm_case<T>(false, false, 1026, 1030, 200, 1, 1, 1, false, 2);
    test_gemm_case<T>(true, true, 777, 777, 1000, -1, 1, 1, true, 0);
}

// ----------------------------------------------------------------------------
// kernel matrix
// ----------------------------------------------------------------------------
static double kappa(int kind, double r2, double dot, double il) {
    const double q = r2 * il * il;
    switch (kind) {
        case GPK_K_EQ: return std::exp(-0.5 * q);
        case GPK_K_MATERN12: return std::exp(-std::sqrt(q));
        case GPK_K_MATERN32: { double s = std::sqrt(3 * q); return (1 + s) * std::exp(-s); }
        case GPK_K_MATERN52: { double s = std::sqrt(5 * q); return (1 + s + s * s / 3) * std::exp(-s); }
        case GPK_K_LINEAR: return dot * il * il;
        default: return 1.0;
    }
}
template <typename T>
static void test_kmat_case(std::vector<int> kinds, int n, int m, int d, int batch, bool sym, bool lower, bool acc, bool dvec) {
    const int nt = (int)kinds.size();
    std::vector<double> var(nt), il(nt);
    for (int t = 0; t < nt; ++t) { var[t] = 0.5 + 0.3 * t; il[t] = 1.0 / (0.7 + 0.2 * t); }
    if (sym) m = n;
    auto X = randv<T>((size_t)batch * n * d), Y = sym ? X : randv<T>((size_t)batch * m * d);
    const int64_t ld = m + 3;
    auto O = randv<T>((size_t)batch * n * ld);
    auto dv = randv<T>((size_t)batch * n);
    Dev<T> dX(X.size()), dY(Y.size()), dO(O.size()), dD(dv.size());
    dX.up(X); dY.up(Y); dO.up(O); dD.up(dv);
    int st = gpk_kmat(DT<T>::v, kinds.data(), var.data(), il.data(), nt, dX.p, n, d, (int64_t)n * d, sym ? dX.p : dY.p, m, d, (int64_t)m * d, d,
                      dO.p, ld, (int64_t)n * ld, batch, lower, sym, sym ? 0.25 : 0.0, dvec ? dD.p : nullptr, n, acc, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto got = dO.down();
    std::vector<double> ref(O.begin(), O.end());
    const int TN = 64 * (16 / (int)sizeof(T));
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < m; ++j) {
                if (lower && (j / TN) * TN > (i / 32) * 32 + 31) continue;
                double r2 = 0, dot = 0;
                for (int k = 0; k < d; ++k) {
                    const double a = X[((size_t)b * n + i) * d + k], c = Y[((size_t)b * m + j) * d + k];
                    r2 += (a - c) * (a - c);
                    dot += a * c;
                }
                double v = 0;
                for (int t = 0; t < nt; ++t) v += var[t] * kappa(kinds[t], r2, dot, il[t]);
                if (sym && i == j) v += 0.25 + (dvec ? (double)dv[(size_t)b * n + i] : 0.0);
                const size_t o = (size_t)b * n * ld + (size_t)i * ld + j;
                ref[o] = acc ? (double)O[o] + v : v;
            }
    char nm[160];
    snprintf(nm, sizeof nm, "kmat_%s k%d.. nt%d n%d m%d d%d b%d sym%d low%d acc%d dv%d st%d", DT<T>::name(), kinds[0], nt, n, m, d, batch, sym, lower, acc, dvec, st);
    report(nm, st ? INFINITY : relerr(got, ref), DT<T>::eps * 50);
}
// kernel values ELEMENT by element against an 80-bit reference, in units of eps * (1 + |argument of the exponential|): that
// factor is what the rounding of the squared distance (D terms, a few ulp) is amplified by; what is left measures the device's own
// exp / sqrt (a table-based exp and a shortened rsqrt iteration since round 5).  Scales from near-duplicates to e^-600.
template <typename T>
static void test_kmat_ulp(int kind, double scale, double ilv) {
    const int n = 192, m = 320, d = 8;
    auto X = randv<T>((size_t)n * d, scale), Y = randv<T>((size_t)m * d, scale);
    for (int k = 0; k < d; ++k) Y[k] = X[k] * (T)(1 + 64 * (double)std::numeric_limits<T>::epsilon());                 // one near-duplicate pair, one exact duplicate
    for (int k = 0; k < d; ++k) Y[d + k] = X[d + k];
    Dev<T> dX(X.size()), dY(Y.size()), dO((size_t)n * m);
    dX.up(X); dY.up(Y);
    double var = 1.3, il = ilv;
    int st = gpk_kmat(DT<T>::v, &kind, &var, &il, 1, dX.p, n, d, 0, dY.p, m, d, 0, d, dO.p, m, 0, 1, 0, 0, 0.0, nullptr, 0, 0, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto got = dO.down();
    double worst = 0, amin = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            long double r2 = 0;
            for (int k = 0; k < d; ++k) { const long double df = (long double)X[(size_t)i * d + k] - (long double)Y[(size_t)j * d + k]; r2 += df * df; }
            const long double q = r2 * (long double)il * (long double)il;
            long double a, v;
            if (kind == GPK_K_EQ) { a = -0.5L * q; v = expl(a); }
            else if (kind == GPK_K_MATERN12) { a = -sqrtl(q); v = expl(a); }
            else if (kind == GPK_K_MATERN32) { a = -sqrtl(3 * q); v = (1 - a) * expl(a); }
            else { a = -sqrtl(5 * q); v = (1 - a + a * a / 3) * expl(a); }
            v *= (long double)var;
            if (a < (sizeof(T) == 8 ? -700 : -85)) continue;      // (denormal results: absolute, not relative, accuracy there)
            const double e = (double)(fabsl((long double)got[(size_t)i * m + j] - v) / v) / ((double)std::numeric_limits<T>::epsilon() * (1.0 + (double)-a));
            worst = std::max(worst, e);
            amin = std::min(amin, (double)a);
        }
    char nm[160];
    snprintf(nm, sizeof nm, "kmat_%s kind%d elementwise, exp arguments down to %.0f: worst error / (eps (1 + |arg|)) st%d", DT<T>::name(), kind, amin, st);
    report(nm, st ? INFINITY : worst, 6.0);
}
template <typename T>
static void test_kmat() {
    for (int kind : {GPK_K_EQ, GPK_K_MATERN12, GPK_K_MATERN32, GPK_K_MATERN52}) {
        test_kmat_ulp<T>(kind, 1.0, 0.9);
        test_kmat_ulp<T>(kind, 1.0, (sizeof(T) == 8 ? 1.0 : 0.35) * (kind == GPK_K_EQ ? 6.0 : 60.0));
        test_kmat_ulp<T>(kind, 1e-3, 1.0);
    }
    for (int k = 0; k <= 5; ++k) test_kmat_case<T>({k}, 70, 45, 3, 1, false, false, false, false);
    test_kmat_case<T>({GPK_K_EQ}, 300, 300, 8, 2, true, false, false, true);
    test_kmat_case<T>({GPK_K_EQ}, 700, 700, 8, 1, true, true, false, false);
    test_kmat_case<T>({GPK_K_EQ, GPK_K_LINEAR}, 257, 129, 4, 1, false, false, true, false);
    test_kmat_case<T>({GPK_K_MATERN52, GPK_K_MATERN32, GPK_K_CONST}, 64, 512, 20, 1, false, false, false, false);
    test_kmat_case<T>({GPK_K_EQ}, 33, 33, 1, 3, true, false, false, false);
    // kdiag
    {
        const int n = 100, d = 5;
        std::vector<int> kinds = {GPK_K_EQ, GPK_K_LINEAR};
        std::vector<double> var = {0.7, 1.3}, il = {1.0, 0.5};
        auto X = randv<T>(n * d);
        Dev<T> dX(X.size()), dO(n);
        dX.up(X);
        int st = gpk_kdiag(DT<T>::v, kinds.data(), var.data(), il.data(), 2, dX.p, n, d, n * d, d, dO.p, n, 1, nullptr);
        HIPCHK(hipDeviceSynchronize());
        std::vector<double> ref(n);
        for (int i = 0; i < n; ++i) { double nr = 0; for (int k = 0; k < d; ++k) nr += (double)X[i * d + k] * X[i * d + k]; ref[i] = 0.7 + 1.3 * 0.25 * nr; }
        report(std::string("kdiag_") + DT<T>::name(), st ? INFINITY : relerr(dO.down(), ref), DT<T>::eps * 50);
    }
}

// ----------------------------------------------------------------------------
// Cholesky + solves
// ----------------------------------------------------------------------------
template <typename T>
static std::vector<T> make_spd(int n, int batch, int64_t ld) {
    // EQ kernel on random 3-d points + 0.5 I  (condition number modest)
    std::vector<T> A((size_t)batch * n * ld, (T)0);
    for (int b = 0; b < batch; ++b) {
        auto X = randv<double>((size_t)n * 3);
        auto fill_row = [&](int i) {
            for (int j = 0; j <= i; ++j) {
                double r2 = 0;
                for (int k = 0; k < 3; ++k) { double df = X[i * 3 + k] - X[j * 3 + k]; r2 += df * df; }
                double v = std::exp(-0.5 * r2) + (i == j ? 0.5 : 0.0);
                A[(size_t)b * n * ld + (size_t)i * ld + j] = (T)v;
                A[(size_t)b * n * ld + (size_t)j * ld + i] = (T)v;
            }
        };
        if (n >= 1024) parallel_for(n, fill_row);       // (row i writes the pairs (i, j <= i): no two rows share an element)
        else for (int i = 0; i < n; ++i) fill_row(i);
    }
    return A;
}
// L[i][j] = (A[i][j] - sum_{k<j} L[i][k] L[j][k]) / L[j][j], every sum over ascending k (the bits of the plain column-by-column loop,
// whatever the schedule).  Row i's entries in the columns [j0, min(j1, i + 1)) of a column block: eight sums at a time over the columns
// in front of the block -- independent chains for the core to overlap -- then each entry's few terms inside the block.
static void host_chol_row(double* A, int64_t ld, int i, int j0, int j1) {
    double* Li = A + (int64_t)i * ld;
    const int je = std::min(j1, i + 1);
    for (int jb = j0; jb < je; jb += 8) {
        const int nj = std::min(8, je - jb);
        double s[8];
        const double* Lj[8];
        for (int q = 0; q < 8; ++q) {
            const int j = jb + std::min(q, nj - 1);
            Lj[q] = A + (int64_t)j * ld;
            s[q] = Li[j];
        }
        for (int k = 0; k < j0; ++k) {
            const double a = Li[k];
            s[0] -= a * Lj[0][k]; s[1] -= a * Lj[1][k]; s[2] -= a * Lj[2][k]; s[3] -= a * Lj[3][k];
            s[4] -= a * Lj[4][k]; s[5] -= a * Lj[5][k]; s[6] -= a * Lj[6][k]; s[7] -= a * Lj[7][k];
        }
        for (int q = 0; q < nj; ++q) {
            const int j = jb + q;
            double t = s[q];
            for (int k = j0; k < j; ++k) t -= Li[k] * Lj[q][k];
            Li[j] = (i == j) ? std::sqrt(t) : t / Lj[q][j];
        }
    }
}
// column blocks of 32: the block's own rows (serial), then every row below it on its own (`par`: spread over the host's cores)
static void host_chol(double* A, int n, int64_t ld, bool par) {
    constexpr int W = 32;
    for (int j0 = 0; j0 < n; j0 += W) {
        const int j1 = std::min(n, j0 + W);
        for (int i = j0; i < j1; ++i) host_chol_row(A, ld, i, j0, j1);
        if (par && n - j1 >= 1024) parallel_for(n - j1, [&](int r) { host_chol_row(A, ld, j1 + r, j0, j1); }, 32);
        else for (int i = j1; i < n; ++i) host_chol_row(A, ld, i, j0, j1);
    }
}
template <typename T>
static void test_potrf_case(int n, int batch, int nbo, int nrhs_small, int nrhs_big, int sb) {
    const int64_t ld = n + (n % 2);   // keep rows 16-byte aligned for f64, exercise ld != n
    auto A = make_spd<T>(n, batch, ld);
    const int64_t sA = (int64_t)n * ld;
    const int64_t de = gpk_dinv_elems(n);
    Dev<T> dA(A.size()), dinv((size_t)de * batch);
    Dev<int> info(batch);
    dA.up(A); info.zero();
    int st = gpk_potrf(DT<T>::v, dA.p, n, ld, sA, batch, dinv.p, info.p, nbo, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto L = dA.down();
    auto inf = info.down();
    // host reference factor
    std::vector<double> Lref(A.begin(), A.end());
    if (batch == 1) host_chol(Lref.data(), n, ld, true);
    else parallel_for(batch, [&](int b) { host_chol(Lref.data() + b * sA, n, ld, false); });      // (the members of a batch are independent)
    double num = 0, den = 0;
    bool finite = true;
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                const double g = L[b * sA + i * ld + j], r = Lref[b * sA + i * ld + j];
                if (!std::isfinite(g)) finite = false;
                num = std::max(num, std::fabs(g - r));
                den = std::max(den, std::fabs(r));
            }
    char nm[160];
    snprintf(nm, sizeof nm, "potrf_%s n%d batch%d nbo%d st%d info%d", DT<T>::name(), n, batch, nbo, st, inf[0]);
    report(nm, (st || !finite || inf[0]) ? INFINITY : num / den, DT<T>::eps * 100);

    // dinv check: inv(L_cc) * L_cc = I on the first and last diagonal block of batch 0
    {
        auto W = dinv.down();
        const int nblk = (n + 127) / 128;
        double worst = 0;
        for (int blk : {0, nblk - 1}) {
            const int o = blk * 128, nv = std::min(128, n - o);
            for (int i = 0; i < nv; ++i)
                for (int j = 0; j < nv; ++j) {
                    double s = 0;
                    for (int k = 0; k < nv; ++k) s += (double)W[(size_t)blk * 16384 + i * 128 + k] * Lref[(o + k) * ld + o + j] * (k >= j ? 1.0 : 0.0);
                    worst = std::max(worst, std::fabs(s - (i == j ? 1.0 : 0.0)));
                }
        }
        snprintf(nm, sizeof nm, "dinv128_%s n%d", DT<T>::name(), n);
        report(nm, worst, DT<T>::eps * 1000);
    }

    // logdet
    {
        Dev<T> out(batch);
        int s2 = gpk_logdet_chol(DT<T>::v, dA.p, n, ld, sA, batch, out.p, nullptr);
        HIPCHK(hipDeviceSynchronize());
        std::vector<double> ref(batch, 0.0);
        for (int b = 0; b < batch; ++b) for (int i = 0; i < n; ++i) ref[b] += 2 * std::log(Lref[b * sA + i * ld + i]);
        snprintf(nm, sizeof nm, "logdet_%s n%d st%d", DT<T>::name(), n, s2);
        report(nm, s2 ? INFINITY : relerr(out.down(), ref), DT<T>::eps * 100);
    }

    // merged inverses + solves
    Dev<T> dsb((size_t)batch * ((n + sb - 1) / sb) * sb * sb), tmpm((size_t)((n + sb - 1) / sb) * sb * sb / 4 + 16);
    int s3 = gpk_trtri_merge(DT<T>::v, dA.p, n, ld, sA, batch, dinv.p, sb, dsb.p, tmpm.p, nullptr);
    HIPCHK(hipDeviceSynchronize());
    for (int pass = 0; pass < 2; ++pass) {
        const int nrhs = pass == 0 ? nrhs_small : nrhs_big;
        if (nrhs <= 0) continue;
        const int64_t ldb = nrhs + (pass == 0 ? 0 : 2);
        auto Bm = randv<T>((size_t)batch * n * ldb);
        Dev<T> dB(Bm.size()), tmp((size_t)batch * sb * nrhs + GPK_TRSV_CTRL_ELEMS);
        dB.up(Bm);
        int s4 = pass == 0 ? gpk_trsv_lower(DT<T>::v, dA.p, n, ld, sA, dsb.p, sb, dB.p, nrhs, ldb, (int64_t)n * ldb, tmp.p, batch, nullptr)
                           : gpk_trsm_lower(DT<T>::v, dA.p, n, ld, sA, dsb.p, sb, dB.p, nrhs, ldb, (int64_t)n * ldb, tmp.p, batch, nullptr);
        HIPCHK(hipDeviceSynchronize());
        std::vector<double> ref(Bm.begin(), Bm.end());
        parallel_for(batch * nrhs, [&](int bc) {          // (column c of member b touches only its own entries of `ref`)
            const int b = bc / nrhs, c = bc % nrhs;
            for (int i = 0; i < n; ++i) {
                double s = ref[((size_t)b * n + i) * ldb + c];
                for (int k = 0; k < i; ++k) s -= Lref[b * sA + i * ld + k] * ref[((size_t)b * n + k) * ldb + c];
                ref[((size_t)b * n + i) * ldb + c] = s / Lref[b * sA + i * ld + i];
            }
        });
        // compare only the nrhs columns
        auto got = dB.down();
        double nu = 0, de2 = 0;
        for (int b = 0; b < batch; ++b) for (int i = 0; i < n; ++i) for (int c = 0; c < nrhs; ++c) {
            const size_t o = ((size_t)b * n + i) * ldb + c;
            if (!std::isfinite((double)got[o])) nu = INFINITY;
            nu = std::max(nu, std::fabs((double)got[o] - ref[o]));
            de2 = std::max(de2, std::fabs(ref[o]));
        }
        snprintf(nm, sizeof nm, "%s_%s n%d nrhs%d sb%d batch%d st%d/%d", pass == 0 ? "trsv" : "trsm", DT<T>::name(), n, nrhs, sb, batch, s3, s4);
        report(nm, (s3 || s4) ? INFINITY : nu / de2, DT<T>::eps * 1000);
        if (pass == 1) {   // the same solve out of place: the solution goes to a second buffer (its own leading dimension), b is workspace
            const int64_t ldx = nrhs + 1;
            Dev<T> dX((size_t)batch * n * ldx);
            dX.up(std::vector<T>((size_t)batch * n * ldx, T(7)));
            dB.up(Bm);
            const int s5 = gpk_trsm_lower_to(DT<T>::v, dA.p, n, ld, sA, dsb.p, sb, dB.p, nrhs, ldb, (int64_t)n * ldb, dX.p, ldx, (int64_t)n * ldx, batch, nullptr);
            HIPCHK(hipDeviceSynchronize());
            auto gx = dX.down();
            double nu2 = 0;
            for (int b = 0; b < batch; ++b) for (int i = 0; i < n; ++i) {
                for (int c = 0; c < nrhs; ++c) {
                    const double g = (double)gx[((size_t)b * n + i) * ldx + c];
                    if (!std::isfinite(g)) nu2 = INFINITY;
                    nu2 = std::max(nu2, std::fabs(g - ref[((size_t)b * n + i) * ldb + c]));
                }
                if ((double)gx[((size_t)b * n + i) * ldx + nrhs] != 7.0) nu2 = INFINITY;      // the padding column of x is not touched
            }
            snprintf(nm, sizeof nm, "trsm_to_%s n%d nrhs%d sb%d batch%d st%d", DT<T>::name(), n, nrhs, sb, batch, s5);
            report(nm, s5 ? INFINITY : nu2 / de2, DT<T>::eps * 1000);
        }
    }
}
template <typename T>
static void test_potrf() {
    // the single-launch sweep for one right-hand side (trsv_sweep_kernel): orders from 2048 take it by default; ragged last blocks,
    // every block width it serves, and (dev build) small orders with the threshold lowered
    gpk_tune(49, 1);           // (off by default since it measured slower than the per-block sweep: profiles/r05_ab_trsv_sweep.log)
    test_potrf_case<T>(2500, 1, 0, 1, 0, 512);
    test_potrf_case<T>(4500, 1, 0, 1, 0, 1024);
    test_potrf_case<T>(2304, 1, 0, 1, 0, 256);
    test_potrf_case<T>(4096, 1, 0, 1, 0, 2048);
    gpk_tune(50, 0);
    test_potrf_case<T>(700, 1, 0, 1, 0, 256);
    test_potrf_case<T>(1664, 1, 256, 1, 0, 512);
    gpk_tune(50, 2048);
    gpk_tune(49, 0);
    test_potrf_case<T>(10, 1, 0, 1, 3, 128);
    test_potrf_case<T>(100, 3, 0, 2, 0, 128);
    test_potrf_case<T>(128, 1, 0, 1, 130, 128);
    test_potrf_case<T>(200, 2, 0, 3, 64, 128);
    test_potrf_case<T>(256, 1, 128, 1, 0, 256);
    test_potrf_case<T>(384, 1, 256, 8, 200, 256);
    test_potrf_case<T>(500, 2, 256, 1, 129, 512);
    test_potrf_case<T>(1024, 1, 256, 1, 256, 512);
    test_potrf_case<T>(1200, 1, 512, 4, 100, 512);
    test_potrf_case<T>(1664, 1, 256, 1, 300, 256);
    test_potrf_case<T>(700, 70, 256, 1, 0, 128);     // larger batch, ragged
    test_potrf_case<T>(512, 64, 128, 2, 0, 128);
    // the mixed-phase batched steps (batch >= 64, orders that are multiples of 128 from 512 on; fp32 by default, knob 53 = 2: fp64 too):
    // every halving level, a partial last outer panel, batches that do not divide by the 8 queues
    gpk_tune(53, 2);
    test_potrf_case<T>(1024, 64, 256, 1, 0, 128);
    test_potrf_case<T>(640, 67, 256, 1, 0, 128);
    test_potrf_case<T>(1536, 65, 512, 1, 0, 128);
    test_potrf_case<T>(2048, 72, 0, 1, 0, 128);
    test_potrf_case<T>(1152, 64, 1024, 1, 0, 128);
    gpk_tune(53, 1);
}


// ----------------------------------------------------------------------------
// persistent two-problem update + look-ahead Cholesky
// ----------------------------------------------------------------------------
template <typename T>
static void test_update2_case(int M0, int N0, int M1, int K, int reserve, int pad) {
    // segment 0: rectangular, OUT OF PLACE (cin != c); segment 1: square, lower tiles only, in place
    const int64_t lda0 = K + pad, ldb0 = K + pad, ldcin = N0 + pad, ldc0 = N0 + 2 * pad, lda1 = K + pad, ldc1 = M1 + pad;
    auto A0 = randv<T>((size_t)M0 * lda0), B0 = randv<T>((size_t)N0 * ldb0), Cin = randv<T>((size_t)M0 * ldcin);
    auto A1 = randv<T>((size_t)M1 * lda1), C1 = randv<T>((size_t)M1 * ldc1);
    std::vector<T> C0((size_t)M0 * ldc0, (T)7);
    Dev<T> dA0(A0.size()), dB0(B0.size()), dCin(Cin.size()), dC0(C0.size()), dA1(A1.size()), dC1(C1.size());
    Dev<unsigned> ctrl(64);
    dA0.up(A0); dB0.up(B0); dCin.up(Cin); dC0.up(C0); dA1.up(A1); dC1.up(C1);
    gpk_update_t u[2];
    u[0] = gpk_update_t{M0, N0, K, dA0.p, lda0, dB0.p, ldb0, dCin.p, ldcin, dC0.p, ldc0, 0};
    u[1] = gpk_update_t{M1, M1, K, dA1.p, lda1, dA1.p, lda1, dC1.p, ldc1, dC1.p, ldc1, 1};
    const int st = gpk_gemm_update2(DT<T>::v, u, 2, -1.0, ctrl.p, reserve, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto g0 = dC0.down(), g1 = dC1.down();
    double num = 0, den = 0;
    bool ok = true;
    for (int i = 0; i < M0; ++i)
        for (int j = 0; j < N0; ++j) {
            double r = Cin[(size_t)i * ldcin + j];
            for (int k = 0; k < K; ++k) r -= (double)A0[(size_t)i * lda0 + k] * (double)B0[(size_t)j * ldb0 + k];
            const double g = g0[(size_t)i * ldc0 + j];
            if (!std::isfinite(g)) ok = false;
            num = std::max(num, std::fabs(g - r)); den = std::max(den, std::fabs(r));
        }
    const int ts = 64;   // tiles strictly above the diagonal (at the finest tile size) must be untouched
    for (int i = 0; i < M1; ++i)
        for (int j = 0; j < M1; ++j) {
            double r = C1[(size_t)i * ldc1 + j];
            const double g = g1[(size_t)i * ldc1 + j];
            if (j <= i) {
                for (int k = 0; k < K; ++k) r -= (double)A1[(size_t)i * lda1 + k] * (double)A1[(size_t)j * lda1 + k];
            } else if (j / 128 > i / 128 || (j / ts > i / ts && g == r)) {
                // untouched (above the tile diagonal for whichever tile size ran)
            } else {
                for (int k = 0; k < K; ++k) r -= (double)A1[(size_t)i * lda1 + k] * (double)A1[(size_t)j * lda1 + k];
            }
            if (!std::isfinite(g)) ok = false;
            num = std::max(num, std::fabs(g - r)); den = std::max(den, std::fabs(r));
        }
    char nm[160];
    snprintf(nm, sizeof nm, "update2_%s %dx%d+%d^2 k%d reserve%d pad%d st%d", DT<T>::name(), M0, N0, M1, K, reserve, pad, st);
    report(nm, (st || !ok) ? INFINITY : num / den, DT<T>::eps * 100);
}

template <typename T>
static void test_potrf_la_case(int n, int nb, int mode, int64_t min_rows, int64_t tail_rows = 0, int sb = 0) {
    const int64_t ld = n + (n % 2);
    auto A = make_spd<T>(n, 1, ld);
    const int64_t de = gpk_dinv_elems(n);
    const int wb = sb > 0 ? sb : nb;           // width of the explicit inverses
    const int nblk = (n + wb - 1) / wb;
    Dev<T> dA(A.size()), dRef(A.size()), dinv((size_t)de), dinv2((size_t)de), dbig((size_t)nblk * wb * wb), ws((size_t)gpk_potrf_la_ws_elems(n, nb));
    Dev<int> info(1), info2(1);
    dA.up(A); dRef.up(A); info.zero(); info2.zero();
    gpk_tune(7, mode); gpk_tune(6, min_rows); gpk_tune(9, tail_rows);
    const int st = sb > 0 ? gpk_potrf_la_split(DT<T>::v, dA.p, n, ld, dinv.p, dbig.p, nb, sb, ws.p, info.p, nullptr)
                          : gpk_potrf_la(DT<T>::v, dA.p, n, ld, dinv.p, dbig.p, nb, ws.p, info.p, nullptr);
    gpk_tune(7, 1); gpk_tune(6, 2048); gpk_tune(9, 0);
    const int st2 = gpk_potrf(DT<T>::v, dRef.p, n, ld, 0, 1, dinv2.p, info2.p, 0, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto L = dA.down(), R = dRef.down();
    double num = 0, den = 0;
    bool finite = true;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            const double g = L[(size_t)i * ld + j], r = R[(size_t)i * ld + j];
            if (!std::isfinite(g)) finite = false;
            num = std::max(num, std::fabs(g - r)); den = std::max(den, std::fabs(r));
        }
    char nm[160];
    snprintf(nm, sizeof nm, "potrf_la_%s n%d nb%d sb%d mode%d minrows%lld tail%lld st%d/%d info%d", DT<T>::name(), n, nb, wb, mode, (long long)min_rows, (long long)tail_rows, st, st2, info.down()[0]);
    report(nm, (st || st2 || !finite || info.down()[0]) ? INFINITY : num / den, DT<T>::eps * 100);
    // upper triangle untouched
    {
        double worst = 0;
        for (int i = 0; i < n; ++i) for (int j = (i / 128 + 1) * 128; j < n; ++j) worst = std::max(worst, std::fabs((double)L[(size_t)i * ld + j] - (double)A[(size_t)i * ld + j]));
        snprintf(nm, sizeof nm, "potrf_la_%s n%d above the 128-block diagonal untouched", DT<T>::name(), n);
        report(nm, worst, 0.0);
    }
    // merged inverses: W_j L_jj = I on the first and the last block; 128-block inverses equal the plain path's
    {
        auto W = dbig.down();
        double worst = 0;
        for (int blk : {0, nblk / 2, nblk - 1}) {
            const int o = blk * wb, nv = std::min(wb, n - o);
            for (int i = 0; i < nv; i += 7)
                for (int j = 0; j < nv; ++j) {
                    double sacc = 0;
                    for (int k = j; k <= i; ++k) sacc += (double)W[(size_t)blk * wb * wb + (size_t)i * wb + k] * (double)R[(size_t)(o + k) * ld + o + j];
                    worst = std::max(worst, std::fabs(sacc - (i == j ? 1.0 : 0.0)));
                }
        }
        snprintf(nm, sizeof nm, "potrf_la_%s n%d nb%d dinv_nb", DT<T>::name(), n, nb);
        report(nm, worst, DT<T>::eps * 1000);
        auto d1 = dinv.down(), d2 = dinv2.down();
        double nu = 0, dn = 0;
        for (size_t i = 0; i < d1.size(); ++i) { nu = std::max(nu, std::fabs((double)d1[i] - (double)d2[i])); dn = std::max(dn, std::fabs((double)d2[i])); }
        snprintf(nm, sizeof nm, "potrf_la_%s n%d nb%d dinv128", DT<T>::name(), n, nb);
        report(nm, nu / dn, DT<T>::eps * 1000);
    }
}

// factorisation with rows under the matrix (gpk_potrf_rows): the factor against the plain path's, the extra rows against E L^{-T}
// by substitution on the host
template <typename T>
static void test_potrf_rows_case(int n, int extra, int nb, int sb, int64_t tail, int agg = 2, int flags = 0) {
    // flags & 1: the last 64 rows are the right-hand side's strip -- `extra` counts the rows in front of it plus the vector itself
    const int strip = (flags & 1) ? GPK_ROWS_RHS_STRIP : 0;
    const int64_t ld = n, rows = n + extra + (strip ? strip - 1 : 0);
    auto A = make_spd<T>(n, 1, ld);
    auto E = randv<T>((size_t)(rows - n) * ld);
    for (size_t i = (size_t)extra * ld; i < E.size(); ++i) E[i] = T(0);      // (the strip's padding)
    std::vector<T> full((size_t)rows * ld);
    std::copy(A.begin(), A.end(), full.begin());
    std::copy(E.begin(), E.end(), full.begin() + (size_t)n * ld);
    const int wb = sb > 0 ? sb : (nb > 0 ? nb : 128);
    const int nblk = (n + wb - 1) / wb;
    Dev<T> dA(full.size()), dRef(A.size()), dinv((size_t)gpk_dinv_elems(n) + 128 * 128), dinv2((size_t)gpk_dinv_elems(n)), dbig((size_t)nblk * wb * wb),
        ws((size_t)(nb > 0 ? gpk_potrf_la_ws_elems(rows, nb) : 16));
    Dev<int> info(1), info2(1);
    dA.up(full); dRef.up(A); info.zero(); info2.zero();
    gpk_tune(9, tail); gpk_tune(47, agg);
    // flags: the last row as a right-hand side (round 6) -- the same numbers are expected in it
    const int st = flags ? gpk_potrf_rows_rhs(DT<T>::v, dA.p, n, rows, ld, dinv.p, dbig.p, nb, sb, ws.p, info.p, flags, nullptr)
                         : gpk_potrf_rows(DT<T>::v, dA.p, n, rows, ld, dinv.p, dbig.p, nb, sb, ws.p, info.p, nullptr);
    gpk_tune(9, 0); gpk_tune(47, 2);
    const int st2 = gpk_potrf(DT<T>::v, dRef.p, n, ld, 0, 1, dinv2.p, info2.p, 0, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto G = dA.down(), R = dRef.down();
    double num = 0, den = 0, znum = 0, zden = 0;
    bool finite = true;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            const double g = G[(size_t)i * ld + j], r = R[(size_t)i * ld + j];
            if (!std::isfinite(g)) finite = false;
            num = std::max(num, std::fabs(g - r)); den = std::max(den, std::fabs(r));
        }
    std::vector<double> znum_e(extra, 0.0), zden_e(extra, 0.0);
    std::vector<char> bad_e(extra, 0);
    parallel_for(extra, [&](int e) {            // z L^T = E[e]:  z_j = (E[e][j] - sum_{k<j} z_k L[j][k]) / L[j][j]   (the rows are independent)
        std::vector<double> z(n);
        for (int j = 0; j < n; ++j) {
            double acc = E[(size_t)e * ld + j];
            for (int k = 0; k < j; ++k) acc -= z[k] * (double)R[(size_t)j * ld + k];
            z[j] = acc / (double)R[(size_t)j * ld + j];
        }
        double zn = 0, zd = 0;
        char bad = 0;
        for (int j = 0; j < n; ++j) {
            const double g = G[(size_t)(n + e) * ld + j];
            if (!std::isfinite(g)) bad = 1;
            zn = std::max(zn, std::fabs(g - z[j])); zd = std::max(zd, std::fabs(z[j]));
        }
        znum_e[e] = zn; zden_e[e] = zd; bad_e[e] = bad;
    });
    for (int e = 0; e < extra; ++e) {
        if (bad_e[e]) finite = false;
        znum = std::max(znum, znum_e[e]); zden = std::max(zden, zden_e[e]);
    }
    char nm[200];
    snprintf(nm, sizeof nm, "potrf_rows_%s n%d +%d rows nb%d sb%d tail%lld agg%d flags%d st%d/%d info%d: factor", DT<T>::name(), n, extra, nb, wb, (long long)tail, agg, flags, st, st2, info.down()[0]);
    report(nm, (st || st2 || !finite || info.down()[0]) ? INFINITY : num / den, DT<T>::eps);
    snprintf(nm, sizeof nm, "potrf_rows_%s n%d +%d rows nb%d sb%d tail%lld agg%d flags%d: rows = E L^-T", DT<T>::name(), n, extra, nb, wb, (long long)tail, agg, flags);
    report(nm, (st || !finite) ? INFINITY : znum / zden, DT<T>::eps * 2);
}

// batched factorisation with one right-hand side per matrix solved along (gpk_potrf_rhs): factors bit-identical to gpk_potrf's,
// the solved vectors against gpk_trsv_lower on those factors
template <typename T>
static void test_potrf_rhs_case(int n, int batch) {
    const int64_t ld = n, bs = (int64_t)n * ld;
    auto A = make_spd<T>(n, batch, ld);
    auto Bv = randv<T>((size_t)batch * n);
    Dev<T> dA(A.size()), dR(A.size()), dinv((size_t)batch * gpk_dinv_elems(n)), dinv2((size_t)batch * gpk_dinv_elems(n)), dB(Bv.size()), dB2(Bv.size()),
        tmp((size_t)batch * 128 + 16), tmp2((size_t)batch * 128 + 16);
    Dev<int> info(batch), info2(batch);
    dA.up(A); dR.up(A); dB.up(Bv); dB2.up(Bv); info.zero(); info2.zero();
    const int st = gpk_potrf_rhs(DT<T>::v, dA.p, n, ld, bs, batch, dinv.p, info.p, 0, dB.p, n, tmp.p, nullptr);
    const int st2 = gpk_potrf(DT<T>::v, dR.p, n, ld, bs, batch, dinv2.p, info2.p, 0, nullptr);
    const int st3 = gpk_trsv_lower(DT<T>::v, dR.p, n, ld, bs, dinv2.p, 128, dB2.p, 1, 1, n, tmp2.p, batch, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto G = dA.down(), R = dR.down();
    auto x = dB.down(), x2 = dB2.down();
    size_t diff = 0;
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) diff += G[(size_t)b * bs + (size_t)i * ld + j] != R[(size_t)b * bs + (size_t)i * ld + j];
    double num = 0, den = 0;
    bool finite = true;
    for (size_t i = 0; i < x.size(); ++i) {
        if (!std::isfinite((double)x[i])) finite = false;
        num = std::max(num, std::fabs((double)x[i] - (double)x2[i])); den = std::max(den, std::fabs((double)x2[i]));
    }
    int bad = 0;
    for (int v : info.down()) bad += v != 0;
    char nm[200];
    snprintf(nm, sizeof nm, "potrf_rhs_%s n%d batch%d st%d/%d/%d bad_info%d: factors differ in %zu entries", DT<T>::name(), n, batch, st, st2, st3, bad, diff);
    report(nm, (st || st2 || st3 || bad || diff) ? INFINITY : 0.0, DT<T>::eps);
    snprintf(nm, sizeof nm, "potrf_rhs_%s n%d batch%d: L^-1 b against gpk_trsv_lower", DT<T>::name(), n, batch);
    report(nm, (st || !finite) ? INFINITY : num / den, DT<T>::eps * 200);
}

template <typename T>
static void test_potrf_rhs() {
    test_potrf_rhs_case<T>(512, 64);      // mixed-phase steps (fp32): the sweep's steps on the side stream
    test_potrf_rhs_case<T>(1024, 70);
    test_potrf_rhs_case<T>(2048, 64);
    test_potrf_rhs_case<T>(640, 9);       // lockstep launches: the sweep behind the factorisation
    test_potrf_rhs_case<T>(300, 5);       // ragged order
    test_potrf_rhs_case<T>(1024, 1);
}

template <typename T>
static void test_potrf_rows() {
    test_potrf_rows_case<T>(128, 64, 0, 0, 0);             // one diagonal block + rows
    test_potrf_rows_case<T>(640, 200, 0, 0, 0);            // one pipelined panel
    test_potrf_rows_case<T>(1024, 300, 0, 0, 0);
    test_potrf_rows_case<T>(2560, 333, 0, 0, 0);
    test_potrf_rows_case<T>(5120, 260, 0, 0, 0);           // several panels (1024 wide), fill tiles with rows under the square part
    test_potrf_rows_case<T>(3072, 300, 512, 0, 1024);      // look-ahead + plain tail
    test_potrf_rows_case<T>(3072, 300, 512, 256, 1024);    // ... explicit inverses narrower than the outer blocks
    test_potrf_rows_case<T>(4096, 520, 1024, 0, 2048, 1);
    test_potrf_rows_case<T>(4096, 130, 256, 0, 512, 3);
    test_potrf_rows_case<T>(3840, 1000, 256, 0, 100, 2);   // tail shorter than a block: raised to one block
    test_potrf_rows_case<T>(2048, 77, 256, 0, 0);          // n <= the default tail: all plain
    // orders that are whole 128-blocks but not whole panels: a last panel of one block, 17 / 33 blocks, the look-ahead's last outer
    // block 128 wide with a plain tail whose last panel is one block
    test_potrf_rows_case<T>(2176, 130, 0, 0, 0);
    test_potrf_rows_case<T>(4224, 64, 0, 0, 0);
    test_potrf_rows_case<T>(5248, 200, 0, 0, 0);
    test_potrf_rows_case<T>(4224, 96, 1024, 0, 2048);
    test_potrf_rows_case<T>(7296, 64, 1024, 512, 3072);
    // round 6: the last row as a right-hand side vector (side stream through the look-ahead steps, a row in the tail)
    test_potrf_rows_case<T>(640, 65, 0, 0, 0, 2, 1);               // pipelined panel: a row like the others
    test_potrf_rows_case<T>(3072, 301, 512, 0, 1024, 2, 1);
    test_potrf_rows_case<T>(3072, 65, 512, 256, 1024, 2, 3);       // inverses narrower than the outer blocks, no tail inverses
    test_potrf_rows_case<T>(4096, 1, 1024, 0, 2048, 1, 1);         // the right-hand side alone
    test_potrf_rows_case<T>(7296, 129, 1024, 512, 3072, 2, 3);
    test_potrf_rows_case<T>(2048, 78, 256, 0, 0, 2, 3);            // all plain
}

template <typename T>
static void test_lookahead() {
    for (int pass = 0; pass < 2; ++pass) {      // 128-tile kernels forced, then 64-tile kernels forced
        gpk_tune(1, pass == 0 ? 0 : ((int64_t)1 << 40));
        test_update2_case<T>(300, 200, 260, 70, 0, 0);
        test_update2_case<T>(384, 128, 512, 64, 1, 0);
        test_update2_case<T>(1000, 256, 1100, 128, 1, 2);
        test_update2_case<T>(129, 1, 5, 3, 1, 1);
    }
    gpk_tune(1, 512);
    test_potrf_la_case<T>(700, 256, 1, 0);
    test_potrf_la_case<T>(1024, 256, 1, 0);
    test_potrf_la_case<T>(1200, 512, 1, 0);
    test_potrf_la_case<T>(1664, 256, 0, 0);
    test_potrf_la_case<T>(3000, 512, 1, 0);
    test_potrf_la_case<T>(3000, 1024, 1, 1024);
    test_potrf_la_case<T>(4096, 1024, 1, 0);
    test_potrf_la_case<T>(5000, 512, 1, 2048);
    test_potrf_la_case<T>(4096, 1024, 1, 0, 0, 512);       // explicit inverses narrower than the outer blocks (gpk_potrf_la_split)
    test_potrf_la_case<T>(5000, 1024, 1, 1024, 1000, 256);
    test_potrf_la_case<T>(3000, 512, 1, 0, 700, 256);
    test_potrf_la_case<T>(2100, 1024, 0, 0, 300, 512);
    gpk_tune(40, 256); gpk_tune(41, 4096);              // the next diagonal block's update inside the trailing update, at these small orders and every block width
    test_potrf_la_case<T>(4096, 1024, 1, 0);
    test_potrf_la_case<T>(5000, 512, 1, 1024);
    test_potrf_la_case<T>(6000, 256, 1, 0, 700);
    gpk_tune(40, 9216); gpk_tune(41, 512);
    test_potrf_la_case<T>(10, 256, 1, 0, 5120);        // all plain + merge
    test_potrf_la_case<T>(1200, 512, 1, 0, 5120);
    test_potrf_la_case<T>(1664, 256, 1, 0, 700);       // look-ahead, then a plain tail
    test_potrf_la_case<T>(3000, 512, 1, 0, 1500);
    test_potrf_la_case<T>(4096, 1024, 1, 0, 2048);
    test_potrf_la_case<T>(5000, 1024, 1, 1024, 1000);
    // aggregated trailing updates (round 5): every depth m, ragged orders, with / without the fused diagonal-block segment, a change of
    // policy half-way (knob 48: several depths in one step), explicit inverses narrower than the outer blocks
    for (int m : {1, 2, 3, 4, 7}) {
        gpk_tune(47, m);
        test_potrf_la_case<T>(3000, 256, 1, 0, 300);
        test_potrf_la_case<T>(4100, 512, 1, 1024, 600);
        test_potrf_la_case<T>(2900, 256, 0, 0, 300, 128 * 2);
        gpk_tune(48, 1800);
        test_potrf_la_case<T>(3333, 256, 1, 0, 500);
        gpk_tune(48, 0);
        gpk_tune(40, 256); gpk_tune(41, 4096);
        test_potrf_la_case<T>(3000, 256, 1, 0, 300);
        gpk_tune(48, 1500);
        test_potrf_la_case<T>(3000, 256, 1, 0, 300);
        gpk_tune(48, 0);
        gpk_tune(40, 9216); gpk_tune(41, 512);
    }
    gpk_tune(47, 2);
    test_potrf_la_case<T>(16896, 256, 1, 0, 1024);       // 66 outer blocks: more than a column-group mask has bits (every block, every step)
    // a non-positive-definite matrix is reported with the global pivot order
    {
        const int n = 1500, nb = 512, bad = 1111;
        const int64_t ld = n;
        auto A = make_spd<T>(n, 1, ld);
        A[(size_t)bad * ld + bad] = (T)(-5);
        Dev<T> dA(A.size()), dinv((size_t)gpk_dinv_elems(n)), dbig((size_t)3 * nb * nb), ws((size_t)gpk_potrf_la_ws_elems(n, nb));
        Dev<int> info(1);
        dA.up(A); info.zero();
        for (int64_t tail : {0, 700}) {
            dA.up(A); info.zero();
            gpk_tune(6, 0); gpk_tune(9, tail);
            const int st = gpk_potrf_la(DT<T>::v, dA.p, n, ld, dinv.p, dbig.p, nb, ws.p, info.p, nullptr);
            gpk_tune(6, 2048); gpk_tune(9, 0);
            HIPCHK(hipDeviceSynchronize());
            char nm[160];
            snprintf(nm, sizeof nm, "potrf_la_%s not PD (tail %lld): info %d (expect %d) st%d", DT<T>::name(), (long long)tail, info.down()[0], bad + 1, st);
            report(nm, (st == 0 && info.down()[0] == bad + 1) ? 0.0 : INFINITY, 0.0);
        }
    }
}

// where do workgroups land?  (XCC id, SE/SH/CU fields of HW_ID) of a 2048-workgroup grid
__global__ void census_kernel(unsigned* out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
    // stay a little so that the grid spreads over every CU
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
}
static void census() {
    const int nb = 2048;
    Dev<unsigned> out(2 * nb);
    hipLaunchKernelGGL(census_kernel, dim3(nb), dim3(256), 0, 0, out.p);
    HIPCHK(hipDeviceSynchronize());
    auto h = out.down();
    std::vector<int> cnt(8 * 256, 0);
    int b2x = 0;
    for (int b = 0; b < nb; ++b) {
        const unsigned xcc = h[2 * b] & 0xf, key = (h[2 * b + 1] >> 8) & 0xff;
        if ((int)(xcc & 7) == b % 8) ++b2x;
        cnt[(xcc & 7) * 256 + key]++;
    }
    printf("CENSUS block b on XCC b%%8 for %d of %d blocks; raw xcc/hw_id of blocks 0..7:", b2x, nb);
    for (int b = 0; b < 8; ++b) printf(" %x/%08x", h[2 * b], h[2 * b + 1]);
    printf("\n");
    for (int x = 0; x < 8; ++x) {
        int cus = 0;
        printf("CENSUS xcc %d keys(se.sh.cu:count):", x);
        for (int k = 0; k < 256; ++k)
            if (cnt[x * 256 + k]) { ++cus; printf(" %d.%d.%d:%d", (k >> 5) & 7, (k >> 4) & 1, k & 15, cnt[x * 256 + k]); }
        printf("  => %d distinct CUs\n", cus);
    }
}

// ----------------------------------------------------------------------------
// reductions / misc
// ----------------------------------------------------------------------------
template <typename T>
static void test_misc() {
    const int rows = 1000, cols = 300, batch = 2;
    const int64_t ld = cols + 4;
    auto V = randv<T>((size_t)batch * rows * ld), w = randv<T>((size_t)batch * rows);
    Dev<T> dV(V.size()), dw(w.size()), od((size_t)batch * cols), os((size_t)batch * cols), ws(2 * (size_t)batch * gpk_colreduce_chunks(rows) * cols);
    dV.up(V); dw.up(w);
    int st = gpk_colreduce(DT<T>::v, dV.p, rows, cols, ld, (int64_t)rows * ld, dw.p, rows, od.p, os.p, ws.p, batch, nullptr);
    HIPCHK(hipDeviceSynchronize());
    std::vector<double> rd((size_t)batch * cols, 0), rs((size_t)batch * cols, 0);
    for (int b = 0; b < batch; ++b) for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) {
        const double v = V[((size_t)b * rows + r) * ld + c];
        rd[b * cols + c] += v * w[b * rows + r];
        rs[b * cols + c] += v * v;
    }
    report(std::string("colreduce_dot_") + DT<T>::name(), st ? INFINITY : relerr(od.down(), rd), DT<T>::eps * 100);
    report(std::string("colreduce_ss_") + DT<T>::name(), st ? INFINITY : relerr(os.down(), rs), DT<T>::eps * 100);
    // tril + add_diag + copy2d
    const int n = 300;
    auto A = randv<T>((size_t)n * n), dv = randv<T>(n);
    Dev<T> dA(A.size()), dd(n), dC(A.size());
    dA.up(A); dd.up(dv);
    gpk_add_diag(DT<T>::v, dA.p, n, n, (int64_t)n * n, 0.5, dd.p, n, 1, nullptr);
    gpk_tril(DT<T>::v, dA.p, n, n, (int64_t)n * n, 1, nullptr);
    gpk_copy2d(DT<T>::v, dA.p, n, 0, dC.p, n, 0, n, n, 1, nullptr);
    HIPCHK(hipDeviceSynchronize());
    std::vector<double> ref((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) ref[i * n + j] = j > i ? 0.0 : (double)A[i * n + j] + (i == j ? 0.5 + (double)dv[i] : 0.0);
    report(std::string("add_diag_tril_copy_") + DT<T>::name(), relerr(dC.down(), ref), DT<T>::eps);
}

// ----------------------------------------------------------------------------
// kernel VJP with an explicit cotangent (gradient of the pseudo-point ELBO)
// ----------------------------------------------------------------------------
static void kappa_all_host(int kind, double q, double& k, double& dkq, double& dk) {
    switch (kind) {
        case GPK_K_EQ: k = std::exp(-0.5 * q); dk = -0.5 * k; dkq = dk * q; break;
        case GPK_K_MATERN12: { double r = std::sqrt(q); k = std::exp(-r); dkq = -0.5 * r * k; dk = r > 0 ? -0.5 * k / r : 0.0; break; }
        case GPK_K_MATERN32: { double s = std::sqrt(3 * q), e = std::exp(-s); k = (1 + s) * e; dk = -1.5 * e; dkq = dk * q; break; }
        case GPK_K_MATERN52: { double s = std::sqrt(5 * q), e = std::exp(-s); k = (1 + s + s * s / 3) * e; dk = -(5.0 / 6.0) * (1 + s) * e; dkq = dk * q; break; }
        case GPK_K_LINEAR: k = q; dkq = q; dk = 1; break;
        default: k = 1; dkq = 0; dk = 0;
    }
}
template <typename T>
static void test_vjp_dense_case(std::vector<int> kinds, int n, int m, int d, bool scale_rank1, bool want_gx) {
    const int nt = (int)kinds.size();
    std::vector<double> var(nt), il(nt);
    for (int t = 0; t < nt; ++t) { var[t] = 0.6 + 0.25 * t; il[t] = 1.0 / (0.8 + 0.3 * t); }
    const int64_t ldg = m + 5;
    auto X = randv<T>((size_t)n * d), Y = randv<T>((size_t)m * d), G = randv<T>((size_t)n * ldg);
    auto cs = randv<T>(m), w = randv<T>(n), b = randv<T>(m);
    int64_t rt = 0, nc = 0;
    gpk_kmat_vjp_dense_grid(n, m, &rt, &nc);
    const int W = 2 * GPK_MAX_TERMS + 1;
    Dev<T> dX(X.size()), dY(Y.size()), dG(G.size()), dcs(m), dw(n), db(m), dP((size_t)rt * nc * W), dC((size_t)rt * m), dGX((size_t)nc * n * d);
    dX.up(X); dY.up(Y); dG.up(G); dcs.up(cs); dw.up(w); db.up(b);
    int st = gpk_kmat_vjp_dense(DT<T>::v, kinds.data(), var.data(), il.data(), nt, dX.p, n, d, dY.p, m, d, d, dG.p, ldg,
                                scale_rank1 ? dcs.p : nullptr, scale_rank1 ? dw.p : nullptr, scale_rank1 ? db.p : nullptr,
                                dP.p, dC.p, want_gx ? dGX.p : nullptr, nullptr);
    HIPCHK(hipDeviceSynchronize());
    auto P = dP.down(), C = dC.down(), GX = dGX.down();
    std::vector<double> rS(2 * nt, 0.0), rC(m, 0.0), rGX((size_t)n * d, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double ge = (double)G[(size_t)i * ldg + j];
            if (scale_rank1) ge = ge * (double)cs[j] + (double)w[i] * (double)b[j];
            double r2 = 0, dot = 0;
            for (int k = 0; k < d; ++k) { const double a = X[(size_t)i * d + k], c = Y[(size_t)j * d + k]; r2 += (a - c) * (a - c); dot += a * c; }
            double kfull = 0;
            for (int t = 0; t < nt; ++t) {
                const double q = (kinds[t] == GPK_K_LINEAR ? dot : r2) * il[t] * il[t];
                double k, dkq, dk;
                kappa_all_host(kinds[t], q, k, dkq, dk);
                rS[2 * t] += ge * k; rS[2 * t + 1] += ge * dkq; kfull += var[t] * k;
                for (int c = 0; c < d; ++c) {
                    const double a = X[(size_t)i * d + c], y = Y[(size_t)j * d + c];
                    rGX[(size_t)i * d + c] += kinds[t] == GPK_K_LINEAR ? ge * var[t] * il[t] * il[t] * y
                                                                       : ge * var[t] * dk * 2 * il[t] * il[t] * (a - y);
                }
            }
            rC[j] += ge * kfull;
        }
    std::vector<T> gS(2 * nt, T(0)), gC(m, T(0)), gGX((size_t)n * d, T(0));
    for (int64_t wg = 0; wg < rt * nc; ++wg) for (int t = 0; t < 2 * nt; ++t) gS[t] += P[(size_t)wg * W + t];
    for (int64_t r = 0; r < rt; ++r) for (int j = 0; j < m; ++j) gC[j] += C[(size_t)r * m + j];
    if (want_gx) for (int64_t c = 0; c < nc; ++c) for (size_t e = 0; e < (size_t)n * d; ++e) gGX[e] += GX[(size_t)c * n * d + e];
    char nm[160];
    snprintf(nm, sizeof nm, "vjp_dense_%s k%d nt%d n%d m%d d%d sr%d st%d", DT<T>::name(), kinds[0], nt, n, m, d, scale_rank1, st);
    report(std::string(nm) + " sums", st ? INFINITY : relerr(gS, rS), DT<T>::eps * 50);
    report(std::string(nm) + " colsum", st ? INFINITY : relerr(gC, rC), DT<T>::eps * 50);
    if (want_gx) report(std::string(nm) + " gradx", st ? INFINITY : relerr(gGX, rGX), DT<T>::eps * 50);
}
template <typename T>
static void test_vjp_dense() {
    for (int k = 0; k <= 5; ++k) test_vjp_dense_case<T>({k}, 70, 45, 3, false, true);
    test_vjp_dense_case<T>({GPK_K_EQ, GPK_K_LINEAR}, 130, 1000, 8, true, true);
    test_vjp_dense_case<T>({GPK_K_MATERN52, GPK_K_MATERN32, GPK_K_CONST}, 64, 513, 20, true, false);
    test_vjp_dense_case<T>({GPK_K_EQ}, 300, 9000, 2, true, true);
    {   // d/dX beyond 8 input dimensions is refused, not silently wrong
        Dev<T> z(64);
        std::vector<int> kinds = {GPK_K_EQ};
        std::vector<double> one = {1.0};
        int st = gpk_kmat_vjp_dense(DT<T>::v, kinds.data(), one.data(), one.data(), 1, z.p, 2, 9, z.p, 2, 9, 9, z.p, 2, nullptr, nullptr, nullptr, z.p, nullptr, z.p, nullptr);
        report(std::string("vjp_dense_refuses_gradx_d9_") + DT<T>::name(), st < 0 ? 0.0 : INFINITY, 1.0);
    }
}

// ----------------------------------------------------------------------------
// perf
// ----------------------------------------------------------------------------
struct Timer {
    hipEvent_t a, b;
    Timer() { hipEventCreate(&a); hipEventCreate(&b); }
    void start() { hipEventRecord(a, 0); }
    float stop() { hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms; }
};
template <typename T>
static void perf() {
    Timer tm;
    const double peak = sizeof(T) == 8 ? 78.6 : 157.3;
    // GEMM NT
    for (int n : {4096, 8192}) {
        Dev<T> A((size_t)n * n), B((size_t)n * n), C((size_t)n * n);
        auto h = randv<T>((size_t)n * n);
        A.up(h); B.up(h); C.zero();
        for (int rep = 0; rep < 2; ++rep) {
            tm.start();
            gpk_gemm(DT<T>::v, 1, 1, n, n, n, -1.0, A.p, n, 0, B.p, n, 0, 1.0, C.p, n, 0, 1, 0, nullptr);
            const float ms = tm.stop();
            const double tf = 2.0 * n * n * (double)n / ms * 1e-9;
            printf("PERF gemm_nt_%s n=%d  %.3f ms  %.2f TFLOP/s  (%.1f%% of %.1f)\n", DT<T>::name(), n, ms, tf, 100 * tf / peak, peak);
        }
    }
    // SYRK-like trailing update: M=N=16384-ish, K=256, lower
    for (int k : {128, 256, 512}) {
        const int n = 12288;
        Dev<T> P((size_t)n * k), C((size_t)n * n);
        auto h = randv<T>((size_t)n * k);
        P.up(h); C.zero();
        for (int rep = 0; rep < 2; ++rep) {
            tm.start();
            gpk_gemm(DT<T>::v, 1, 1, n, n, k, -1.0, P.p, k, 0, P.p, k, 0, 1.0, C.p, n, 0, 1, 1, nullptr);
            const float ms = tm.stop();
            const double tf = 1.0 * n * (double)n * k / ms * 1e-9;
            printf("PERF syrk_lower_%s n=%d k=%d  %.3f ms  %.2f TFLOP/s (sym count) (%.1f%%)\n", DT<T>::name(), n, k, ms, tf, 100 * tf / peak);
        }
    }
    // kmat + potrf + trsv + trsm at growing N
    for (int n : {2048, 4096, 8192, 16384}) {
        const int d = 8;
        auto hx = randv<T>((size_t)n * d);
        Dev<T> X(hx.size()), K((size_t)n * n), dinv(gpk_dinv_elems(n));
        Dev<int> info(1);
        X.up(hx);
        int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
        float ms_k = 0, ms_p = 0;
        for (int rep = 0; rep < 2; ++rep) {
            info.zero();
            tm.start();
            gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, d, 0, X.p, n, d, 0, d, K.p, n, 0, 1, 1, 1, 0.1, nullptr, 0, 0, nullptr);
            ms_k = tm.stop();
            tm.start();
            gpk_potrf(DT<T>::v, K.p, n, n, 0, 1, dinv.p, info.p, 0, nullptr);
            ms_p = tm.stop();
        }
        const double tf = (double)n * n * n / 3.0 / ms_p * 1e-9;
        printf("PERF kmat_lower_%s n=%d d=%d  %.3f ms  %.1f GB/s (lower bytes)\n", DT<T>::name(), n, d, ms_k, 0.5 * n * (double)n * sizeof(T) / ms_k * 1e-6);
        printf("PERF potrf_%s n=%d  %.3f ms  %.2f TFLOP/s (%.1f%% of %.1f) info=%d\n", DT<T>::name(), n, ms_p, tf, 100 * tf / peak, peak, info.down()[0]);
        for (int nbo : {128, 512}) {
            info.zero();
            gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, d, 0, X.p, n, d, 0, d, K.p, n, 0, 1, 1, 1, 0.1, nullptr, 0, 0, nullptr);
            tm.start();
            gpk_potrf(DT<T>::v, K.p, n, n, 0, 1, dinv.p, info.p, nbo, nullptr);
            const float ms = tm.stop();
            printf("PERF potrf_%s n=%d nbo=%d  %.3f ms  %.2f TFLOP/s\n", DT<T>::name(), n, nbo, ms, (double)n * n * n / 3.0 / ms * 1e-9);
        }
        // trsv (nrhs = 1), sb = 128
        {
            Dev<T> y(n), tmp(512 + GPK_TRSV_CTRL_ELEMS);
            auto hy = randv<T>(n);
            for (int rep = 0; rep < 2; ++rep) {
                y.up(hy);
                tm.start();
                gpk_trsv_lower(DT<T>::v, K.p, n, n, 0, dinv.p, 128, y.p, 1, 1, 0, tmp.p, 1, nullptr);
                const float ms = tm.stop();
                if (rep) printf("PERF trsv_%s n=%d sb=128  %.3f ms  %.1f GB/s\n", DT<T>::name(), n, ms, 0.5 * n * (double)n * sizeof(T) / ms * 1e-6);
            }
        }
        // merge + trsm (nrhs = 2048)
        for (int sb : {128, 512}) {
            const int nrhs = 2048;
            Dev<T> dsb((size_t)((n + sb - 1) / sb) * sb * sb), tmpm((size_t)((n + sb - 1) / sb) * sb * sb / 4 + 16), Bm((size_t)n * nrhs), tmp((size_t)sb * nrhs);
            auto hb = randv<T>((size_t)n * nrhs);
            Bm.up(hb);
            tm.start();
            gpk_trtri_merge(DT<T>::v, K.p, n, n, 0, 1, dinv.p, sb, dsb.p, tmpm.p, nullptr);
            const float ms_m = tm.stop();
            tm.start();
            gpk_trsm_lower(DT<T>::v, K.p, n, n, 0, dsb.p, sb, Bm.p, nrhs, nrhs, 0, tmp.p, 1, nullptr);
            const float ms = tm.stop();
            printf("PERF trsm_%s n=%d nrhs=%d sb=%d  merge %.3f ms  solve %.3f ms  %.2f TFLOP/s\n", DT<T>::name(), n, nrhs, sb, ms_m, ms, (double)n * n * nrhs / ms * 1e-9);
        }
    }
    // batched: 512 x 2048 (f32 config 4) -- only for float to bound memory/time
    if (sizeof(T) == 4) {
        const int n = 2048, d = 3, batch = 512;
        auto hx = randv<T>((size_t)batch * n * d);
        Dev<T> X(hx.size()), K((size_t)batch * n * n), dinv((size_t)batch * gpk_dinv_elems(n));
        Dev<int> info(batch);
        X.up(hx);
        int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
        for (int rep = 0; rep < 2; ++rep) {
            info.zero();
            tm.start();
            gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, d, (int64_t)n * d, X.p, n, d, (int64_t)n * d, d, K.p, n, (int64_t)n * n, batch, 1, 1, 0.1, nullptr, 0, 0, nullptr);
            const float ms_k = tm.stop();
            tm.start();
            gpk_potrf(DT<T>::v, K.p, n, n, (int64_t)n * n, batch, dinv.p, info.p, 0, nullptr);
            const float ms_p = tm.stop();
            printf("PERF batched_%s 512x2048: kmat %.3f ms, potrf %.3f ms  %.2f TFLOP/s\n", DT<T>::name(), ms_k, ms_p, batch * (double)n * n * n / 3.0 / ms_p * 1e-9);
        }
    }
}


// look-ahead Cholesky and persistent update, timed:  --perf-la
template <typename T>
static void perf_la(int nmax) {
    Timer tm;
    const double peak = sizeof(T) == 8 ? 78.6 : 157.3;
    // trailing-update shape of the first outer step at N = 16384, nbo = 1024
    {
        const int n = std::min(15360, nmax - 1024), k = 1024;
        Dev<T> P((size_t)(n + 1024) * k), C((size_t)n * n), Tn((size_t)n * 1024), Cs((size_t)n * 1024);
        Dev<unsigned> ctrl(64);
        P.up(randv<T>((size_t)(n + 1024) * k, 0.01)); C.zero(); Cs.zero();
        for (int rep = 0; rep < 2; ++rep) {
            tm.start();
            gpk_gemm(DT<T>::v, 1, 1, n, n, k, -1.0, P.p, k, 0, P.p, k, 0, 1.0, C.p, n, 0, 1, 1, nullptr);
            float ms = tm.stop();
            if (rep) printf("PERFLA trail_%s n=%d k=%d plain kernel        %.3f ms  %.2f TFLOP/s (%.1f%%)\n", DT<T>::name(), n, k, ms, 1.0 * n * (double)n * k / ms * 1e-9, 100.0 * n * (double)n * k / ms * 1e-9 / peak);
        }
        for (int reserve = 0; reserve < 2; ++reserve)
            for (int rep = 0; rep < 2; ++rep) {
                gpk_update_t u{n, n, k, P.p, k, P.p, k, C.p, n, C.p, n, 1};
                tm.start();
                gpk_gemm_update2(DT<T>::v, &u, 1, -1.0, ctrl.p, reserve, nullptr);
                float ms = tm.stop();
                if (rep) printf("PERFLA trail_%s n=%d k=%d persistent reserve=%d %.3f ms  %.2f TFLOP/s (%.1f%%)\n", DT<T>::name(), n, k, reserve, ms, 1.0 * n * (double)n * k / ms * 1e-9, 100.0 * n * (double)n * k / ms * 1e-9 / peak);
            }
        for (int rep = 0; rep < 2; ++rep) {   // strip (out of place) + triangle in one launch
            gpk_update_t u[2];
            u[0] = gpk_update_t{n, 1024, k, P.p + (size_t)1024 * k, k, P.p, k, Cs.p, 1024, Tn.p, 1024, 0};
            u[1] = gpk_update_t{n, n, k, P.p + (size_t)1024 * k, k, P.p + (size_t)1024 * k, k, C.p, n, C.p, n, 1};
            tm.start();
            gpk_gemm_update2(DT<T>::v, u, 2, -1.0, ctrl.p, 1, nullptr);
            float ms = tm.stop();
            const double fl = 1.0 * n * (double)n * k + 2.0 * n * 1024.0 * k;
            if (rep) printf("PERFLA trail_%s n=%d k=%d persistent strip+triangle reserve=1 %.3f ms  %.2f TFLOP/s (%.1f%%)\n", DT<T>::name(), n, k, ms, fl / ms * 1e-9, 100 * fl / ms * 1e-9 / peak);
        }
    }
    for (int n : {2048, 4096, 8192, 16384}) {
        if (n > nmax) continue;
        const int d = 8;
        auto hx = randv<T>((size_t)n * d);
        Dev<T> X(hx.size()), K((size_t)n * n), dinv(gpk_dinv_elems(n));
        Dev<int> info(1);
        X.up(hx);
        int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
        for (int rep = 0; rep < 2; ++rep) {
            info.zero();
            gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, d, 0, X.p, n, d, 0, d, K.p, n, 0, 1, 1, 1, 0.1, nullptr, 0, 0, nullptr);
            tm.start();
            gpk_potrf(DT<T>::v, K.p, n, n, 0, 1, dinv.p, info.p, 0, nullptr);
            const float ms = tm.stop();
            if (rep) printf("PERFLA potrf_%s n=%d plain            %.3f ms  %.2f TFLOP/s (%.1f%%)\n", DT<T>::name(), n, ms, (double)n * n * n / 3.0 / ms * 1e-9, 100 * (double)n * n * n / 3.0 / ms * 1e-9 / peak);
        }
        for (int nb : {512, 1024}) {
            Dev<T> dbig((size_t)((n + nb - 1) / nb) * nb * nb), ws((size_t)gpk_potrf_la_ws_elems(n, nb));
            for (int cfg = 0; cfg < 4; ++cfg) {
                const int mode = cfg == 0 ? 0 : 1;
                const int64_t minrows = cfg == 1 ? 0 : (cfg == 2 ? 2048 : 4096);
                gpk_tune(7, mode); gpk_tune(6, minrows);
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    info.zero();
                    gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, d, 0, X.p, n, d, 0, d, K.p, n, 0, 1, 1, 1, 0.1, nullptr, 0, 0, nullptr);
                    tm.start();
                    gpk_potrf_la(DT<T>::v, K.p, n, n, dinv.p, dbig.p, nb, ws.p, info.p, nullptr);
                    const float ms = tm.stop();
                    if (rep) best = std::min(best, ms);
                }
                printf("PERFLA potrf_%s n=%d look-ahead nb=%d mode=%d minrows=%lld  %.3f ms  %.2f TFLOP/s (%.1f%%) info=%d\n", DT<T>::name(), n, nb, mode, (long long)minrows, best,
                       (double)n * n * n / 3.0 / best * 1e-9, 100 * (double)n * n * n / 3.0 / best * 1e-9 / peak, info.down()[0]);
            }
            gpk_tune(7, 1); gpk_tune(6, 2048);
        }
    }
}


// --perf-la-tail: plain (pipelined panels) against look-ahead at several orders, outer blocks and plain-tail lengths
template <typename T>
static void perf_la_tail(std::initializer_list<int> ns) {
    Timer tm;
    for (int n : ns) {
        const int d = 8;
        auto hx = randv<T>((size_t)n * d);
        Dev<T> X(hx.size()), K((size_t)n * n), dinv(gpk_dinv_elems(n));
        Dev<int> info(1);
        X.up(hx);
        int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
        auto run = [&](auto&& call) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                info.zero();
                gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, d, 0, X.p, n, d, 0, d, K.p, n, 0, 1, 1, 1, 0.1, nullptr, 0, 0, nullptr);
                tm.start();
                call();
                const float ms = tm.stop();
                if (rep) best = std::min(best, ms);
            }
            return best;
        };
        for (int nbo : {512, 1024, 2048}) {
            const float ms = run([&] { gpk_potrf(DT<T>::v, K.p, n, n, 0, 1, dinv.p, info.p, nbo, nullptr); });
            printf("PERFTAIL potrf_%s n=%d plain nbo=%d  %.3f ms  %.2f TFLOP/s\n", DT<T>::name(), n, nbo, ms, (double)n * n * n / 3.0 / ms * 1e-9);
        }
        for (int nb : {512, 1024}) {
            Dev<T> dbig((size_t)((n + nb - 1) / nb) * nb * nb), ws((size_t)gpk_potrf_la_ws_elems(n, nb));
            for (int tail : {4096, 6144, 8192, 10240}) {
                if (tail >= n) continue;
                gpk_tune(9, tail);
                const float ms = run([&] { gpk_potrf_la(DT<T>::v, K.p, n, n, dinv.p, dbig.p, nb, ws.p, info.p, nullptr); });
                printf("PERFTAIL potrf_%s n=%d look-ahead nb=%d tail=%d  %.3f ms  %.2f TFLOP/s info=%d\n", DT<T>::name(), n, nb, tail, ms,
                       (double)n * n * n / 3.0 / ms * 1e-9, info.down()[0]);
            }
            gpk_tune(9, 0);
        }
        {   // 1024-column outer blocks with 512-wide explicit inverses (what fp32 runs)
            const int nb = 1024, sb = 512;
            Dev<T> dbig((size_t)((n + sb - 1) / sb) * sb * sb), ws((size_t)gpk_potrf_la_ws_elems(n, nb));
            const float ms = run([&] { gpk_potrf_la_split(DT<T>::v, K.p, n, n, dinv.p, dbig.p, nb, sb, ws.p, info.p, nullptr); });
            printf("PERFTAIL potrf_%s n=%d look-ahead nb=%d sb=%d tail=default  %.3f ms  %.2f TFLOP/s info=%d\n", DT<T>::name(), n, nb, sb, ms,
                   (double)n * n * n / 3.0 / ms * 1e-9, info.down()[0]);
        }
    }
}

// one look-ahead factorisation on an explicit stream, for rocprofv3 traces:  --la-one f64|f32 N NB MODE MINROWS
template <typename T>
static void la_one(int n, int nb, int mode, int64_t minrows, int reps, int ldpad = 0) {
    const int64_t ld = n + ldpad;
    const int d = 8;
    auto hx = randv<T>((size_t)n * d);
    Dev<T> X(hx.size()), K((size_t)n * ld), dinv(gpk_dinv_elems(n));
    Dev<T> dbig((size_t)((n + nb - 1) / nb) * nb * nb), ws((size_t)gpk_potrf_la_ws_elems(n, nb));
    Dev<int> info(1);
    X.up(hx);
    int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
    hipStream_t st;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    gpk_tune(7, mode); gpk_tune(6, minrows);
    for (int rep = 0; rep < reps; ++rep) {
        HIPCHK(hipMemsetAsync(info.p, 0, sizeof(int), st));
        gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, d, 0, X.p, n, d, 0, d, K.p, ld, 0, 1, 1, 1, 0.1, nullptr, 0, 0, st);
        hipEventRecord(a, st);
        if (mode >= 0) gpk_potrf_la(DT<T>::v, K.p, n, ld, dinv.p, dbig.p, nb, ws.p, info.p, st);
        else gpk_potrf(DT<T>::v, K.p, n, ld, 0, 1, dinv.p, info.p, 0, st);
        hipEventRecord(b, st);
        HIPCHK(hipStreamSynchronize(st));
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("LAONE ldpad=%d potrf_%s n=%d nb=%d mode=%d minrows=%lld  %.3f ms  %.2f TFLOP/s info=%d\n", ldpad, DT<T>::name(), n, nb, mode, (long long)minrows, ms, (double)n * n * n / 3.0 / ms * 1e-9, info.down()[0]);
    }
}


// aggregated trailing updates, A/B in one process:  --perf-agg f64|f32 N NB SB ROUNDS TAIL M1 M2 ...
template <typename T>
static void perf_agg(int n, int nb, int sb, int rounds, const std::vector<int>& ms_list, int64_t tail = 0) {
    const int d = 8;
    auto hx = randv<T>((size_t)n * d);
    const int wb = sb > 0 ? sb : nb;
    Dev<T> X(hx.size()), K((size_t)n * n), dinv(gpk_dinv_elems(n));
    Dev<T> dbig((size_t)((n + wb - 1) / wb) * wb * wb), ws((size_t)gpk_potrf_la_ws_elems(n, nb));
    Dev<int> info(1);
    X.up(hx);
    int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
    hipStream_t st;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    if (tail > 0) gpk_tune(9, tail);
    std::vector<double> sum(ms_list.size(), 0.0), best(ms_list.size(), 1e30);
    for (int r = 0; r <= rounds; ++r)          // round 0: warm-up
        for (size_t c = 0; c < ms_list.size(); ++c) {
            gpk_tune(47, ms_list[c]);
            HIPCHK(hipMemsetAsync(info.p, 0, sizeof(int), st));
            gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, d, 0, X.p, n, d, 0, d, K.p, n, 0, 1, 1, 1, 0.1, nullptr, 0, 0, st);
            hipEventRecord(a, st);
            if (sb > 0 && sb < nb) gpk_potrf_la_split(DT<T>::v, K.p, n, n, dinv.p, dbig.p, nb, sb, ws.p, info.p, st);
            else gpk_potrf_la(DT<T>::v, K.p, n, n, dinv.p, dbig.p, nb, ws.p, info.p, st);
            hipEventRecord(b, st);
            HIPCHK(hipStreamSynchronize(st));
            float ms; hipEventElapsedTime(&ms, a, b);
            if (r > 0) { sum[c] += ms; best[c] = std::min(best[c], (double)ms); }
            printf("PERFAGG potrf_%s n=%d nb=%d sb=%d tail=%lld m=%d round %d  %.3f ms  info=%d\n", DT<T>::name(), n, nb, wb, (long long)tail, ms_list[c], r, ms, info.down()[0]);
        }
    for (size_t c = 0; c < ms_list.size(); ++c)
        printf("PERFAGG SUMMARY potrf_%s n=%d nb=%d sb=%d tail=%lld m=%d  mean %.3f ms  best %.3f ms  (%.2f TFLOP/s)\n", DT<T>::name(), n, nb, wb, (long long)tail, ms_list[c],
               sum[c] / rounds, best[c], (double)n * n * n / 3.0 / best[c] * 1e-9);
    gpk_tune(47, 2); gpk_tune(9, 0);
}

// factorisation with rows under the matrix, timed against the factorisation alone + the separate solve:  --perf-rows f64|f32 N EXTRA NB SB ROUNDS
template <typename T>
static void perf_rows(int n, int extra, int nb, int sb, int rounds) {
    const int d = 8;
    const int64_t rows = (int64_t)n + extra;
    auto hx = randv<T>((size_t)n * d);
    const int wb = sb > 0 ? sb : nb;
    Dev<T> X(hx.size()), K((size_t)rows * n), dinv((size_t)gpk_dinv_elems(n) + 128 * 128), E((size_t)extra * n);
    Dev<T> dbig((size_t)((n + wb - 1) / wb) * wb * wb), ws((size_t)gpk_potrf_la_ws_elems(rows, nb));
    Dev<int> info(1);
    X.up(hx);
    E.up(randv<T>((size_t)extra * n, 0.01));
    int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
    hipStream_t st;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int r = 0; r <= rounds; ++r)
        for (int with_rows = 0; with_rows < 2; ++with_rows) {
            HIPCHK(hipMemsetAsync(info.p, 0, sizeof(int), st));
            gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, d, 0, X.p, n, d, 0, d, K.p, n, 0, 1, 1, 1, 0.1, nullptr, 0, 0, st);
            HIPCHK(hipMemcpyAsync(K.p + (size_t)n * n, E.p, sizeof(T) * (size_t)extra * n, hipMemcpyDeviceToDevice, st));
            hipEventRecord(a, st);
            gpk_potrf_rows(DT<T>::v, K.p, n, with_rows ? rows : n, n, dinv.p, dbig.p, nb, (sb > 0 && sb < nb) ? sb : 0, ws.p, info.p, st);
            hipEventRecord(b, st);
            HIPCHK(hipStreamSynchronize(st));
            float ms; hipEventElapsedTime(&ms, a, b);
            if (r > 0) printf("PERFROWS potrf_%s n=%d +%d rows nb=%d sb=%d %s round %d  %.3f ms  info=%d\n", DT<T>::name(), n, extra, nb, wb,
                              with_rows ? "with the rows " : "matrix alone  ", r, ms, info.down()[0]);
        }
}

// shader clock and k-loop pace of chosen trailing updates INSIDE a look-ahead factorisation:  --la-clock f64|f32 N NB
template <typename T>
static void la_clock(int n, int nb, int warm = 0) {
    const int d = 8, grid = 512;
    // `warm` big updates enqueued right before every factorisation (no idle in between): what does the load history do to the clock?
    Dev<T> WP(warm ? (size_t)15360 * 1024 : 1), WC(warm ? (size_t)15360 * 15360 : 1);
    Dev<unsigned> wctrl(64);
    if (warm) { WP.up(randv<T>((size_t)15360 * 1024, 0.01)); WC.zero(); }
    hipEvent_t ea, eb;
    hipEventCreate(&ea); hipEventCreate(&eb);
    auto hx = randv<T>((size_t)n * d);
    Dev<T> X(hx.size()), K((size_t)n * n), dinv(gpk_dinv_elems(n));
    Dev<T> dbig((size_t)((n + nb - 1) / nb) * nb * nb), ws((size_t)gpk_potrf_la_ws_elems(n, nb));
    Dev<int> info(1);
    Dev<long long> prof((size_t)grid * 8 * 8);
    X.up(hx);
    int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
    hipStream_t st;
    HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int kchunks = nb / (int)(128 / sizeof(T));
    for (int which = -2; which < 10; ++which) {      // two warm-up factorisations, then one per stamped update
        HIPCHK(hipMemsetAsync(info.p, 0, sizeof(int), st));
        gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, d, 0, X.p, n, d, 0, d, K.p, n, 0, 1, 1, 1, 0.1, nullptr, 0, 0, st);
        if (which >= 0) { prof.zero(); }
        for (int w = 0; w < warm; ++w) {
            gpk_update_t u{15360, 15360, 1024, WP.p, 1024, WP.p, 1024, WC.p, 15360, WC.p, 15360, 1};
            gpk_gemm_update2(DT<T>::v, &u, 1, -1.0, wctrl.p, 0, st);
        }
        if (which >= 0) { gpk_tune(20, which); gpk_tune_tile_prof(prof.p); }
        hipEventRecord(ea, st);
        gpk_potrf_la(DT<T>::v, K.p, n, n, dinv.p, dbig.p, nb, ws.p, info.p, st);
        hipEventRecord(eb, st);
        HIPCHK(hipStreamSynchronize(st));
        gpk_tune_tile_prof(nullptr);
        float pms = 0; hipEventElapsedTime(&pms, ea, eb);
        printf("LACLOCK potrf %.3f ms (%d updates enqueued right before it)\n", pms, warm);
        if (which < 0) continue;
        auto h = prof.down();
        for (int ti = 1; ti < 8; ti += 3) {
            double b = 0, cyc = 0; int cnt = 0;
            for (int w = 0; w < grid; ++w) {
                const long long* q = &h[((size_t)w * 8 + ti) * 8];
                if (q[3] == 0) continue;
                b += (q[2] - q[1]) * 0.01; cyc += (double)(q[5] - q[4]); ++cnt;
            }
            if (cnt) printf("LACLOCK %s n=%d update #%d, tile #%d of a workgroup (%d workgroups): k loop %.1f us = %.3f us per chunk = %.0f cycles per chunk at %.0f MHz\n",
                            DT<T>::name(), n, which, ti, cnt, b / cnt, b / cnt / kchunks, cyc / cnt / kchunks, cyc / b);
        }
    }
    gpk_tune(20, -1);
}

// kernel-matrix kernel alone, the shapes of the four GPU configs:  --perf-kmat
template <typename T>
static void perf_kmat_case(const char* nm, int nterms, const int* kinds, int64_t n, int64_t m, int d, int batch, int lower, int ldpad = 0) {
    const bool sym = (m == 0);
    if (sym) m = n;
    const int64_t ldk = m + ldpad;
    auto hx = randv<T>((size_t)batch * n * d);
    Dev<T> X(hx.size()), Y(sym ? 1 : (size_t)m * d), K((size_t)batch * n * ldk);
    X.up(hx);
    if (!sym) Y.up(randv<T>((size_t)m * d));
    double var[2] = {1.0, 1.0}, il[2] = {1.0, 1.0};
    Timer tm;
    for (int band = 1; band >= 0; --band) {
        gpk_tune(12, band);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            tm.start();
            gpk_kmat(DT<T>::v, kinds, var, il, nterms, X.p, n, d, n * d, sym ? X.p : Y.p, m, d, sym ? n * d : 0, d, K.p, ldk, n * ldk, batch, lower, sym ? 1 : 0, 0.1, nullptr, 0, 0, nullptr);
            const float ms = tm.stop();
            if (rep) best = std::min(best, ms);
        }
        const double bytes = (lower ? 0.5 : 1.0) * batch * (double)n * m * sizeof(T);
        printf("PERFKMAT %-28s %s %s  %.3f ms  %.2f TB/s (%s bytes)\n", nm, DT<T>::name(), band ? "row-band kernel" : "tile kernel    ", best, bytes / best * 1e-9, lower ? "lower" : "all");
    }
    gpk_tune(12, 1);
}
static void perf_kmat() {
    const int eq[1] = {GPK_K_EQ}, eql[2] = {GPK_K_EQ, GPK_K_LINEAR}, m52[1] = {GPK_K_MATERN52};
    perf_kmat_case<double>("cfg2 N=16384 D=8 EQ lower", 1, eq, 16384, 0, 8, 1, 1);
    perf_kmat_case<double>("cfg2, ld = N + 16", 1, eq, 16384, 0, 8, 1, 1, 16);
    perf_kmat_case<double>("cfg2, ld = N + 80", 1, eq, 16384, 0, 8, 1, 1, 80);
    perf_kmat_case<double>("cfg2 full matrix", 1, eq, 16384, 0, 8, 1, 0);
    perf_kmat_case<double>("N=16000 D=8 EQ lower", 1, eq, 16000, 0, 8, 1, 1);
    perf_kmat_case<float>("N=16384 D=8 EQ lower", 1, eq, 16384, 0, 8, 1, 1);
    perf_kmat_case<float>("cfg3 N=32768 D=4 EQ+Lin lower", 2, eql, 32768, 0, 4, 1, 1);
    perf_kmat_case<float>("cfg4 512xN=2048 D=3 EQ lower", 1, eq, 2048, 0, 3, 512, 1);
    perf_kmat_case<float>("cfg5 4096x200000 D=8 EQ", 1, eq, 4096, 200000, 8, 1, 0);
    const int m32[1] = {GPK_K_MATERN32}, m12[1] = {GPK_K_MATERN12};
    perf_kmat_case<double>("N=16384 D=8 Matern52 lower", 1, m52, 16384, 0, 8, 1, 1);
    perf_kmat_case<double>("N=16384 D=8 Matern32 lower", 1, m32, 16384, 0, 8, 1, 1);
    perf_kmat_case<double>("N=16384 D=8 Matern12 lower", 1, m12, 16384, 0, 8, 1, 1);
    perf_kmat_case<double>("N=16384 D=3 Matern52 lower", 1, m52, 16384, 0, 3, 1, 1);
    perf_kmat_case<float>("N=32768 D=4 Matern52 lower", 1, m52, 32768, 0, 4, 1, 1);
    perf_kmat_case<float>("N=16384x2048 D=8 EQ (K_x*)", 1, eq, 16384, 2048, 8, 1, 0);
}

// one problem, for rocprofv3: kmat + potrf (+ trsv, merge, trsm) at order n
template <typename T>
static void profile_one(int n, int nbo, int reps) {
    const int d = 8, nrhs = 2048, sb = 512;
    auto hx = randv<T>((size_t)n * d);
    Dev<T> X(hx.size()), K((size_t)n * n), dinv(gpk_dinv_elems(n)), y(n), tmp((size_t)sb * nrhs + GPK_TRSV_CTRL_ELEMS);
    Dev<T> dsb((size_t)((n + sb - 1) / sb) * sb * sb), tmpm((size_t)((n + sb - 1) / sb) * sb * sb / 4 + 16), Bm((size_t)n * nrhs);
    Dev<int> info(1);
    X.up(hx);
    y.up(randv<T>(n));
    Bm.up(randv<T>((size_t)n * nrhs));
    int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
    Timer tm;
    for (int rep = 0; rep < reps; ++rep) {
        info.zero();
        gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, d, 0, X.p, n, d, 0, d, K.p, n, 0, 1, 1, 1, 0.1, nullptr, 0, 0, nullptr);
        tm.start();
        gpk_potrf(DT<T>::v, K.p, n, n, 0, 1, dinv.p, info.p, nbo, nullptr);
        const float ms = tm.stop();
        printf("PROFILE potrf_%s n=%d nbo=%d %.3f ms %.2f TFLOP/s\n", DT<T>::name(), n, nbo, ms, (double)n * n * n / 3.0 / ms * 1e-9);
        gpk_trsv_lower(DT<T>::v, K.p, n, n, 0, dinv.p, 128, y.p, 1, 1, 0, tmp.p, 1, nullptr);
        gpk_trtri_merge(DT<T>::v, K.p, n, n, 0, 1, dinv.p, sb, dsb.p, tmpm.p, nullptr);
        gpk_trsm_lower(DT<T>::v, K.p, n, n, 0, dsb.p, sb, Bm.p, nrhs, nrhs, 0, tmp.p, 1, nullptr);
        HIPCHK(hipDeviceSynchronize());
    }
}

// one GEMM shape, timed alone:  --gemm f64|f32 M N K [lower]   (k-major operands, C = C - A B^T)
template <typename T>
static void gemm_one(int M, int N, int K, int flags, int reps, int64_t lda = -1) {
    if (lda < 0) lda = K;
    Dev<T> A((size_t)M * K), B((size_t)N * K), C((size_t)M * N);
    A.up(randv<T>((size_t)M * K, 0.01)); B.up(randv<T>((size_t)N * K, 0.01)); C.zero();
    Timer tm;
    for (int rep = 0; rep < reps; ++rep) {
        tm.start();
        // flags: 1/2/4/8 as in gpk.h, 32: beta = 0, 64: B stored K x N (n contiguous)
        gpk_gemm(DT<T>::v, 1, (flags & 64) ? 0 : 1, M, N, K, -1.0, A.p, lda, 0, B.p, (flags & 64) ? N : K, 0, (flags & 32) ? 0.0 : 1.0, C.p, N, 0, 1, flags & 15, nullptr);
        const float ms = tm.stop();
        const double fl = ((flags & 13) ? 1.0 : 2.0) * M * (double)N * K;
        printf("GEMM %s M=%d N=%d K=%d flags=%d lda=%lld  %.3f ms  %.2f TFLOP/s\n", DT<T>::name(), M, N, K, flags, (long long)lda, ms, fl / ms * 1e-9);
    }
}

__global__ void rsq_probe(const double* x, double* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_rsq(x[i]);
}
static void rsq_precision() {
    const int n = 1 << 16;
    std::vector<double> x(n);
    std::uniform_real_distribution<double> u(0.01, 100.0);
    for (auto& v : x) v = u(rng);
    Dev<double> dx(n), dy(n);
    dx.up(x);
    hipLaunchKernelGGL(rsq_probe, dim3(n / 256), dim3(256), 0, 0, dx.p, dy.p, n);
    HIPCHK(hipDeviceSynchronize());
    auto y = dy.down();
    double worst = 0;
    for (int i = 0; i < n; ++i) worst = std::max(worst, std::fabs(y[i] * std::sqrt(x[i]) - 1.0));
    printf("RSQ_F64 max relative error of v_rsq_f64: %.3e (= 2^%.1f)\n", worst, std::log2(worst));
}


// pure-register MFMA ceiling: every wave runs `iters` x 16 independent v_mfma (no memory)
__global__ __launch_bounds__(256) void mfma_peak64(double* out, int iters, long long* clk) {
    d4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
__global__ __launch_bounds__(256) void mfma_peak32(float* out, int iters, long long* clk) {
    f4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
static void mfma_peak() {
    Timer tm;
    Dev<double> o64((size_t)512 * 256);
    Dev<float> o32((size_t)512 * 256);
    Dev<long long> clk(2);
    int wallrate = 0;
    hipDeviceGetAttribute(&wallrate, hipDeviceAttributeWallClockRate, 0);
    for (int blocks : {256, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 4000;
            tm.start();
            hipLaunchKernelGGL(mfma_peak64, dim3(blocks), dim3(256), 0, 0, o64.p, iters, clk.p);
            float ms = tm.stop();
            auto c = clk.down();
            double tf = (double)blocks * 4 * iters * 16 * 2048.0 / ms * 1e-9;
            if (rep) printf("MFMAPEAK f64 blocks=%d: %.3f ms  %.2f TFLOP/s  cycles=%lld wall_ticks=%lld (wall clock %d kHz) -> shader clock %.0f MHz, %.1f cycles/MFMA/SIMD\n", blocks, ms, tf,
                            c[0], c[1], wallrate, (double)c[0] / ((double)c[1] / wallrate) * 1e-3, (double)c[0] / (iters * 16.0 * (blocks / 256)));
            tm.start();
            hipLaunchKernelGGL(mfma_peak32, dim3(blocks), dim3(256), 0, 0, o32.p, iters, clk.p);
            ms = tm.stop();
            c = clk.down();
            tf = (double)blocks * 4 * iters * 16 * 2048.0 / ms * 1e-9;
            if (rep) printf("MFMAPEAK f32 blocks=%d: %.3f ms  %.2f TFLOP/s  shader clock %.0f MHz\n", blocks, ms, tf, (double)c[0] / ((double)c[1] / wallrate) * 1e-3);
        }
    }
}

template <int BYTES>
__global__ __launch_bounds__(256) void lds_hog(double* out, int iters) {
    __shared__ double buf[BYTES / 8];
    for (int i = threadIdx.x; i < BYTES / 8; i += 256) buf[i] = i;
    __syncthreads();
    double acc = 0;
    for (int it = 0; it < iters; ++it) acc += buf[(threadIdx.x * 17 + it * 31) % (BYTES / 8)];
    out[threadIdx.x] = acc;
}


// ---------------------------------------------------------------------------
// experiment: where and when does the dispatcher place a one-workgroup kernel of a second stream while a
// resident grid holds every CU but a few?   --dispatch
// ---------------------------------------------------------------------------
struct HogArgs { unsigned keys[8]; unsigned* ctrl; long long* t_start; long long spin_ticks; int mode; };   // mode 0: stay, 1: first arrival per XCC leaves, 2: static keys leave
__global__ __launch_bounds__(256, 2) void hog_kernel(HogArgs p) {
    __shared__ char pad[72 * 1024];
    __shared__ int leave;
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        xcc &= 7u;
        const unsigned key = ((hw >> 8) & 0xffu) + 1u;
        int lv = 0;
        if (p.mode == 1) { const unsigned old = atomicCAS(&p.ctrl[1 + xcc], 0u, key); lv = (old == 0u || old == key); }
        if (p.mode == 2) lv = (p.keys[xcc] == key);
        if (lv) atomicAdd(&p.ctrl[9 + xcc], 1u);
        if (blockIdx.x == 0) *p.t_start = wall_clock64();
        leave = lv;
        pad[threadIdx.x] = (char)lv;
    }
    __syncthreads();
    if (leave) return;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < p.spin_ticks) {}
    if (pad[1] == 77) p.ctrl[31] = 1;
}
__global__ __launch_bounds__(256, 1) void probe_kernel(unsigned long long* out, int idx) {
    extern __shared__ char big[];
    if (threadIdx.x == 0) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * (idx * gridDim.x + blockIdx.x)] = ((unsigned long long)(xcc & 7u) << 32) | ((hw >> 8) & 0xffu);
        out[2 * (idx * gridDim.x + blockIdx.x) + 1] = (unsigned long long)wall_clock64();
        big[0] = 1;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}    // 20 us
}
static void masked_census(const char* nm, const uint32_t* mask) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask);
    if (e != hipSuccess) { printf("MASK %s: create failed: %s\n", nm, hipGetErrorString(e)); return; }
    const int nb = 256;
    Dev<unsigned> out(2 * nb);
    hipLaunchKernelGGL(census_kernel, dim3(nb), dim3(256), 0, st, out.p);
    HIPCHK(hipStreamSynchronize(st));
    auto h = out.down();
    std::vector<int> cnt(8 * 256, 0);
    for (int b = 0; b < nb; ++b) cnt[(h[2 * b] & 7) * 256 + ((h[2 * b + 1] >> 8) & 0xff)]++;
    printf("MASK %-28s ->", nm);
    for (int x = 0; x < 8; ++x) for (int k = 0; k < 256; ++k) if (cnt[x * 256 + k]) printf(" x%d:%d.%d(%d)", x, (k >> 5) & 7, k & 15, cnt[x * 256 + k]);
    printf("\n");
    hipStreamDestroy(st);
}
static void dispatch_experiment() {
    HIPCHK(hipFuncSetAttribute((const void*)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 134144));
    // 1. which CUs does a CU-mask bit enable?
    { uint32_t m[8] = {1, 0, 0, 0, 0, 0, 0, 0}; masked_census("bit 0", m); }
    { uint32_t m[8] = {2, 0, 0, 0, 0, 0, 0, 0}; masked_census("bit 1", m); }
    { uint32_t m[8] = {0x100, 0, 0, 0, 0, 0, 0, 0}; masked_census("bit 8", m); }
    { uint32_t m[8] = {0, 1, 0, 0, 0, 0, 0, 0}; masked_census("bit 32", m); }
    { uint32_t m[8] = {0xff, 0, 0, 0, 0, 0, 0, 0}; masked_census("bits 0..7", m); }
    { uint32_t m[8] = {1, 1, 1, 1, 1, 1, 1, 1}; masked_census("bit 0 of every word", m); }
    { uint32_t m[8] = {0x01010101, 0x01010101, 0, 0, 0, 0, 0, 0}; masked_census("bits 0,8,..,56", m); }
    { uint32_t m[8] = {0x11111111, 0, 0, 0, 0, 0, 0, 0}; masked_census("bits 0,4,..,28", m); }
    // 2. probes beside a resident grid
    hipStream_t s_main, s_aux, s_mask;
    HIPCHK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&s_aux, hipStreamNonBlocking));
    uint32_t mk[8] = {0xff, 0, 0, 0, 0, 0, 0, 0};
    const bool have_mask = hipExtStreamCreateWithCUMask(&s_mask, 8, mk) == hipSuccess;
    // keys the masked stream can reach
    unsigned keys[8] = {0};
    if (have_mask) {
        Dev<unsigned> out(2 * 256);
        hipLaunchKernelGGL(census_kernel, dim3(256), dim3(256), 0, s_mask, out.p);
        HIPCHK(hipStreamSynchronize(s_mask));
        auto h = out.down();
        for (int b = 0; b < 256; ++b) keys[h[2 * b] & 7] = ((h[2 * b + 1] >> 8) & 0xff) + 1;
        printf("DISPATCH masked stream keys:");
        for (int x = 0; x < 8; ++x) printf(" x%d:%d.%d", x, ((keys[x] - 1) >> 5) & 7, (keys[x] - 1) & 15);
        printf("\n");
    }
    Dev<unsigned> ctrl(64);
    Dev<long long> tstart(1);
    const int nprobe = 12;
    for (int cfg = 0; cfg < 6; ++cfg) {
        // cfg 0: hog stays everywhere, probes unmasked; 1: first-arrival reserve, probes unmasked; 2: static keys, probes unmasked;
        // 3: static keys, probes on the masked stream; 4: static keys, 64-workgroup small probes (40 KB) unmasked; 5: the same, masked
        const int mode = cfg == 0 ? 0 : (cfg == 1 ? 1 : 2);
        hipStream_t sp = (cfg == 3 || cfg == 5) ? s_mask : s_aux;
        if ((cfg == 3 || cfg == 5) && !have_mask) continue;
        const int pgrid = cfg >= 4 ? 64 : 1;
        const size_t plds = cfg >= 4 ? 40960 : 134144;
        Dev<unsigned long long> out(2 * nprobe * pgrid);
        ctrl.zero();
        HogArgs ha;
        for (int x = 0; x < 8; ++x) ha.keys[x] = keys[x];
        ha.ctrl = ctrl.p; ha.t_start = tstart.p; ha.spin_ticks = 200000; ha.mode = mode;   // 2 ms
        HIPCHK(hipDeviceSynchronize());
        hipLaunchKernelGGL(hog_kernel, dim3(512), dim3(256), 0, s_main, ha);
        for (int i = 0; i < nprobe; ++i) hipLaunchKernelGGL(probe_kernel, dim3(pgrid), dim3(256), plds, sp, out.p, i);
        HIPCHK(hipDeviceSynchronize());
        auto h = out.down();
        auto c = ctrl.down();
        const long long t0 = tstart.down()[0];
        int left = 0; for (int x = 0; x < 8; ++x) left += c[9 + x];
        printf("DISPATCH cfg %d (hog mode %d, %d workgroups left; probes: grid %d, %zu B LDS, %s stream):", cfg, mode, left, pgrid, plds, (sp == s_mask) ? "masked" : "plain");
        for (int i = 0; i < nprobe; ++i) {
            long long tmin = (long long)h[2 * (i * pgrid) + 1], tmax = tmin;
            for (int b = 0; b < pgrid; ++b) { tmin = std::min(tmin, (long long)h[2 * (i * pgrid + b) + 1]); tmax = std::max(tmax, (long long)h[2 * (i * pgrid + b) + 1]); }
            const unsigned long long loc = h[2 * (i * pgrid)];
            printf(" [%lld..%lld us @x%llu:%llu.%llu]", (tmin - t0) / 100, (tmax - t0) / 100, loc >> 32, ((loc & 0xff) >> 5) & 7, loc & 15);
        }
        printf("\n");
    }
}


// where does a tile of the persistent trailing update spend its time?   --tileprof
static void tile_profile(int k = 1024, int lower = 1, int warm = 0) {
    const int n = 15360, grid = 512;
    Dev<double> P((size_t)n * k), C((size_t)n * n);
    Dev<unsigned> ctrl(64);
    Dev<long long> prof((size_t)grid * 8 * 8);
    P.up(randv<double>((size_t)n * k, 0.01)); C.zero();
    Timer tm;
    for (int rep = 0; rep < 2; ++rep) {
        prof.zero();
        gpk_update_t u{n, n, k, P.p, k, P.p, k, C.p, n, C.p, n, lower};
        // `warm` launches of the same update enqueued right before the measured one: the chip is busy (and clocked up) when it starts
        for (int w = 0; rep && w < warm; ++w) gpk_gemm_update2(GPK_F64, &u, 1, -1.0, ctrl.p, 0, nullptr);
        gpk_tune_tile_prof(rep ? prof.p : nullptr);
        tm.start();
        gpk_gemm_update2(GPK_F64, &u, 1, -1.0, ctrl.p, 0, nullptr);
        const float ms = tm.stop();
        printf("TILEPROF update 15360^2 %s k=%d %s, %d launches right before it: %.3f ms  %.1f TFLOP/s\n", lower ? "lower" : "full", k, rep ? "with stamps" : "plain (first launch)",
               rep ? warm : 0, ms, (lower ? 1.0 : 2.0) * n * (double)n * k / ms * 1e-9);
    }
    gpk_tune_tile_prof(nullptr);
    auto h = prof.down();
    for (int ti = 0; ti < 8; ++ti) {
        double a = 0, b = 0, c = 0, gap = 0, cyc = 0, arr = 0; int cnt = 0, cg = 0;
        for (int w = 0; w < grid; ++w) {
            const long long* q = &h[((size_t)w * 8 + ti) * 8];
            if (q[3] == 0) continue;
            a += (q[1] - q[0]) * 0.01; b += (q[2] - q[1]) * 0.01; c += (q[3] - q[2]) * 0.01; cyc += (double)(q[5] - q[4]); ++cnt;
            if (q[6]) arr += (q[6] - q[1]) * 0.01;
            if (ti > 0) { const long long* pq = &h[((size_t)w * 8 + ti - 1) * 8]; gap += (q[0] - pq[3]) * 0.01; ++cg; }
        }
        if (cnt) printf("TILEPROF tile #%d of a workgroup (%d workgroups): request C %.1f us | k loop (incl. C arrival, first chunk) %.1f us = %.3f us per chunk at %.0f MHz shader clock, of it C + first chunk arrived after %.1f us | stores retired %.1f us | gap to previous tile %.1f us\n",
                        ti, cnt, a / cnt, b / cnt, b / cnt / (k / 16), cyc / b, arr / cnt, c / cnt, cg ? gap / cg : 0.0);
    }
    // start-time spread of the first tiles and of the last stamps
    long long t0 = LLONG_MAX, t1 = 0, e0 = LLONG_MAX, e1 = 0;
    for (int w = 0; w < grid; ++w) { const long long* q = &h[(size_t)w * 8 * 8]; if (q[3]) { t0 = std::min(t0, q[0]); t1 = std::max(t1, q[0]); } }
    printf("TILEPROF first tiles start within %.1f us of each other\n", (t1 - t0) * 0.01);
}

// experiment: CU-masked streams (does a reserved CU let a small kernel overlap a big GEMM?)
static void cumask_experiment() {
    const int n = 8192;
    Dev<double> A((size_t)n * n), Bm((size_t)n * n), C((size_t)n * n), D((size_t)2048 * 2048), dinv(gpk_dinv_elems(2048));
    Dev<int> info(1);
    auto h = randv<double>((size_t)n * n);
    A.up(h); Bm.up(h); C.zero();
    auto spd = make_spd<double>(2048, 1, 2048);
    hipStream_t s_def, s_m8, s_half, s_small;
    HIPCHK(hipStreamCreate(&s_def));
    HIPCHK(hipStreamCreate(&s_small));
    uint32_t m8[8], mh[8];
    for (int i = 0; i < 8; ++i) { m8[i] = 0xFFFFFFFFu; mh[i] = 0x0000FFFFu; }
    m8[0] = 0xFFFFFF00u;   // bits 0..7: one CU per XCC if bit i <-> XCC i % 8
    hipError_t e1 = hipExtStreamCreateWithCUMask(&s_m8, 8, m8), e2 = hipExtStreamCreateWithCUMask(&s_half, 8, mh);
    printf("CUMASK create: %s %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto time_gemm = [&](hipStream_t st, const char* nm) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a, st);
            gpk_gemm(GPK_F64, 1, 1, n, n, n, -1.0, A.p, n, 0, Bm.p, n, 0, 1.0, C.p, n, 0, 1, 0, st);
            hipEventRecord(b, st);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("CUMASK gemm 8192^3 on %-10s %.3f ms\n", nm, ms);
        }
    };
    time_gemm(s_def, "default");
    if (e1 == hipSuccess) time_gemm(s_m8, "mask-8");
    if (e2 == hipSuccess) time_gemm(s_half, "mask-half");
    // overlap: big GEMM on X, potrf(2048) chain on s_small; wall time of both
    auto overlap = [&](hipStream_t big, const char* nm) {
        for (int rep = 0; rep < 2; ++rep) {
            D.up(spd); info.zero();
            HIPCHK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            gpk_gemm(GPK_F64, 1, 1, n, n, n, -1.0, A.p, n, 0, Bm.p, n, 0, 1.0, C.p, n, 0, 1, 0, big);
            gpk_potrf(GPK_F64, D.p, 2048, 2048, 0, 1, dinv.p, info.p, 0, s_small);
            hipEventRecord(b, s_small);
            hipEventSynchronize(b);
            auto t1 = std::chrono::high_resolution_clock::now();
            HIPCHK(hipDeviceSynchronize());
            auto t2 = std::chrono::high_resolution_clock::now();
            if (rep) printf("CUMASK overlap big=%-10s potrf(2048) done after %.3f ms, everything after %.3f ms\n", nm,
                            std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t0).count());
        }
    };
    {   // mask-layout sweep: which bits cost the least?
        struct M { const char* nm; std::vector<int> clear; };
        std::vector<M> ms = {{"clr{0}", {0}}, {"clr{0,4..28}", {0, 4, 8, 12, 16, 20, 24, 28}}, {"clr{0..7}", {0, 1, 2, 3, 4, 5, 6, 7}},
                             {"clr{0,32..224}", {0, 32, 64, 96, 128, 160, 192, 224}}, {"clr{0,1,2,3}", {0, 1, 2, 3}},
                             {"clr{0,8..56}", {0, 8, 16, 24, 32, 40, 48, 56}}, {"clr{0..31}", {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31}}};
        for (auto& m : ms) {
            uint32_t w[8];
            for (int i = 0; i < 8; ++i) w[i] = 0xFFFFFFFFu;
            for (int bit : m.clear) w[bit / 32] &= ~(1u << (bit % 32));
            hipStream_t st;
            if (hipExtStreamCreateWithCUMask(&st, 8, w) == hipSuccess) {
                time_gemm(st, m.nm);
                hipStreamDestroy(st);
            }
        }
    }
    overlap(s_def, "default");
    if (e1 == hipSuccess) overlap(s_m8, "mask-8");
    // diag-kernel-only chain (20 x potrf of one 128 block) beside the big GEMM
    auto overlap_diag = [&](hipStream_t big, const char* nm, bool with_big) {
        for (int rep = 0; rep < 2; ++rep) {
            D.up(spd); info.zero();
            HIPCHK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            if (with_big) gpk_gemm(GPK_F64, 1, 1, n, n, n, -1.0, A.p, n, 0, Bm.p, n, 0, 1.0, C.p, n, 0, 1, 0, big);
            for (int q = 0; q < 20; ++q) gpk_potrf(GPK_F64, D.p + (size_t)q * 128 * 2049 * 0, 128, 2048, 0, 1, dinv.p, info.p, 0, s_small);
            hipEventRecord(b, s_small);
            hipEventSynchronize(b);
            auto t1 = std::chrono::high_resolution_clock::now();
            HIPCHK(hipDeviceSynchronize());
            auto t2 = std::chrono::high_resolution_clock::now();
            if (rep) printf("CUMASK diag-chain big=%-10s 20 diag kernels done after %.3f ms, everything after %.3f ms\n", nm,
                            std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t0).count());
        }
    };
    {   // can an 80 KB / 20 KB workgroup on a HIGH-PRIORITY stream get in beside the big GEMM?
        int lo, hi;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipStream_t s_hi;
        HIPCHK(hipStreamCreateWithPriority(&s_hi, hipStreamNonBlocking, hi));
        Dev<double> o(256);
        auto hog = [&](int kb, bool with_big, hipStream_t st, const char* nm) {
            for (int rep = 0; rep < 2; ++rep) {
                HIPCHK(hipDeviceSynchronize());
                auto t0 = std::chrono::high_resolution_clock::now();
                if (with_big) gpk_gemm(GPK_F64, 1, 1, n, n, n, -1.0, A.p, n, 0, Bm.p, n, 0, 1.0, C.p, n, 0, 1, 0, s_def);
                for (int q = 0; q < 20; ++q) {
                    if (kb == 80) hipLaunchKernelGGL(lds_hog<81920>, dim3(1), dim3(256), 0, st, o.p, 20000);
                    else if (kb == 134) hipLaunchKernelGGL(lds_hog<137216>, dim3(1), dim3(256), 0, st, o.p, 20000);
                    else hipLaunchKernelGGL(lds_hog<20480>, dim3(1), dim3(256), 0, st, o.p, 20000);
                }
                hipEventRecord(b, st);
                hipEventSynchronize(b);
                auto t1 = std::chrono::high_resolution_clock::now();
                HIPCHK(hipDeviceSynchronize());
                if (rep) printf("PRIO hog %3d KB x20 on %-8s stream, big GEMM %s: done after %.3f ms\n", kb, nm, with_big ? "yes" : "no ",
                                std::chrono::duration<double, std::milli>(t1 - t0).count());
            }
        };
        printf("PRIO priority range lo=%d hi=%d\n", lo, hi);
        for (int kb : {20, 80, 134}) {
            hog(kb, false, s_hi, "hi-prio");
            hog(kb, true, s_hi, "hi-prio");
            hog(kb, true, s_small, "normal");
        }
    }
    overlap_diag(s_def, "none", false);
    overlap_diag(s_def, "default", true);
    if (e1 == hipSuccess) overlap_diag(s_m8, "mask-8", true);
}

template <typename T>
static void diag_phase_profile(int n, int ver, int pipe = 0) {
    const int nblk = (n + 127) / 128;
    Dev<long long> prof((size_t)nblk * 32);
    prof.zero();
    (void)ver;
    gpk_tune(37, pipe);
    gpk_tune_diag_prof(prof.p);
    profile_one<T>(n, 0, 1);
    gpk_tune_diag_prof(nullptr);
    auto h = prof.down();
    for (int blk : {0, nblk / 2 + 1, nblk - 1}) {
        if (ver == 0) {
            const long long* q = &h[(size_t)blk * 16];
            printf("DIAGPROF %s blk %d cycles: load %lld  factor %lld [trsm %lld  c1 %lld  chol(wave0) %lld]  storeL %lld  invert %lld [16x16 %lld]  storeW %lld  total %lld\n",
                   DT<T>::name(), blk, q[1] - q[0], q[2] - q[1], q[8], q[9], q[10], q[3] - q[2], q[4] - q[3], q[6] - q[3], q[5] - q[4], q[5] - q[0]);
        } else {
            const long long* q = &h[(size_t)blk * 32];
            printf("DIAGPROF3 %s blk %d cycles: load %lld | panel0 %lld | steps", DT<T>::name(), blk, q[1] - q[0], q[2] - q[1]);
            for (int s = 0; s < 7; ++s) printf(" %lld", q[3 + s] - q[2 + s]);
            printf(" | last two row blocks of the inverse %lld | total %lld\n", q[12] - q[10], q[13] - q[0]);
            printf("DIAGPROF3 %s blk %d   of which column update before the panel (U1):", DT<T>::name(), blk);
            for (int s = 0; s < 7; ++s) printf(" %lld", q[21 + s] - q[2 + s]);
            printf("\n");
            if (pipe && q[29] > q[16] && q[30] > q[16])
                printf("PIPEPROF %s blk %d critical path after the inverse is out (us): strip solve starts %.2f, computed %.2f, published %.2f | last tile update starts %.2f, computed %.2f, published %.2f | chain sees the block %.2f\n",
                       DT<T>::name(), blk, (q[17] - q[16]) * 0.01, (q[18] - q[16]) * 0.01, (q[19] - q[16]) * 0.01, (q[20] - q[16]) * 0.01, (q[28] - q[16]) * 0.01,
                       (q[29] - q[16]) * 0.01, (q[30] - q[16]) * 0.01);
            if (pipe && q[15] > q[13] && blk + 1 < nblk)      // the chain workgroup of the pipelined panel
                printf("PIPEPROF %s blk %d cycles: wait for the block's tiles %lld | publish the inverse %lld | step period %lld\n",
                       DT<T>::name(), blk, q[0] - q[14], q[15] - q[13], q[32 + 14] - q[14]);
        }
    }
    gpk_tune(37, 1);
}

// --perf-pipe: the factorisation of small / chain-bound matrices with and without the pipelined panel kernel, at several outer
// block widths (nbo = n: the whole matrix is ONE pipelined panel)
template <typename T>
static void perf_pipe() {
    Timer tm;
    for (int n : {1024, 2048, 4096, 8192}) {
        const int d = 8;
        auto hx = randv<T>((size_t)n * d);
        Dev<T> X(hx.size()), K((size_t)n * n), dinv(gpk_dinv_elems(n));
        Dev<int> info(1);
        X.up(hx);
        int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
        for (int pipe = 0; pipe < 2; ++pipe) {
            gpk_tune(37, pipe);
            for (int nbo : {256, 512, 1024, 2048, 4096, 8192}) {
                if (nbo > n || (!pipe && nbo > 1024)) continue;
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    info.zero();
                    gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, d, 0, X.p, n, d, 0, d, K.p, n, 0, 1, 1, 1, 0.1, nullptr, 0, 0, nullptr);
                    tm.start();
                    gpk_potrf(DT<T>::v, K.p, n, n, 0, 1, dinv.p, info.p, nbo, nullptr);
                    const float ms = tm.stop();
                    if (rep && ms < best) best = ms;
                }
                printf("PERFPIPE potrf_%s n=%d nbo=%d pipe=%d  %.3f ms  %.2f TFLOP/s info=%d\n", DT<T>::name(), n, nbo, pipe, best,
                       (double)n * n * n / 3.0 / best * 1e-9, info.down()[0]);
            }
        }
        gpk_tune(37, 1);
    }
}

// --perf-trsm: the 2048-column solve of the posterior path at N = 16384 (cfg2) / 32768 fp32 (cfg3), in place and out of place
template <typename T>
static void perf_trsm(int n, int nrhs) {
    auto hx = randv<T>((size_t)n * 8);
    Dev<T> X(hx.size()), K((size_t)n * n), dinv(gpk_dinv_elems(n)), Bm((size_t)n * nrhs), Xo((size_t)n * nrhs);
    Dev<int> info(1);
    X.up(hx);
    int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
    info.zero();
    gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, 8, 0, X.p, n, 8, 0, 8, K.p, n, 0, 1, 1, 1, 0.1, nullptr, 0, 0, nullptr);
    gpk_potrf(DT<T>::v, K.p, n, n, 0, 1, dinv.p, info.p, 0, nullptr);
    auto hb = randv<T>((size_t)n * nrhs);
    Timer tm;
    for (int sb : {512, 1024, 2048}) {
        Dev<T> dsb((size_t)((n + sb - 1) / sb) * sb * sb), tmpm((size_t)((n + sb - 1) / sb) * sb * sb / 4 + 16), tmp((size_t)sb * nrhs);
        gpk_trtri_merge(DT<T>::v, K.p, n, n, 0, 1, dinv.p, sb, dsb.p, tmpm.p, nullptr);
        for (int oop = 0; oop < 2; ++oop)
            for (int rep = 0; rep < 3; ++rep) {
                Bm.up(hb);
                tm.start();
                if (oop) gpk_trsm_lower_to(DT<T>::v, K.p, n, n, 0, dsb.p, sb, Bm.p, nrhs, nrhs, 0, Xo.p, nrhs, 0, 1, nullptr);
                else gpk_trsm_lower(DT<T>::v, K.p, n, n, 0, dsb.p, sb, Bm.p, nrhs, nrhs, 0, tmp.p, 1, nullptr);
                const float ms = tm.stop();
                if (rep) printf("PERFTRSM %s n=%d nrhs=%d sb=%d %s  %.3f ms  %.2f TFLOP/s\n", DT<T>::name(), n, nrhs, sb, oop ? "out-of-place" : "in-place    ", ms,
                                (double)n * n * nrhs / ms * 1e-9);
            }
    }
}

// one right-hand side: the per-block sweep (2 n / sb launches) against the single resident launch   --perf-trsv
template <typename T>
static void perf_trsv(int n, std::vector<int> sbs) {
    auto hx = randv<T>((size_t)n * 8);
    Dev<T> X(hx.size()), K((size_t)n * n), dinv(gpk_dinv_elems(n)), y(n);
    Dev<int> info(1);
    X.up(hx);
    int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
    info.zero();
    gpk_kmat(DT<T>::v, &kind, &var, &il, 1, X.p, n, 8, 0, X.p, n, 8, 0, 8, K.p, n, 0, 1, 1, 1, 0.1, nullptr, 0, 0, nullptr);
    gpk_potrf(DT<T>::v, K.p, n, n, 0, 1, dinv.p, info.p, 0, nullptr);
    auto hy = randv<T>((size_t)n);
    Timer tm;
    for (int sb : sbs) {
        Dev<T> dsb((size_t)((n + sb - 1) / sb) * sb * sb), tmpm((size_t)((n + sb - 1) / sb) * sb * sb / 4 + 16), tmp((size_t)sb + GPK_TRSV_CTRL_ELEMS);
        gpk_trtri_merge(DT<T>::v, K.p, n, n, 0, 1, dinv.p, sb, dsb.p, tmpm.p, nullptr);
        std::vector<T> ref;
        for (int sweep = 0; sweep < 2; ++sweep) {
            gpk_tune(49, sweep);
            for (int rep = 0; rep < 4; ++rep) {
                y.up(hy);
                tm.start();
                gpk_trsv_lower(DT<T>::v, K.p, n, n, 0, dsb.p, sb, y.p, 1, 1, 0, tmp.p, 1, nullptr);
                const float ms = tm.stop();
                if (rep) printf("PERFTRSV %s n=%d sb=%d %s  %.3f ms  %.2f TB/s\n", DT<T>::name(), n, sb, sweep ? "one launch  " : "per-block   ", ms,
                                0.5 * n * (double)n * sizeof(T) / ms * 1e-9);
            }
            auto got = y.down();
            if (!sweep) ref = got;
            else {
                double num = 0, den = 0;
                for (int i = 0; i < n; ++i) { num = std::max(num, std::fabs((double)got[i] - (double)ref[i])); den = std::max(den, std::fabs((double)ref[i])); }
                printf("PERFTRSV %s n=%d sb=%d one launch vs per-block: max rel diff %.3e\n", DT<T>::name(), n, sb, num / den);
            }
        }
        gpk_tune(49, 1);
    }
}

// XOR of every 32-bit word of the lower triangles of `batch` n x n fp32 matrices: an order-independent fingerprint of ALL factors of a
// batched factorisation (--batched-stress)
__global__ void xor_lower_kernel(const unsigned* a, int n, long long per, unsigned* out) {
    const long long b = blockIdx.y;
    const int row = blockIdx.x;
    unsigned v = 0;
    for (int c = threadIdx.x; c <= row; c += blockDim.x) v ^= a[b * per + (long long)row * n + c] * (unsigned)(2654435761u + row + 7 * c);
    for (int off = 32; off > 0; off >>= 1) v ^= __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) atomicXor(out, v);
}

int main(int argc, char** argv) {
    bool do_perf = false, only_perf = false;
    for (int i = 1; i + 2 < argc; ++i)     // --set KEY VALUE: tuning knobs (gpk_debug_set)
        if (!strcmp(argv[i], "--set")) gpk_tune(atoi(argv[i + 1]), atoll(argv[i + 2]));
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--batched-stress") && i + 1 < argc) {   // REPS runs of the mixed-phase batched factorisation against ONE lockstep run: every factor, bit for bit
            const int reps = atoi(argv[i + 1]);
            const int n = 2048, d = 3, batch = 512;
            auto hx = randv<float>((size_t)batch * n * d);
            Dev<float> X(hx.size()), K((size_t)batch * n * n), dinv((size_t)batch * gpk_dinv_elems(n));
            Dev<int> info(batch);
            Dev<unsigned> fp(1);
            X.up(hx);
            int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
            auto run = [&](int mode) -> unsigned {
                gpk_tune(53, mode);
                info.zero();
                fp.zero();
                gpk_kmat(GPK_F32, &kind, &var, &il, 1, X.p, n, d, (int64_t)n * d, X.p, n, d, (int64_t)n * d, d, K.p, n, (int64_t)n * n, batch, 1, 1, 0.1, nullptr, 0, 0, nullptr);
                gpk_potrf(GPK_F32, K.p, n, n, (int64_t)n * n, batch, dinv.p, info.p, 0, nullptr);
                hipLaunchKernelGGL(xor_lower_kernel, dim3(n, batch), dim3(256), 0, nullptr, reinterpret_cast<const unsigned*>(K.p), n, (long long)n * n, fp.p);
                unsigned h = 0;
                hipMemcpy(&h, fp.p, sizeof(h), hipMemcpyDeviceToHost);
                std::vector<int> inf(batch);
                hipMemcpy(inf.data(), info.p, sizeof(int) * batch, hipMemcpyDeviceToHost);
                for (int b = 0; b < batch; ++b) if (inf[b] != 0) { printf("info[%d] = %d\n", b, inf[b]); h ^= 0xdeadu; }
                return h;
            };
            const unsigned ref = run(0);
            int bad = 0;
            for (int r = 0; r < reps; ++r) bad += run(1) != ref;
            printf("BATCHED-STRESS %d mixed-phase factorisations of 512 x 2048^2 against the lockstep one (fingerprint %08x of every lower triangle): %d differ\n", reps, ref, bad);
            return bad ? 1 : 0;
        }
        if (!strcmp(argv[i], "--batched") && i + 1 < argc) {   // --batched NBO [BATCH]: BATCH (512) x 2048 f32 potrf with outer block NBO
            const int nbo = atoi(argv[i + 1]);
            const int n = 2048, d = 3, batch = (i + 2 < argc && atoi(argv[i + 2]) >= 8) ? atoi(argv[i + 2]) : 512;
            auto hx = randv<float>((size_t)batch * n * d);
            Dev<float> X(hx.size()), K((size_t)batch * n * n), dinv((size_t)batch * gpk_dinv_elems(n));
            Dev<int> info(batch);
            X.up(hx);
            int kind = GPK_K_EQ; double var = 1.0, il = 1.0;
            Timer tm;
            for (int rep = 0; rep < 3; ++rep) {
                info.zero();
                gpk_kmat(GPK_F32, &kind, &var, &il, 1, X.p, n, d, (int64_t)n * d, X.p, n, d, (int64_t)n * d, d, K.p, n, (int64_t)n * n, batch, 1, 1, 0.1, nullptr, 0, 0, nullptr);
                tm.start();
                gpk_potrf(GPK_F32, K.p, n, n, (int64_t)n * n, batch, dinv.p, info.p, nbo, nullptr);
                const float ms = tm.stop();
                if (rep) printf("BATCHED potrf_f32 %dx2048 nbo=%d  %.3f ms  %.2f TFLOP/s\n", batch, nbo, ms, batch * (double)n * n * n / 3.0 / ms * 1e-9);
            }
            {   // the mixed-phase steps against the lockstep launches (dev build: knob 53): the same arithmetic in the same order per entry --
                // the factors of the first and the last 4 matrices, lower triangles, bit by bit
                std::vector<float> ref[2];
                for (int mode = 0; mode < 2; ++mode) {
                    gpk_tune(53, mode);
                    info.zero();
                    gpk_kmat(GPK_F32, &kind, &var, &il, 1, X.p, n, d, (int64_t)n * d, X.p, n, d, (int64_t)n * d, d, K.p, n, (int64_t)n * n, batch, 1, 1, 0.1, nullptr, 0, 0, nullptr);
                    gpk_potrf(GPK_F32, K.p, n, n, (int64_t)n * n, batch, dinv.p, info.p, nbo, nullptr);
                    hipDeviceSynchronize();
                    ref[mode].resize((size_t)8 * n * n);
                    hipMemcpy(ref[mode].data(), K.p, (size_t)4 * n * n * sizeof(float), hipMemcpyDeviceToHost);
                    hipMemcpy(ref[mode].data() + (size_t)4 * n * n, K.p + (size_t)(batch - 4) * n * n, (size_t)4 * n * n * sizeof(float), hipMemcpyDeviceToHost);
                }
                size_t diff = 0;
                for (int b = 0; b < 8; ++b)
                    for (int i = 0; i < n; ++i)
                        for (int j = 0; j <= i; ++j) {
                            const size_t at = (size_t)b * n * n + (size_t)i * n + j;
                            if (memcmp(&ref[0][at], &ref[1][at], sizeof(float)) != 0) ++diff;
                        }
                printf("BATCHED mixed-phase steps vs lockstep launches: %zu differing entries in 8 factors (release library: both runs are the default path)\n", diff);
                gpk_tune(53, 1);
            }
            // the forward solve of one right-hand side per matrix (the quadratic form of the log-density)
            auto hy = randv<float>((size_t)batch * n);
            Dev<float> Y(hy.size());
            for (int rep = 0; rep < 4; ++rep) {
                Y.up(hy);
                tm.start();
                gpk_trsv_lower(GPK_F32, K.p, n, n, (int64_t)n * n, dinv.p, 128, Y.p, 1, 1, n, nullptr, batch, nullptr);
                const float ms = tm.stop();
                if (rep) printf("BATCHED trsv_f32 512x2048  %.3f ms  %.2f TB/s (lower triangle + inverted diagonal blocks)\n", ms,
                                batch * (0.5 * n * (double)(n - 128) + 128.0 * n) * sizeof(float) / ms * 1e-9);
            }
            return 0;
        }
        if (!strcmp(argv[i], "--cumask")) { cumask_experiment(); return 0; }
        if (!strcmp(argv[i], "--dispatch")) { dispatch_experiment(); return 0; }
        if (!strcmp(argv[i], "--tileprof")) {                  // --tileprof [K [LOWER [WARM]]]
            const int k = (i + 1 < argc && atoi(argv[i + 1]) > 0) ? atoi(argv[i + 1]) : 1024;
            tile_profile(k, (i + 2 < argc) ? atoi(argv[i + 2]) : 1, (i + 3 < argc) ? atoi(argv[i + 3]) : 0);
            return 0;
        }
        if (!strcmp(argv[i], "--mfmapeak")) { mfma_peak(); return 0; }
        if (!strcmp(argv[i], "--diagprof") && i + 1 < argc) {
            rsq_precision();
            for (int ver = 0; ver < 3; ++ver) {      // old kernel, 512-thread kernel, the same inside the pipelined panel
                diag_phase_profile<double>(atoi(argv[i + 1]), ver > 0, ver == 2);
                diag_phase_profile<float>(atoi(argv[i + 1]), ver > 0, ver == 2);
            }
            return 0;
        }
        if (!strcmp(argv[i], "--profile") && i + 3 < argc) {   // --profile f64|f32 N NBO
            const int n = atoi(argv[i + 2]), nbo = atoi(argv[i + 3]);
            if (!strcmp(argv[i + 1], "f64")) profile_one<double>(n, nbo, 2); else profile_one<float>(n, nbo, 2);
            return 0;
        }
        if (!strcmp(argv[i], "--gemm") && i + 4 < argc) {      // --gemm f64|f32 M N K [flags]
            const int M = atoi(argv[i + 2]), N = atoi(argv[i + 3]), K = atoi(argv[i + 4]);
            const int flags = (i + 5 < argc) ? atoi(argv[i + 5]) : 0;
            const int64_t lda = (i + 6 < argc) ? atoll(argv[i + 6]) : -1;
            if (!strcmp(argv[i + 1], "f64")) gemm_one<double>(M, N, K, flags, 4, lda); else gemm_one<float>(M, N, K, flags, 4, lda);
            return 0;
        }
        if (!strcmp(argv[i], "--census")) { census(); return 0; }
        if (!strcmp(argv[i], "--perf-kmat")) { perf_kmat(); return 0; }
        if (!strcmp(argv[i], "--perf-fill") && i + 1 < argc) {      // --perf-fill N: plain POTRF at N for several splits of the workers between panel tasks and fill tiles
            const int n = atoi(argv[i + 1]);
            for (int wg : {-1, 16, 32, 48, 64, 85, 128, 170}) {
                gpk_tune(38, wg < 0 ? 0 : 1);
                gpk_tune(39, wg < 0 ? 0 : wg);
                printf("FILL panel workgroups %d (fill %s):\n", wg, wg < 0 ? "off" : "on");
                perf_la_tail<double>({n});
                perf_la_tail<float>({n});
            }
            gpk_tune(38, 1); gpk_tune(39, 0);
            return 0;
        }
        if (!strcmp(argv[i], "--perf-la-tail")) { perf_la_tail<double>({6144, 8192, 10240, 12288, 16384}); perf_la_tail<float>({8192, 12288, 16384, 32768}); return 0; }
        if (!strcmp(argv[i], "--perf-pipe")) { perf_pipe<double>(); perf_pipe<float>(); return 0; }
        if (!strcmp(argv[i], "--perf-trsv")) { perf_trsv<double>(16384, {512, 1024, 2048}); perf_trsv<float>(32768, {256, 512, 1024}); return 0; }
        if (!strcmp(argv[i], "--perf-trsm")) { perf_trsm<double>(16384, 2048); perf_trsm<float>(32768, 2048); return 0; }
        if (!strcmp(argv[i], "--kmat")) {                      // only the kernel-matrix checks, both kernels
            for (int band = 1; band >= 0; --band) { gpk_tune(12, band); test_kmat<double>(); test_kmat<float>(); }
            printf("SUMMARY pass=%d fail=%d\n", g_pass, g_fail);
            return g_fail ? 1 : 0;
        }
        if (!strcmp(argv[i], "--potrf")) {                     // only the factorisation checks (plain + look-ahead)
            test_potrf<double>(); test_potrf<float>();
            test_lookahead<double>(); test_lookahead<float>();
            printf("SUMMARY pass=%d fail=%d\n", g_pass, g_fail);
            return g_fail ? 1 : 0;
        }
        if (!strcmp(argv[i], "--rows")) {                      // only the factorisations with rows under the matrix
            test_potrf_rows<double>(); test_potrf_rows<float>();
        test_potrf_rhs<double>(); test_potrf_rhs<float>();
            test_potrf_rhs<double>(); test_potrf_rhs<float>();
            printf("SUMMARY pass=%d fail=%d\n", g_pass, g_fail);
            return g_fail ? 1 : 0;
        }
        if (!strcmp(argv[i], "--lookahead")) {                 // only the look-ahead / persistent-update checks
            test_lookahead<double>(); test_lookahead<float>();
            printf("SUMMARY pass=%d fail=%d\n", g_pass, g_fail);
            return g_fail ? 1 : 0;
        }
        if (!strcmp(argv[i], "--la-one") && i + 5 < argc) {      // --la-one f64|f32 N NB MODE(-1 = plain potrf) MINROWS [REPS]
            const int n = atoi(argv[i + 2]), nb = atoi(argv[i + 3]), mode = atoi(argv[i + 4]);
            const int64_t mr = atoll(argv[i + 5]);
            const int reps = (i + 6 < argc) ? atoi(argv[i + 6]) : 3;
            const int ldpad = (i + 7 < argc) ? atoi(argv[i + 7]) : 0;
            if (!strcmp(argv[i + 1], "f64")) la_one<double>(n, nb, mode, mr, reps, ldpad); else la_one<float>(n, nb, mode, mr, reps, ldpad);
            return 0;
        }
        if (!strcmp(argv[i], "--perf-agg") && i + 6 < argc) {    // --perf-agg f64|f32 N NB SB ROUNDS TAIL M1 [M2 ...]
            const int n = atoi(argv[i + 2]), nb = atoi(argv[i + 3]), sb = atoi(argv[i + 4]), rounds = atoi(argv[i + 5]);
            const int64_t tail = atoll(argv[i + 6]);
            std::vector<int> ml;
            for (int q = i + 7; q < argc && argv[q][0] != '-'; ++q) ml.push_back(atoi(argv[q]));
            if (ml.empty()) ml = {1, 2};
            if (!strcmp(argv[i + 1], "f64")) perf_agg<double>(n, nb, sb, rounds, ml, tail); else perf_agg<float>(n, nb, sb, rounds, ml, tail);
            return 0;
        }
        if (!strcmp(argv[i], "--perf-rows") && i + 6 < argc) {    // --perf-rows f64|f32 N EXTRA NB SB ROUNDS
            const int n = atoi(argv[i + 2]), extra = atoi(argv[i + 3]), nb = atoi(argv[i + 4]), sb = atoi(argv[i + 5]), rounds = atoi(argv[i + 6]);
            if (!strcmp(argv[i + 1], "f64")) perf_rows<double>(n, extra, nb, sb, rounds); else perf_rows<float>(n, extra, nb, sb, rounds);
            return 0;
        }
        if (!strcmp(argv[i], "--la-clock") && i + 3 < argc) {
            const int warm = (i + 4 < argc) ? atoi(argv[i + 4]) : 0;
            if (!strcmp(argv[i + 1], "f64")) la_clock<double>(atoi(argv[i + 2]), atoi(argv[i + 3]), warm); else la_clock<float>(atoi(argv[i + 2]), atoi(argv[i + 3]), warm);
            return 0;
        }
        if (!strcmp(argv[i], "--perf-la")) {                   // --perf-la [NMAX]
            const int nmax = (i + 1 < argc && atoi(argv[i + 1]) > 0) ? atoi(argv[i + 1]) : 16384;
            perf_la<double>(nmax); perf_la<float>(nmax);
            return 0;
        }
        if (!strcmp(argv[i], "--perf")) do_perf = true;
        if (!strcmp(argv[i], "--only-perf")) do_perf = only_perf = true;
    }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  arch=%s  CUs=%d  clock=%d MHz  gpk_version=%d\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000, gpk_version());
    if (!only_perf) {
        test_probe<double>(); test_probe<float>();
        gpk_tune(1, 0);                 // force the 128x128-tile kernels
        test_gemm<double>(); test_gemm<float>();
        gpk_tune(1, (int64_t)1 << 40);  // force the 64x64-tile kernels
        test_gemm<double>(); test_gemm<float>();
        gpk_tune(1, 512);               // library default
        test_kmat<double>(); test_kmat<float>();
        test_potrf<double>(); test_potrf<float>();
        test_lookahead<double>(); test_lookahead<float>();
        test_potrf_rows<double>(); test_potrf_rows<float>();
        test_potrf_rhs<double>(); test_potrf_rhs<float>();
        test_misc<double>(); test_misc<float>();
        test_vjp_dense<double>(); test_vjp_dense<float>();
        printf("SUMMARY pass=%d fail=%d\n", g_pass, g_fail);
    }
    if (do_perf) { perf<double>(); perf<float>(); }
    return g_fail ? 1 : 0;
}
