// gpk_potrf.hip -- right-looking blocked Cholesky  A = L L^T  (lower, row-major,
// in place), batched, f64/f32, for gfx950.
//
// Replaces: `B.cholesky(B.reg(K))` of the reference's dependency stack, i.e.
// LAPACK dpotrf/spotrf under `B.logdet` / `B.iqf_diag` (stheno/random.py:274-276)
// and under `B.cholesky(K_z)` (stheno/model/observations.py:300).
//
// Structure per outer block of `nbo` columns (nbo = 256 by default):
//   for each 128-column inner block c:
//     1. potrf_diag_kernel: ONE workgroup factorises the 128x128 diagonal block
//        entirely in LDS (16-wide micro-panels: shuffle-based 16x16 Cholesky on
//        one wave, thread-per-row micro-TRSM, MFMA rank-16 update), writes L_cc,
//        then inverts L_cc in place in LDS (16x16 substitutions + recursive
//        doubling with MFMA) and writes inv(L_cc) to the `dinv` workspace.
//     2. panel TRSM as an MFMA GEMM:  A[c+128:, c:c+128] <- A[...] * inv(L_cc)^T
//     3. strip update (rank 128) of the remaining columns of the outer block.
//   4. trailing SYRK update (rank nbo, lower tiles only, XCD-aware tile order):
//        A[k1:, k1:] -= A[k1:, k0:k1] A[k1:, k0:k1]^T        <- the MFMA-bound part
//
// inv(L_cc) blocks are kept: gpk_solve.hip turns every triangular solve of the
// path into GEMM/GEMV work with them.
#include "gpk_common.hpp"

namespace {

constexpr int LDP = 130;   // padded LDS row pitch (elements) of the 128x128 block

template <typename T>
struct DiagArgs {
    T* A;
    int64_t ld, bstride, off;
    int n;
    T* dinv;
    int64_t dinv_bstride;
    int* info;
    long long* prof;   // debug: per-phase cycle stamps of workgroup 0 (nullable)
};

#define PROF_MARK(i)                                                            \
    do {                                                                        \
        if (p.prof != nullptr && blockIdx.x == 0 && threadIdx.x == 0)           \
            p.prof[(p.off / GPK_DB) * 8 + (i)] = (long long)__builtin_readcyclecounter(); \
    } while (0)

// sqrt(d) and 1/sqrt(d) together, Goldschmidt from the hardware rsq estimate:
// ~1 ulp for both, a dozen dependent FMAs instead of the library sqrt + IEEE
// divide (which dominated the serial 16x16 step).  d is a Cholesky pivot of a
// jittered SPD matrix: no range scaling needed; d <= 0 / NaN yields NaN and is
// reported through `info`.
__device__ __forceinline__ void sqrt_rsqrt(double d, double& s, double& r) {
    double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    const double res = fma(-g, g, d);
    g = fma(res, h, g);
    s = g;
    r = h + h;
}
__device__ __forceinline__ void sqrt_rsqrt(float d, float& s, float& r) {
    float y = __builtin_amdgcn_rsqf(d);
    float g = d * y, h = 0.5f * y;
    float e = fmaf(-h, g, 0.5f);
    g = fmaf(g, e, g);
    h = fmaf(h, e, h);
    const float res = fmaf(-g, g, d);
    g = fmaf(res, h, g);
    s = g;
    r = h + h;
}

// value of `v` in lane `src` (compile-time constant after unrolling) as a scalar
__device__ __forceinline__ double lane_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// acc += (neg ? -1 : 1) * A[ar.., ac..] (16 x K, row-major in S) * B[br.., bc..] (K x 16)
template <typename T>
__device__ __forceinline__ typename Traits<T>::acc_t lds_mm_nn(const T* S, int ar, int ac, int br,
                                                               int bc, int K, int lr, int kq,
                                                               bool neg,
                                                               typename Traits<T>::acc_t acc) {
    for (int kk = 0; kk < K / 4; ++kk) {
        T a = S[(ar + lr) * LDP + ac + 4 * kk + kq];
        T b = S[(br + 4 * kk + kq) * LDP + bc + lr];
        if (neg) a = -a;
        acc = Traits<T>::mfma(a, b, acc);
    }
    return acc;
}

template <typename T>
__global__ __launch_bounds__(256, 1) void potrf_diag_kernel(DiagArgs<T> p) {
    typedef typename Traits<T>::acc_t acc_t;
    __shared__ __attribute__((aligned(16))) T S[GPK_DB * LDP + GPK_DB];
    T* rdiag = S + GPK_DB * LDP;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;
    const int64_t b = blockIdx.x;
    T* __restrict__ A = p.A + b * p.bstride + p.off * p.ld + p.off;
    int rem = p.n - (int)p.off;
    const int nv = rem < GPK_DB ? rem : GPK_DB;

    // ---- phase 0: load the lower triangle; pad with identity ----
    // All global loads of a thread are issued before the first LDS store (one
    // latency, not one per element).
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    constexpr int CPR = GPK_DB / VEC;              // 16-byte chunks per row
    constexpr int PER = GPK_DB * CPR / 256;        // chunks per thread
    const bool vec_io = (nv == GPK_DB) && ((uintptr_t)A % 16 == 0) && (p.ld % VEC == 0);
    PROF_MARK(0);
    if (vec_io) {
        vec_t buf[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CPR, c = (id % CPR) * VEC;
            if (c <= r) {
                buf[i] = *reinterpret_cast<const vec_t*>(A + (int64_t)r * p.ld + c);
            } else {
#pragma unroll
                for (int v = 0; v < VEC; ++v) buf[i][v] = T(0);
            }
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CPR, c = (id % CPR) * VEC;
#pragma unroll
            for (int v = 0; v < VEC; ++v) S[r * LDP + c + v] = (c + v <= r) ? buf[i][v] : T(0);
        }
    } else {
        for (int base = 0; base < GPK_DB * GPK_DB; base += 256 * 16) {
            T buf[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int idx = base + tid + 256 * i;
                const int r = idx >> 7, c = idx & 127;
                buf[i] = (r < nv && c <= r) ? A[(int64_t)r * p.ld + c] : ((r == c) ? T(1) : T(0));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int idx = base + tid + 256 * i;
                S[(idx >> 7) * LDP + (idx & 127)] = buf[i];
            }
        }
    }
    __syncthreads();
    PROF_MARK(1);

    // ---- phase 1: factorise, 16-column micro-panels ----
    for (int s = 0; s < 8; ++s) {
        const int c0 = 16 * s;
        // (a) 16x16 diagonal micro-block: lane lr of wave 0 owns row lr.
        if (wave == 0) {
            T a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = S[(c0 + lr) * LDP + c0 + c];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const T d = lane_bcast(a[j], j);
                if (!(d > T(0)) && lane == 0) atomicCAS(p.info + b, 0, (int)p.off + c0 + j + 1);
                T ljj, rinv;
                sqrt_rsqrt(d, ljj, rinv);
                if (lr == j)
                    a[j] = ljj;
                else
                    a[j] *= rinv;
                if (lane == j) rdiag[c0 + j] = rinv;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) {
                    const T lcj = lane_bcast(a[j], c);
                    a[c] -= a[j] * lcj;
                }
            }
            if (lane < 16) {
#pragma unroll
                for (int c = 0; c < 16; ++c) S[(c0 + lr) * LDP + c0 + c] = a[c];
            }
        }
        __syncthreads();
        // (b) micro-panel TRSM: one thread per row below the micro-block.
        {
            const int row = c0 + 16 + tid;
            if (row < GPK_DB) {
                T x[16];
#pragma unroll
                for (int c = 0; c < 16; ++c) x[c] = S[row * LDP + c0 + c];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    T v = x[j];
#pragma unroll
                    for (int c = 0; c < j; ++c) v -= x[c] * S[(c0 + j) * LDP + c0 + c];
                    x[j] = v * rdiag[c0 + j];
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) S[row * LDP + c0 + c] = x[c];
            }
        }
        __syncthreads();
        // (c) rank-16 update of the remaining lower 16x16 tiles on MFMA.
        {
            const int m = 7 - s;                 // tiles bi, bj in (s, 7]
            const int ntile = m * (m + 1) / 2;
            for (int t = wave; t < ntile; t += 4) {
                int i = 0;
                while ((i + 1) * (i + 2) / 2 <= t) ++i;
                const int j = t - i * (i + 1) / 2;
                const int bi = s + 1 + i, bj = s + 1 + j;
                acc_t acc;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[q] = S[(16 * bi + Traits<T>::crow(lane, q)) * LDP + 16 * bj + lr];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const T av = -S[(16 * bi + lr) * LDP + c0 + 4 * kk + kq];
                    const T bv = S[(16 * bj + lr) * LDP + c0 + 4 * kk + kq];
                    acc = Traits<T>::mfma(av, bv, acc);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    S[(16 * bi + Traits<T>::crow(lane, q)) * LDP + 16 * bj + lr] = acc[q];
            }
        }
        __syncthreads();
    }

    PROF_MARK(2);
    // ---- phase 2: write L (lower triangle only; the upper triangle is never touched) ----
    if (vec_io) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int id = tid + 256 * i;
            const int r = id / CPR, c = (id % CPR) * VEC;
            if (c <= r) {
                if (c + VEC - 1 <= r) {
                    vec_t w;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) w[v] = S[r * LDP + c + v];
                    *reinterpret_cast<vec_t*>(A + (int64_t)r * p.ld + c) = w;
                } else {
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
                        if (c + v <= r) A[(int64_t)r * p.ld + c + v] = S[r * LDP + c + v];
                }
            }
        }
    } else {
        for (int idx = tid; idx < GPK_DB * GPK_DB; idx += 256) {
            const int r = idx >> 7, c = idx & 127;
            if (r < nv && c <= r) A[(int64_t)r * p.ld + c] = S[r * LDP + c];
        }
    }

    if (p.dinv == nullptr) return;
    __syncthreads();   // phase 2 reads of S complete before the in-place inversion
    PROF_MARK(3);

    // ---- phase 3: invert L in place ----
    // I. the eight 16x16 diagonal micro-blocks: 16-lane group g of waves 0/1 owns
    //    micro-block 4*wave + g; lane lr solves for column lr of the inverse.
    if (wave < 2) {
        const int c0 = 16 * (4 * wave + kq);
        T x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            T v = (i == lr) ? T(1) : T(0);
#pragma unroll
            for (int k = 0; k < i; ++k) v -= S[(c0 + i) * LDP + c0 + k] * x[k];
            x[i] = v * rdiag[c0 + i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) S[(c0 + i) * LDP + c0 + lr] = x[i];   // zeros above the diagonal
    }
    __syncthreads();
    // II. recursive doubling: [A 0; C D]^-1 = [Ai 0; -Di C Ai, Di].
    for (int h = 16; h < GPK_DB; h *= 2) {
        const int tpp = (h / 16) * (h / 16);          // 16x16 tiles per pair
        const int npair = GPK_DB / (2 * h);
        const int nitem = npair * tpp;                 // 4, 8, 16
        const int per = nitem / 4;                     // items per wave: 1, 2, 4
        acc_t acc[4];
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < per) {
                    const int item = wave * per + q;
                    const int pr = item / tpp, t = item % tpp;
                    const int ti = t / (h / 16), tj = t % (h / 16);
                    const int o = pr * 2 * h;
                    acc_t z;
                    z[0] = z[1] = z[2] = z[3] = T(0);
                    if (pass == 0)   // T = C * Ainv
                        acc[q] = lds_mm_nn<T>(S, o + h + 16 * ti, o, o, o + 16 * tj, h, lr, kq, false, z);
                    else             // C' = -Dinv * T
                        acc[q] = lds_mm_nn<T>(S, o + h + 16 * ti, o + h, o + h, o + 16 * tj, h, lr, kq, true, z);
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < per) {
                    const int item = wave * per + q;
                    const int pr = item / tpp, t = item % tpp;
                    const int ti = t / (h / 16), tj = t % (h / 16);
                    const int o = pr * 2 * h;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        S[(o + h + 16 * ti + Traits<T>::crow(lane, i)) * LDP + o + 16 * tj + lr] = acc[q][i];
                }
            }
            __syncthreads();
        }
    }

    PROF_MARK(4);
    // ---- phase 4: write inv(L) (identity-padded, zeros above the diagonal) ----
    T* __restrict__ W = p.dinv + b * p.dinv_bstride + (p.off / GPK_DB) * (int64_t)(GPK_DB * GPK_DB);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int id = tid + 256 * i;
        const int r = id / CPR, c = (id % CPR) * VEC;
        vec_t w;
#pragma unroll
        for (int v = 0; v < VEC; ++v) w[v] = S[r * LDP + c + v];
        *reinterpret_cast<vec_t*>(W + (int64_t)r * GPK_DB + c) = w;
    }
    PROF_MARK(5);
}

long long* g_diag_prof = nullptr;   // development aid, set through gpk_debug_diag_prof

template <typename T>
struct PanelCtx {
    T* A;
    int64_t n, ld, batch, bstride;
    T* dinv;
    int64_t dstride;
    int* info;
    hipStream_t stream;
};

// Factor the panel columns [c0, c0 + w) (rows c0..n), all updates from columns
// < c0 already applied.  Recursive halving: every off-diagonal flop inside the
// panel is done by a GEMM whose K is the largest power-of-two block available
// (128, 256, ... w/2), not by rank-128 updates.
template <typename T>
int potrf_panel(const PanelCtx<T>& x, int64_t c0, int64_t w) {
    if (c0 >= x.n) return GPK_OK;
    if (w <= GPK_DB) {
        DiagArgs<T> d;
        d.A = x.A; d.ld = x.ld; d.bstride = x.bstride; d.off = c0; d.n = (int)x.n;
        d.dinv = x.dinv; d.dinv_bstride = x.dstride; d.info = x.info;
        d.prof = g_diag_prof;
        hipLaunchKernelGGL((potrf_diag_kernel<T>), dim3((unsigned)x.batch), dim3(256), 0, x.stream, d);
        GPK_CHECK_LAUNCH();
        const int64_t r1 = c0 + GPK_DB;   // first row below the diagonal block
        if (r1 >= x.n) return GPK_OK;
        // panel TRSM as a GEMM: A[r1:, c0:c0+128] <- A[r1:, c0:c0+128] * inv(L_cc)^T
        T* P = x.A + r1 * x.ld + c0;
        const T* Wc = x.dinv + (c0 / GPK_DB) * (int64_t)(GPK_DB * GPK_DB);
        return gpk_gemm_launch<T>(true, true, x.n - r1, GPK_DB, GPK_DB, T(1), P, x.ld, x.bstride, Wc, GPK_DB,
                                  x.dstride, T(0), P, x.ld, x.bstride, x.batch, false, x.stream);
    }
    const int64_t h = w / 2;
    int st = potrf_panel<T>(x, c0, h);
    if (st) return st;
    const int64_t cm = c0 + h;
    if (cm >= x.n) return GPK_OK;
    const int64_t ce = (c0 + w < x.n) ? c0 + w : x.n;
    // A[cm:, cm:ce] -= A[cm:, c0:cm] A[cm:ce, c0:cm]^T   (tiles above the diagonal skipped)
    const T* P = x.A + cm * x.ld + c0;
    st = gpk_gemm_launch<T>(true, true, x.n - cm, ce - cm, h, T(-1), P, x.ld, x.bstride, P, x.ld, x.bstride,
                            T(1), x.A + cm * x.ld + cm, x.ld, x.bstride, x.batch, true, x.stream);
    if (st) return st;
    return potrf_panel<T>(x, cm, h);
}

}  // namespace

// Development aid (not part of include/gpk.h): device buffer of 8 int64 per diagonal
// block that receives cycle-counter stamps of the diag kernel's phases.
extern "C" void gpk_debug_diag_prof(long long* dev_buf) { g_diag_prof = dev_buf; }

template <typename T>
int gpk_potrf_launch(T* A, int64_t n, int64_t ld, int64_t batch, int64_t bstride, T* dinv,
                     int* info, int nbo, hipStream_t stream) {
    if (n <= 0 || batch <= 0) return GPK_OK;
    if (n > INT32_MAX) return GPK_ERR_ARG(2);
    if (ld < n) return GPK_ERR_ARG(3);
    if (nbo <= 0) nbo = (n >= 8192) ? 1024 : (n >= 4096 ? 512 : 256);
    if (nbo < GPK_DB || (nbo & (nbo - 1))) return GPK_ERR_ARG(9);   // 128 * 2^k
    if (info == nullptr) return GPK_ERR_ARG(7);
    if (dinv == nullptr && n > GPK_DB) return GPK_ERR_ARG(6);
    const int64_t nblk = gpk_cdiv(n, GPK_DB);
    const int64_t dstride = nblk * GPK_DB * GPK_DB;

    PanelCtx<T> ctx{A, n, ld, batch, bstride, dinv, dstride, info, stream};
    for (int64_t k0 = 0; k0 < n; k0 += nbo) {
        const int64_t k1 = (k0 + nbo < n) ? k0 + nbo : n;
        int pst = potrf_panel<T>(ctx, k0, nbo);
        if (pst) return pst;
        if (k1 < n) {
            const T* P = A + k1 * ld + k0;
            int st = gpk_gemm_launch<T>(true, true, n - k1, n - k1, k1 - k0, T(-1), P, ld, bstride, P, ld,
                                        bstride, T(1), A + k1 * ld + k1, ld, bstride, batch, true, stream);
            if (st) return st;
        }
    }
    return GPK_OK;
}

template int gpk_potrf_launch<double>(double*, int64_t, int64_t, int64_t, int64_t, double*, int*,
                                      int, hipStream_t);
template int gpk_potrf_launch<float>(float*, int64_t, int64_t, int64_t, int64_t, float*, int*, int,
                                     hipStream_t);
