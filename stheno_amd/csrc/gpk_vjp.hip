// gpk_vjp.hip -- kernel-hyperparameter VJP of the GP log-density ("next" row 8(f)-1:
// hyper-parameter learning, reference usage readme_example13_optimisation_torch.py:47-53).
//
// With  G = d logpdf / dK = 1/2 (A diag(g) A^T - s K^{-1})   (A = K^{-1} r, N x C;
// g = upstream gradients per column, s = sum(g)),  the gradient w.r.t. the parameters of
//   K_ij = sum_t v_t kappa_t(q_t),  q_t = |x_i - x_j|^2 / l_t^2   (or <x_i, x_j> / l_t^2)
// is  d/dv_t = sum_ij G_ij kappa_t,   d/dl_t = -2 v_t / l_t * sum_ij G_ij kappa_t'(q) q,
// d/dnoise_i = G_ii.  One pass over the LOWER triangle of K^{-1} (symmetry: off-diagonal
// entries count twice): pairwise distances are recomputed from x, G is never stored.
// HBM-bound: reads N^2/2 elements once.  Output: per-workgroup partial sums
// [nblocks][2 T + 1] (S1_t, S2_t, trace(G)) -- summed by the caller -- and diag(G).
#include "gpk_common.hpp"

enum { VK_EQ = 0, VK_MATERN12 = 1, VK_MATERN32 = 2, VK_MATERN52 = 3, VK_LINEAR = 4, VK_CONST = 5 };

namespace {

constexpr int VT = 64;    // tile edge
constexpr int VDC = 8;    // input dims per staged chunk
constexpr int VMAXC = 8;  // max columns of A

template <typename T>
struct VjpArgs {
    const T* X;
    const T* Kinv;
    const T* A;
    T* partial;
    T* diagG;
    int64_t ldx, ldk, lda;
    int n, d, C, nterms, ntile;
    int kind[GPK_MAX_TERMS];
    T ils2[GPK_MAX_TERMS];
    T g[VMAXC];
    T s;
};

template <typename T>
__device__ __forceinline__ T vexp(T x);
template <>
__device__ __forceinline__ double vexp<double>(double x) { return exp(x); }
template <>
__device__ __forceinline__ float vexp<float>(float x) { return expf(x); }
template <typename T>
__device__ __forceinline__ T vsqrt(T x);
template <>
__device__ __forceinline__ double vsqrt<double>(double x) { return sqrt(x); }
template <>
__device__ __forceinline__ float vsqrt<float>(float x) { return sqrtf(x); }

// kappa(q) and kappa'(q) * q
template <typename T>
__device__ __forceinline__ void kappa_and_dq(int kind, T q, T& k, T& dkq) {
    if (kind == VK_EQ) {
        k = vexp<T>(T(-0.5) * q);
        dkq = T(-0.5) * q * k;
    } else if (kind == VK_MATERN12) {
        const T r = vsqrt<T>(q);
        k = vexp<T>(-r);
        dkq = T(-0.5) * r * k;
    } else if (kind == VK_MATERN32) {
        const T s = vsqrt<T>(T(3) * q), e = vexp<T>(-s);
        k = (T(1) + s) * e;
        dkq = T(-0.5) * s * s * e;
    } else if (kind == VK_MATERN52) {
        const T s = vsqrt<T>(T(5) * q), e = vexp<T>(-s);
        k = (T(1) + s + s * s * T(1.0 / 3.0)) * e;
        dkq = -(s * s * T(1.0 / 6.0)) * (T(1) + s) * e;
    } else if (kind == VK_LINEAR) {
        k = q;
        dkq = q;
    } else {
        k = T(1);
        dkq = T(0);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void kmat_vjp_kernel(VjpArgs<T> p) {
    __shared__ T xi[VT * VDC], xj[VT * VDC], ai[VT * VMAXC], aj[VT * VMAXC];
    __shared__ T red[4 * (2 * GPK_MAX_TERMS + 1)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // lower-triangular tile enumeration
    const int bid = blockIdx.x;
    int ti = (int)((sqrtf(8.f * (float)bid + 1.f) - 1.f) * 0.5f);
    while ((ti + 1) * (ti + 2) / 2 <= bid) ++ti;
    while (ti * (ti + 1) / 2 > bid) --ti;
    const int tj = bid - ti * (ti + 1) / 2;
    const int i0 = ti * VT, j0 = tj * VT;
    const int ty = tid >> 4, tx = tid & 15;     // 16 x 16 threads, 4 x 4 elements each

    T r2[4][4], dt[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) r2[a][b] = dt[a][b] = T(0);

    for (int dc = 0; dc < p.d; dc += VDC) {
        __syncthreads();
        for (int idx = tid; idx < VT * VDC; idx += 256) {
            const int r = idx / VDC, c = idx % VDC;
            xi[idx] = (i0 + r < p.n && dc + c < p.d) ? p.X[(int64_t)(i0 + r) * p.ldx + dc + c] : T(0);
            xj[idx] = (j0 + r < p.n && dc + c < p.d) ? p.X[(int64_t)(j0 + r) * p.ldx + dc + c] : T(0);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < VDC; ++c) {
            T a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                a[q] = xi[(ty * 4 + q) * VDC + c];
                b[q] = xj[(tx * 4 + q) * VDC + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const T df = a[u] - b[v];
                    r2[u][v] += df * df;
                    dt[u][v] += a[u] * b[v];
                }
        }
    }
    // alphas of the tile rows / columns
    for (int idx = tid; idx < VT * VMAXC; idx += 256) {
        const int r = idx / VMAXC, c = idx % VMAXC;
        ai[idx] = (i0 + r < p.n && c < p.C) ? p.A[(int64_t)(i0 + r) * p.lda + c] : T(0);
        aj[idx] = (j0 + r < p.n && c < p.C) ? p.A[(int64_t)(j0 + r) * p.lda + c] : T(0);
    }
    __syncthreads();

    // G for this thread's 4 x 4 elements (weighted for the symmetric double count)
    T Gw[4][4];
    T tr = T(0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + ty * 4 + u;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int j = j0 + tx * 4 + v;
            Gw[u][v] = T(0);
            if (i < p.n && j < p.n && j <= i) {
                T aa = T(0);
                for (int c = 0; c < p.C; ++c) aa += p.g[c] * ai[(ty * 4 + u) * VMAXC + c] * aj[(tx * 4 + v) * VMAXC + c];
                const T G = T(0.5) * (aa - p.s * p.Kinv[(int64_t)i * p.ldk + j]);
                Gw[u][v] = (i == j) ? G : T(2) * G;
                if (i == j) {
                    tr += G;
                    p.diagG[i] = G;
                }
            }
        }
    }
    // per term: two block-wide sums (no per-thread accumulator array -> no scratch)
    T* out = p.partial + (int64_t)bid * (2 * GPK_MAX_TERMS + 1);
    for (int t = 0; t <= p.nterms; ++t) {
        T s1 = T(0), s2 = T(0);
        if (t < p.nterms) {
            const int kind = p.kind[t];
            const T ils2 = p.ils2[t];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const T q = (kind == VK_LINEAR ? dt[u][v] : r2[u][v]) * ils2;
                    T k, dkq;
                    kappa_and_dq<T>(kind, q, k, dkq);
                    s1 += Gw[u][v] * k;
                    s2 += Gw[u][v] * dkq;
                }
        } else {
            s1 = tr;     // last round carries the trace
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1 += __shfl_xor(s1, o, 64);
            s2 += __shfl_xor(s2, o, 64);
        }
        __syncthreads();
        if (lane == 0) {
            red[wave * 2] = s1;
            red[wave * 2 + 1] = s2;
        }
        __syncthreads();
        if (tid == 0) {
            const T a1 = red[0] + red[2] + red[4] + red[6], a2 = red[1] + red[3] + red[5] + red[7];
            if (t < p.nterms) {
                out[2 * t] = a1;
                out[2 * t + 1] = a2;
            } else {
                out[2 * GPK_MAX_TERMS] = a1;
            }
        }
    }
}

}  // namespace

int64_t gpk_kmat_vjp_blocks_impl(int64_t n) {
    const int64_t nt = gpk_cdiv(n > 0 ? n : 1, VT);
    return nt * (nt + 1) / 2;
}

// partial: gpk_kmat_vjp_blocks(n) * (2 * GPK_MAX_TERMS + 1) elements; diag_g: n elements
template <typename T>
int gpk_kmat_vjp_launch(const int* kinds, const double* inv_ls, int nterms, const T* X, int64_t n, int64_t ldx,
                        int d, const T* Kinv, int64_t ldk, const T* A, int C, int64_t lda, const double* g,
                        T* partial, T* diag_g, hipStream_t stream) {
    if (n <= 0) return GPK_OK;
    if (nterms < 0 || nterms > GPK_MAX_TERMS) return GPK_ERR_ARG(3);
    if (C < 1 || C > VMAXC) return GPK_ERR_ARG(11);
    if (n > INT32_MAX) return GPK_ERR_ARG(5);
    VjpArgs<T> a;
    a.X = X; a.Kinv = Kinv; a.A = A; a.partial = partial; a.diagG = diag_g;
    a.ldx = ldx; a.ldk = ldk; a.lda = lda;
    a.n = (int)n; a.d = d; a.C = C; a.nterms = nterms;
    a.ntile = (int)gpk_cdiv(n, VT);
    double s = 0;
    for (int c = 0; c < VMAXC; ++c) {
        a.g[c] = c < C ? (T)g[c] : T(0);
        if (c < C) s += g[c];
    }
    a.s = (T)s;
    for (int t = 0; t < GPK_MAX_TERMS; ++t) {
        a.kind[t] = t < nterms ? kinds[t] : VK_CONST;
        a.ils2[t] = t < nterms ? (T)(inv_ls[t] * inv_ls[t]) : T(0);
    }
    const int64_t nb = gpk_kmat_vjp_blocks_impl(n);
    hipLaunchKernelGGL((kmat_vjp_kernel<T>), dim3((unsigned)nb), dim3(256), 0, stream, a);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

template int gpk_kmat_vjp_launch<double>(const int*, const double*, int, const double*, int64_t, int64_t, int,
                                         const double*, int64_t, const double*, int, int64_t, const double*,
                                         double*, double*, hipStream_t);
template int gpk_kmat_vjp_launch<float>(const int*, const double*, int, const float*, int64_t, int64_t, int,
                                        const float*, int64_t, const float*, int, int64_t, const double*, float*,
                                        float*, hipStream_t);
