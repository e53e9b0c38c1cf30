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

// ---------------------------------------------------------------------------------------------
// Dense-G variant (gradient of the pseudo-point ELBO, row 8(f)-1): the cotangent of a RECTANGULAR
// kernel matrix K(X, Y) (n x m) is given explicitly,
//     Geff_ij = G_ij * colscale_j + w_i * b_j        (colscale / (w, b) optional),
// and one pass produces
//   * per-term sums   S1_t = sum Geff kappa_t,  S2_t = sum Geff kappa_t'(q) q      (-> d/dv_t, d/dl_t),
//   * column sums     colsum_j = sum_i Geff_ij K_ij                                (-> d/d noise_j),
//   * d/dX            gradx_i = sum_j Geff_ij dK_ij/dx_i   (first argument; input dim <= 8).
// A workgroup owns a 64-row tile and a CHUNK of 64-column tiles: the d/dX accumulators stay in
// registers across the chunk, column sums leave per row tile; partial results are summed by the
// caller (deterministic, no atomics).  HBM-bound: G is read exactly once.
constexpr int DT = 64;

template <typename T>
struct VjpDenseArgs {
    const T *X, *Y, *G, *cs, *w, *b;
    T *partial, *colsum, *gradx;
    int64_t ldx, ldy, ldg;
    int n, m, d, nterms, tiles_per_chunk, ctiles, nchunks;
    int kind[GPK_MAX_TERMS];
    T ils2[GPK_MAX_TERMS], var[GPK_MAX_TERMS];
};

// kappa(q), kappa'(q) q and kappa'(q)   (kappa' of exp(-sqrt(q)) is singular at 0: reported as 0)
template <typename T>
__device__ __forceinline__ void kappa_all(int kind, T q, T& k, T& dkq, T& dk) {
    if (kind == VK_EQ) {
        k = vexp<T>(T(-0.5) * q);
        dk = T(-0.5) * k;
        dkq = dk * q;
    } else if (kind == VK_MATERN12) {
        const T r = vsqrt<T>(q);
        k = vexp<T>(-r);
        dkq = T(-0.5) * r * k;
        dk = r > T(0) ? T(-0.5) * k / r : T(0);
    } else if (kind == VK_MATERN32) {
        const T s = vsqrt<T>(T(3) * q), e = vexp<T>(-s);
        k = (T(1) + s) * e;
        dk = T(-1.5) * e;
        dkq = dk * q;
    } else if (kind == VK_MATERN52) {
        const T s = vsqrt<T>(T(5) * q), e = vexp<T>(-s);
        k = (T(1) + s + s * s * T(1.0 / 3.0)) * e;
        dk = T(-5.0 / 6.0) * (T(1) + s) * e;
        dkq = dk * q;
    } else if (kind == VK_LINEAR) {
        k = q;
        dkq = q;
        dk = T(1);
    } else {
        k = T(1);
        dkq = T(0);
        dk = T(0);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void kmat_vjp_dense_kernel(VjpDenseArgs<T> p) {
    __shared__ T xi[DT * VDC], yj[DT * VDC];
    __shared__ T csred[16 * DT];
    __shared__ T red[4 * 2 * GPK_MAX_TERMS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ty = tid >> 4, tx = tid & 15;     // 16 x 16 threads, 4 x 4 elements each
    const int rt = blockIdx.x, chunk = blockIdx.y;
    const int i0 = rt * DT;
    const bool want_gx = p.gradx != nullptr;     // launcher guarantees d <= VDC then

    T s1[GPK_MAX_TERMS], s2[GPK_MAX_TERMS];
#pragma unroll
    for (int t = 0; t < GPK_MAX_TERMS; ++t) s1[t] = s2[t] = T(0);
    T gx[4][VDC];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < VDC; ++c) gx[u][c] = T(0);
    T wi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + ty * 4 + u;
        wi[u] = (p.w != nullptr && i < p.n) ? p.w[i] : T(0);
    }

    const int ct_end = min((chunk + 1) * p.tiles_per_chunk, p.ctiles);
    for (int ct = chunk * p.tiles_per_chunk; ct < ct_end; ++ct) {
        const int j0 = ct * DT;
        T r2[4][4], dt[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) r2[a][b] = dt[a][b] = T(0);
        for (int dc = 0; dc < p.d; dc += VDC) {
            __syncthreads();
            for (int idx = tid; idx < DT * VDC; idx += 256) {
                const int r = idx / VDC, c = idx % VDC;
                xi[idx] = (i0 + r < p.n && dc + c < p.d) ? p.X[(int64_t)(i0 + r) * p.ldx + dc + c] : T(0);
                yj[idx] = (j0 + r < p.m && dc + c < p.d) ? p.Y[(int64_t)(j0 + r) * p.ldy + dc + c] : T(0);
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < VDC; ++c) {
                T a[4], b[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[q] = xi[(ty * 4 + q) * VDC + c];
                    b[q] = yj[(tx * 4 + q) * VDC + c];
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
        // effective cotangent of this thread's 4 x 4 elements
        T Ge[4][4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int j = j0 + tx * 4 + v;
            const T csj = (p.cs != nullptr && j < p.m) ? p.cs[j] : T(1);
            const T bj = (p.b != nullptr && j < p.m) ? p.b[j] : T(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + ty * 4 + u;
                T g = T(0);
                if (i < p.n && j < p.m) g = p.G[(int64_t)i * p.ldg + j] * csj + wi[u] * bj;
                Ge[u][v] = g;
            }
        }
        T kfull[4][4], cS[4][4], cL[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int v = 0; v < 4; ++v) kfull[u][v] = cS[u][v] = cL[u][v] = T(0);
#pragma unroll
        for (int t = 0; t < GPK_MAX_TERMS; ++t) {
            if (t < p.nterms) {
                const int kind = p.kind[t];
                const T ils2 = p.ils2[t], var = p.var[t];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const T q = (kind == VK_LINEAR ? dt[u][v] : r2[u][v]) * ils2;
                        T k, dkq, dk;
                        kappa_all<T>(kind, q, k, dkq, dk);
                        s1[t] += Ge[u][v] * k;
                        s2[t] += Ge[u][v] * dkq;
                        kfull[u][v] += var * k;
                        if (kind == VK_LINEAR)
                            cL[u][v] += var * ils2;
                        else
                            cS[u][v] += T(2) * var * ils2 * dk;
                    }
            }
        }
        // column sums over this row tile: 4 rows per thread, then the 16 ty-groups through LDS
        if (p.colsum != nullptr) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                T acc = T(0);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += Ge[u][v] * kfull[u][v];
                csred[ty * DT + tx * 4 + v] = acc;
            }
            __syncthreads();
            if (tid < DT) {
                T acc = T(0);
#pragma unroll
                for (int g = 0; g < 16; ++g) acc += csred[g * DT + tid];
                if (j0 + tid < p.m) p.colsum[(int64_t)rt * p.m + j0 + tid] = acc;
            }
        }
        // d/dx_i: sum_j Ge [ cS (x_i - y_j) + cL y_j ]     (xi / yj still hold the only d-chunk)
        if (want_gx) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                T sumS = T(0);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    cS[u][v] *= Ge[u][v];
                    cL[u][v] = cL[u][v] * Ge[u][v] - cS[u][v];
                    sumS += cS[u][v];
                }
#pragma unroll
                for (int c = 0; c < VDC; ++c) {
                    T acc = sumS * xi[(ty * 4 + u) * VDC + c];
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc += cL[u][v] * yj[(tx * 4 + v) * VDC + c];
                    gx[u][c] += acc;
                }
            }
        }
    }

    // per-term sums of the whole workgroup
    T* out = p.partial + ((int64_t)rt * p.nchunks + chunk) * (2 * GPK_MAX_TERMS + 1);
#pragma unroll
    for (int t = 0; t < GPK_MAX_TERMS; ++t) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s1[t] += __shfl_xor(s1[t], o, 64);
            s2[t] += __shfl_xor(s2[t], o, 64);
        }
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int t = 0; t < GPK_MAX_TERMS; ++t) {
            red[(wave * GPK_MAX_TERMS + t) * 2] = s1[t];
            red[(wave * GPK_MAX_TERMS + t) * 2 + 1] = s2[t];
        }
    }
    __syncthreads();
    if (tid < 2 * GPK_MAX_TERMS) {
        out[tid] = red[tid] + red[2 * GPK_MAX_TERMS + tid] + red[4 * GPK_MAX_TERMS + tid] + red[6 * GPK_MAX_TERMS + tid];
    } else if (tid == 2 * GPK_MAX_TERMS) {
        out[tid] = T(0);
    }
    // d/dX: reduce over the 16 tx lanes of each row group, one partial per column chunk
    if (want_gx) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < VDC; ++c) {
                T v = gx[u][c];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                const int i = i0 + ty * 4 + u;
                if (tx == 0 && i < p.n && c < p.d) p.gradx[((int64_t)chunk * p.n + i) * p.d + c] = v;
            }
    }
}

}  // namespace

void gpk_kmat_vjp_dense_grid_impl(int64_t n, int64_t m, int64_t* rowtiles, int64_t* nchunks, int64_t* tiles_per_chunk) {
    const int64_t rt = gpk_cdiv(n > 0 ? n : 1, DT), ct = gpk_cdiv(m > 0 ? m : 1, DT);
    int64_t nc = gpk_cdiv(2048, rt);
    if (nc > ct) nc = ct;
    if (nc < 1) nc = 1;
    const int64_t tpc = gpk_cdiv(ct, nc);
    *rowtiles = rt;
    *nchunks = gpk_cdiv(ct, tpc);
    if (tiles_per_chunk) *tiles_per_chunk = tpc;
}

template <typename T>
int gpk_kmat_vjp_dense_launch(const int* kinds, const double* variances, const double* inv_ls, int nterms,
                              const T* X, int64_t n, int64_t ldx, const T* Y, int64_t m, int64_t ldy, int d,
                              const T* G, int64_t ldg, const T* colscale, const T* w, const T* b, T* partial,
                              T* colsum, T* gradx, hipStream_t stream) {
    if (n <= 0 || m <= 0) return GPK_OK;
    if (nterms < 0 || nterms > GPK_MAX_TERMS) return GPK_ERR_ARG(5);
    if (n > INT32_MAX || m > INT32_MAX) return GPK_ERR_ARG(7);
    if ((w == nullptr) != (b == nullptr)) return GPK_ERR_ARG(17);
    if (gradx != nullptr && d > VDC) return GPK_ERR_ARG(12);     // d/dX is implemented for d <= 8
    VjpDenseArgs<T> a;
    a.X = X; a.Y = Y; a.G = G; a.cs = colscale; a.w = w; a.b = b;
    a.partial = partial; a.colsum = colsum; a.gradx = gradx;
    a.ldx = ldx; a.ldy = ldy; a.ldg = ldg;
    a.n = (int)n; a.m = (int)m; a.d = d; a.nterms = nterms;
    int64_t rt, nc, tpc;
    gpk_kmat_vjp_dense_grid_impl(n, m, &rt, &nc, &tpc);
    a.tiles_per_chunk = (int)tpc; a.nchunks = (int)nc; a.ctiles = (int)gpk_cdiv(m, DT);
    for (int t = 0; t < GPK_MAX_TERMS; ++t) {
        a.kind[t] = t < nterms ? kinds[t] : VK_CONST;
        a.ils2[t] = t < nterms ? (T)(inv_ls[t] * inv_ls[t]) : T(0);
        a.var[t] = t < nterms ? (T)variances[t] : T(0);
    }
    if (nc > 65535) return GPK_ERR_ARG(9);
    hipLaunchKernelGGL((kmat_vjp_dense_kernel<T>), dim3((unsigned)rt, (unsigned)nc), dim3(256), 0, stream, a);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

template int gpk_kmat_vjp_dense_launch<double>(const int*, const double*, const double*, int, const double*, int64_t,
                                               int64_t, const double*, int64_t, int64_t, int, const double*, int64_t,
                                               const double*, const double*, const double*, double*, double*,
                                               double*, hipStream_t);
template int gpk_kmat_vjp_dense_launch<float>(const int*, const double*, const double*, int, const float*, int64_t,
                                              int64_t, const float*, int64_t, int64_t, int, const float*, int64_t,
                                              const float*, const float*, const float*, float*, float*, float*,
                                              hipStream_t);

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
