// gpk_reduce.hip -- the HBM-bound odds and ends of the GP path: log-determinant
// from the Cholesky diagonal, fused column reductions over V = L^{-1} K(x, x*),
// triangle clean-up, diagonal updates, strided copies.
//
// Replaces (reference call sites): `B.logdet` stheno/random.py:274,
// observations.py:334; `B.iqf_diag` / `B.matmul_diag` column sums-of-squares
// random.py:276, observations.py:305; the `(L^{-1}K)^T (L^{-1}y)` contraction
// inside mlkernels.PosteriorMean (observations.py:161-168).
#include "gpk_common.hpp"

namespace {

template <typename T>
__device__ __forceinline__ T gpk_log(T x);
template <>
__device__ __forceinline__ double gpk_log<double>(double x) { return log(x); }
template <>
__device__ __forceinline__ float gpk_log<float>(float x) { return logf(x); }

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    T s = T(0);
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    return s;
}

// out[b] = 2 * sum_i log L[i][i]
// NT threads per matrix.  The diagonal is one element per row of the factor: every load is a cache line of its own, so what counts is how
// many are in flight -- eight per thread, and 1024 threads when the launch is a single large matrix (N = 16384: 48 -> ~10 us; with
// 256 threads and one load at a time the kernel was 64 dependent round trips to HBM).
template <typename T, int NT>
__global__ __launch_bounds__(NT) void logdet_kernel(const T* __restrict__ L, int64_t n, int64_t ld,
                                                    int64_t sL, T* __restrict__ out) {
    __shared__ T red[NT / 64];
    const int64_t b = blockIdx.x;
    const T* Lb = L + b * sL;
    T acc = T(0);
    int64_t i = threadIdx.x;
    for (; i + 7 * NT < n; i += 8 * NT) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = Lb[(i + u * NT) * (ld + 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += gpk_log<T>(v[u]);
    }
    for (; i < n; i += NT) acc += gpk_log<T>(Lb[i * (ld + 1)]);
    const T s = block_sum<T>(acc, red);
    if (threadIdx.x == 0) out[b] = T(2) * s;
}

// Column reductions of V (rows x cols): partial sums over row chunks.
//   pdot[chunk][col] = sum_{r in chunk} V[r][col] * w[r]   (if w)
//   pss [chunk][col] = sum_{r in chunk} V[r][col]^2
template <typename T>
__global__ __launch_bounds__(256) void colreduce_partial_kernel(const T* __restrict__ V, int64_t rows,
                                                                int64_t cols, int64_t ld, int64_t sV,
                                                                const T* __restrict__ w, int64_t sw,
                                                                T* __restrict__ pdot, T* __restrict__ pss,
                                                                int rch, int nchunk) {
    const int64_t b = blockIdx.z;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y;
    if (col >= cols) return;
    const T* Vb = V + b * sV;
    const int64_t r0 = (int64_t)chunk * rch;
    const int64_t r1 = (r0 + rch < rows) ? r0 + rch : rows;
    T d = T(0), s = T(0);
    if (w != nullptr) {
        const T* wb = w + b * sw;
        for (int64_t r = r0; r < r1; ++r) {
            const T v = Vb[r * ld + col];
            d += v * wb[r];
            s += v * v;
        }
        pdot[(b * nchunk + chunk) * cols + col] = d;
    } else {
        for (int64_t r = r0; r < r1; ++r) {
            const T v = Vb[r * ld + col];
            s += v * v;
        }
    }
    if (pss != nullptr) pss[(b * nchunk + chunk) * cols + col] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void colreduce_final_kernel(const T* __restrict__ pdot,
                                                              const T* __restrict__ pss, int64_t cols,
                                                              int nchunk, T* __restrict__ odot,
                                                              T* __restrict__ oss) {
    const int64_t b = blockIdx.y;
    const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    if (odot != nullptr) {
        T d = T(0);
        for (int c = 0; c < nchunk; ++c) d += pdot[(b * nchunk + c) * cols + col];
        odot[b * cols + col] = d;
    }
    if (oss != nullptr) {
        T s = T(0);
        for (int c = 0; c < nchunk; ++c) s += pss[(b * nchunk + c) * cols + col];
        oss[b * cols + col] = s;
    }
}

// zero the strict upper triangle
template <typename T>
__global__ __launch_bounds__(256) void tril_kernel(T* __restrict__ A, int64_t n, int64_t ld, int64_t sA) {
    const int64_t b = blockIdx.z;
    const int64_t r0 = (int64_t)blockIdx.y * 16, c0 = (int64_t)blockIdx.x * 256;
    if (c0 + 255 <= r0) return;   // tile entirely on/below the diagonal
    const int64_t c = c0 + threadIdx.x;
    if (c >= n) return;
    T* Ab = A + b * sA;
    for (int i = 0; i < 16; ++i) {
        const int64_t r = r0 + i;
        if (r < n && c > r) Ab[r * ld + c] = T(0);
    }
}

// out[i][j] = sum_s parts[s][i][j] over the tiles on / below the diagonal (the partial products of a split-K symmetric update:
// only their lower tiles were ever written); fixed summation order s = 0, 1, ...; 16-byte loads when everything is aligned.
template <typename T>
__global__ __launch_bounds__(256) void sum_lower_kernel(const T* __restrict__ parts, int S, int64_t n, int64_t ldp, int64_t sP,
                                                        T* __restrict__ out, int64_t ldo, int vec_ok) {
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    const int64_t r0 = (int64_t)blockIdx.y * 16, c0 = (int64_t)blockIdx.x * (256 * VEC);
    if (c0 > r0 + 15) return;                 // the block of 16 rows x 256 VEC columns lies above the diagonal
    const int64_t c = c0 + (int64_t)threadIdx.x * VEC;
    for (int i = 0; i < 16; ++i) {
        const int64_t r = r0 + i;
        if (r >= n || c > r) continue;        // (whole vectors up to the one holding the diagonal: a few entries above it are written too -- with the sums of what the products wrote there or left there, never read as part of a lower triangle)
        if ((vec_ok & 1) && c + VEC <= n) {
            vec_t acc = *reinterpret_cast<const vec_t*>(parts + r * ldp + c);
            for (int s = 1; s < S; ++s) {
                const vec_t v = *reinterpret_cast<const vec_t*>(parts + (int64_t)s * sP + r * ldp + c);
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] += v[e];
            }
            // only the entries on / below the diagonal (vec_ok bit 1: `out` takes 16-byte stores)
            if ((vec_ok & 2) && c + VEC - 1 <= r) {
                *reinterpret_cast<vec_t*>(out + r * ldo + c) = acc;
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (c + e <= r) out[r * ldo + c + e] = acc[e];
            }
        } else {
            for (int e = 0; e < VEC; ++e) {
                if (c + e >= n || c + e > r) continue;
                T acc = parts[r * ldp + c + e];
                for (int s = 1; s < S; ++s) acc += parts[(int64_t)s * sP + r * ldp + c + e];
                out[r * ldo + c + e] = acc;
            }
        }
    }
}

// A[i][i] += s + (v ? v[i] : 0)
template <typename T>
__global__ void add_diag_kernel(T* __restrict__ A, int64_t n, int64_t ld, int64_t sA, T s,
                                const T* __restrict__ v, int64_t sv) {
    const int64_t b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    T add = s;
    if (v != nullptr) add += v[b * sv + i];
    A[b * sA + i * (ld + 1)] += add;
}

template <typename T>
__global__ __launch_bounds__(256) void copy2d_kernel(const T* __restrict__ src, int64_t lds, int64_t ss,
                                                     T* __restrict__ dst, int64_t ldd, int64_t sd,
                                                     int64_t rows, int64_t cols) {
    const int64_t b = blockIdx.z;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int64_t r0 = (int64_t)blockIdx.y * 8;
    for (int i = 0; i < 8; ++i) {
        const int64_t r = r0 + i;
        if (r < rows) dst[b * sd + r * ldd + c] = src[b * ss + r * lds + c];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void set_identity_kernel(T* __restrict__ dst, int64_t n, int64_t ld,
                                                           int64_t sd) {
    const int64_t b = blockIdx.z;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const int64_t r0 = (int64_t)blockIdx.y * 8;
    for (int i = 0; i < 8; ++i) {
        const int64_t r = r0 + i;
        if (r < n) dst[b * sd + r * ld + c] = (r == c) ? T(1) : T(0);
    }
}

// V[i][j] *= s[j]
template <typename T>
__global__ __launch_bounds__(256) void scale_cols_kernel(T* __restrict__ V, int64_t rows, int64_t cols,
                                                         int64_t ld, int64_t sV, const T* __restrict__ s,
                                                         int64_t ss) {
    const int64_t b = blockIdx.z;
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const T f = s[b * ss + c];
    const int64_t r0 = (int64_t)blockIdx.y * 8;
    for (int i = 0; i < 8; ++i) {
        const int64_t r = r0 + i;
        if (r < rows) V[b * sV + r * ld + c] *= f;
    }
}

// A[i][j] = A[j][i] for j > i (mirror the lower triangle), 32x32 LDS transposes
template <typename T>
__global__ __launch_bounds__(256) void symmetrize_kernel(T* __restrict__ A, int64_t n, int64_t ld,
                                                         int64_t sA) {
    __shared__ T tile[32][33];
    const int64_t b = blockIdx.z;
    const int bi = blockIdx.y, bj = blockIdx.x;   // source tile (rows bi, cols bj), need bj <= bi
    if (bj > bi) return;
    T* Ab = A + b * sA;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int64_t r = (int64_t)bi * 32 + i, c = (int64_t)bj * 32 + tx;
        tile[i][tx] = (r < n && c < n) ? Ab[r * ld + c] : T(0);
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int64_t r = (int64_t)bj * 32 + i, c = (int64_t)bi * 32 + tx;   // destination (transposed tile)
        if (r < n && c < n && c > r) Ab[r * ld + c] = tile[tx][i];
    }
}

}  // namespace

template <typename T>
int gpk_scale_cols_launch(T* V, int64_t rows, int64_t cols, int64_t ld, int64_t sV, const T* s, int64_t ss,
                          int64_t batch, hipStream_t stream) {
    if (rows <= 0 || cols <= 0 || batch <= 0) return GPK_OK;
    const int64_t gy = gpk_cdiv(rows, 8);
    if (gy > 65535 || batch > 65535) return GPK_ERR_ARG(3);
    dim3 g((unsigned)gpk_cdiv(cols, 256), (unsigned)gy, (unsigned)batch);
    hipLaunchKernelGGL((scale_cols_kernel<T>), g, dim3(256), 0, stream, V, rows, cols, ld, sV, s, ss);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

template <typename T>
int gpk_symmetrize_launch(T* A, int64_t n, int64_t ld, int64_t sA, int64_t batch, hipStream_t stream) {
    if (n <= 0 || batch <= 0) return GPK_OK;
    const int64_t t = gpk_cdiv(n, 32);
    if (t > 65535 || batch > 65535) return GPK_ERR_ARG(3);
    dim3 g((unsigned)t, (unsigned)t, (unsigned)batch);
    hipLaunchKernelGGL((symmetrize_kernel<T>), g, dim3(256), 0, stream, A, n, ld, sA);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

template <typename T>
int gpk_logdet_launch(const T* L, int64_t n, int64_t ld, int64_t sL, int64_t batch, T* out,
                      hipStream_t stream) {
    if (batch <= 0) return GPK_OK;
    if (batch <= 16 && n >= 4096)
        hipLaunchKernelGGL((logdet_kernel<T, 1024>), dim3((unsigned)batch), dim3(1024), 0, stream, L, n, ld, sL, out);
    else
        hipLaunchKernelGGL((logdet_kernel<T, 256>), dim3((unsigned)batch), dim3(256), 0, stream, L, n, ld, sL, out);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

int64_t gpk_colreduce_nchunks_impl(int64_t rows) {
    // 128-row chunks, but never more than 1024 chunks
    int64_t rch = 128;
    while (gpk_cdiv(rows, rch) > 1024) rch *= 2;
    return gpk_cdiv(rows > 0 ? rows : 1, rch);
}

// ws: 2 * batch * nchunk * cols elements
template <typename T>
int gpk_colreduce_launch(const T* V, int64_t rows, int64_t cols, int64_t ld, int64_t sV, const T* w,
                         int64_t sw, T* odot, T* oss, T* ws, int64_t batch, hipStream_t stream) {
    if (cols <= 0 || batch <= 0) return GPK_OK;
    if (batch > 65535) return GPK_ERR_ARG(11);
    const int nchunk = (int)gpk_colreduce_nchunks_impl(rows);
    const int rch = (int)gpk_cdiv(rows > 0 ? rows : 1, nchunk);
    T* pdot = ws;
    T* pss = ws + batch * nchunk * cols;
    dim3 g1((unsigned)gpk_cdiv(cols, 256), (unsigned)nchunk, (unsigned)batch);
    hipLaunchKernelGGL((colreduce_partial_kernel<T>), g1, dim3(256), 0, stream, V, rows, cols, ld, sV,
                       (odot != nullptr) ? w : (const T*)nullptr, sw, pdot, (oss != nullptr) ? pss : (T*)nullptr,
                       rch, nchunk);
    GPK_CHECK_LAUNCH();
    dim3 g2((unsigned)gpk_cdiv(cols, 256), (unsigned)batch);
    hipLaunchKernelGGL((colreduce_final_kernel<T>), g2, dim3(256), 0, stream, pdot, pss, cols, nchunk,
                       odot, oss);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

// Row reductions of Z (rows x n, row-major):  dot[i] = sum_k Z[i][k] w[k],  ss[i] = sum_k Z[i][k]^2  in one pass -- the posterior
// mean and marginal variance from the TRANSPOSED whitened cross-covariance K(x*, x) L^{-T} that gpk_potrf_rows leaves under the
// factor.  One wave per row, four 16-byte loads per lane in flight; fixed summation order.  HBM-bound: reads Z once.
template <typename T>
__global__ __launch_bounds__(256) void rowreduce_kernel(const T* __restrict__ Z, int64_t rows, int64_t n, int64_t ld, const T* __restrict__ w,
                                                         T* __restrict__ odot, T* __restrict__ oss, int vec_ok) {
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* __restrict__ zr = Z + row * ld;
    T dot = T(0), ss = T(0);
    if (vec_ok) {
        constexpr int U = 4;
        for (int64_t k = (int64_t)lane * VEC; k < n; k += 64 * VEC * U) {
            vec_t zv[U], wv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t kk = k + (int64_t)u * 64 * VEC;
                const bool ok = kk < n;
                zv[u] = *reinterpret_cast<const vec_t*>(zr + (ok ? kk : 0));
                if (w != nullptr) wv[u] = *reinterpret_cast<const vec_t*>(w + (ok ? kk : 0));
                if (!ok) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) zv[u][v] = T(0);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    if (w != nullptr) dot += zv[u][v] * wv[u][v];
                    ss += zv[u][v] * zv[u][v];
                }
        }
    } else {
        for (int64_t k = lane; k < n; k += 64) {
            const T z = zr[k];
            if (w != nullptr) dot += z * w[k];
            ss += z * z;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dot += __shfl_xor(dot, o, 64);
        ss += __shfl_xor(ss, o, 64);
    }
    if (lane == 0) {
        if (odot != nullptr) odot[row] = dot;
        if (oss != nullptr) oss[row] = ss;
    }
}

template <typename T>
int gpk_rowreduce_launch(const T* Z, int64_t rows, int64_t n, int64_t ld, const T* w, T* odot, T* oss, hipStream_t stream) {
    if (rows <= 0) return GPK_OK;
    if (n < 0 || ld < n) return GPK_ERR_ARG(3);
    if (odot != nullptr && w == nullptr) return GPK_ERR_ARG(5);
    constexpr int VEC = Traits<T>::VEC;
    const int vec_ok = ((uintptr_t)Z % 16 == 0) && (ld % VEC == 0) && (n % VEC == 0) && (w == nullptr || (uintptr_t)w % 16 == 0);
    hipLaunchKernelGGL((rowreduce_kernel<T>), dim3((unsigned)gpk_cdiv(rows, 4)), dim3(256), 0, stream, Z, rows, n, ld,
                       (odot != nullptr) ? w : (const T*)nullptr, odot, oss, vec_ok);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}
template int gpk_rowreduce_launch<double>(const double*, int64_t, int64_t, int64_t, const double*, double*, double*, hipStream_t);
template int gpk_rowreduce_launch<float>(const float*, int64_t, int64_t, int64_t, const float*, float*, float*, hipStream_t);

template <typename T>
int gpk_tril_launch(T* A, int64_t n, int64_t ld, int64_t sA, int64_t batch, hipStream_t stream) {
    if (n <= 0 || batch <= 0) return GPK_OK;
    dim3 g((unsigned)gpk_cdiv(n, 256), (unsigned)gpk_cdiv(n, 16), (unsigned)batch);
    hipLaunchKernelGGL((tril_kernel<T>), g, dim3(256), 0, stream, A, n, ld, sA);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

template <typename T>
int gpk_sum_lower_launch(const T* parts, int64_t nparts, int64_t n, int64_t ldp, int64_t sP, T* out, int64_t ldo, hipStream_t stream) {
    if (n <= 0 || nparts <= 0) return GPK_OK;
    if (nparts > INT32_MAX) return GPK_ERR_ARG(3);
    constexpr int VEC = Traits<T>::VEC;
    const int vec_ok = ((((uintptr_t)parts % 16 == 0) && (ldp % VEC == 0) && (sP % VEC == 0)) ? 1 : 0) | ((((uintptr_t)out % 16 == 0) && (ldo % VEC == 0)) ? 2 : 0);
    dim3 g((unsigned)gpk_cdiv(n, 256 * VEC), (unsigned)gpk_cdiv(n, 16));
    hipLaunchKernelGGL((sum_lower_kernel<T>), g, dim3(256), 0, stream, parts, (int)nparts, n, ldp, sP, out, ldo, vec_ok);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

template <typename T>
int gpk_add_diag_launch(T* A, int64_t n, int64_t ld, int64_t sA, T s, const T* v, int64_t sv,
                        int64_t batch, hipStream_t stream) {
    if (n <= 0 || batch <= 0) return GPK_OK;
    dim3 g((unsigned)gpk_cdiv(n, 256), (unsigned)batch);
    hipLaunchKernelGGL((add_diag_kernel<T>), g, dim3(256), 0, stream, A, n, ld, sA, s, v, sv);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

template <typename T>
int gpk_copy2d_launch(const T* src, int64_t lds, int64_t ss, T* dst, int64_t ldd, int64_t sd,
                      int64_t rows, int64_t cols, int64_t batch, hipStream_t stream) {
    if (rows <= 0 || cols <= 0 || batch <= 0) return GPK_OK;
    dim3 g((unsigned)gpk_cdiv(cols, 256), (unsigned)gpk_cdiv(rows, 8), (unsigned)batch);
    hipLaunchKernelGGL((copy2d_kernel<T>), g, dim3(256), 0, stream, src, lds, ss, dst, ldd, sd, rows, cols);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

template <typename T>
int gpk_set_identity_launch(T* dst, int64_t n, int64_t ld, int64_t sd, int64_t batch,
                            hipStream_t stream) {
    if (n <= 0 || batch <= 0) return GPK_OK;
    dim3 g((unsigned)gpk_cdiv(n, 256), (unsigned)gpk_cdiv(n, 8), (unsigned)batch);
    hipLaunchKernelGGL((set_identity_kernel<T>), g, dim3(256), 0, stream, dst, n, ld, sd);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}

#define GPK_INST(T)                                                                                  \
    template int gpk_scale_cols_launch<T>(T*, int64_t, int64_t, int64_t, int64_t, const T*, int64_t, \
                                          int64_t, hipStream_t);                                     \
    template int gpk_symmetrize_launch<T>(T*, int64_t, int64_t, int64_t, int64_t, hipStream_t);      \
    template int gpk_logdet_launch<T>(const T*, int64_t, int64_t, int64_t, int64_t, T*, hipStream_t); \
    template int gpk_colreduce_launch<T>(const T*, int64_t, int64_t, int64_t, int64_t, const T*,     \
                                         int64_t, T*, T*, T*, int64_t, hipStream_t);                 \
    template int gpk_tril_launch<T>(T*, int64_t, int64_t, int64_t, int64_t, hipStream_t);            \
    template int gpk_sum_lower_launch<T>(const T*, int64_t, int64_t, int64_t, int64_t, T*, int64_t, hipStream_t); \
    template int gpk_add_diag_launch<T>(T*, int64_t, int64_t, int64_t, T, const T*, int64_t, int64_t, \
                                        hipStream_t);                                                \
    template int gpk_copy2d_launch<T>(const T*, int64_t, int64_t, T*, int64_t, int64_t, int64_t,     \
                                      int64_t, int64_t, hipStream_t);                                \
    template int gpk_set_identity_launch<T>(T*, int64_t, int64_t, int64_t, int64_t, hipStream_t);
GPK_INST(double)
GPK_INST(float)
