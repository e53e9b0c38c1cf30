// gpk_solve.hip -- lower-triangular solves of the GP path, with no substitution
// on the critical path: gpk_potrf leaves inv(L_cc) for every 128x128 diagonal
// block, so   X = L^{-1} B   becomes a right-looking sweep of GEMMs (many
// right-hand sides: posterior conditioning, inducing-point V = L_z^{-1} K_zx)
// or GEMVs (few right-hand sides: the logpdf quadratic form).
//
// Replaces: `B.iqf_diag` / `B.iqf` / `B.solve` of the reference's dependency
// stack, i.e. LAPACK dtrsm/dtrsv under stheno/random.py:276 and
// stheno/model/observations.py:301,322,327,329,335 and inside
// mlkernels.PosteriorMean / PosteriorKernel (constructed observations.py:148-168).
//
// For many right-hand sides the 128-blocks are first merged into inverses of
// SB x SB diagonal blocks (SB = 256/512, recursive doubling, all GEMMs, batched
// over the blocks): [A 0; C D]^-1 = [Ai 0; -Di C Ai, Di].  Then per block row q:
//     T   = inv(L_qq) * B_q                 (GEMM, K = SB)
//     B_q = T;  B_below -= L[below, q] * T  (GEMM, K = SB)
// so every update is an MFMA GEMM with K = SB, not a rank-128 one.
#include "gpk_common.hpp"

GPK_KNOB(int, g_trsv_batched, 1);      // tuning knob (gpk_tune(17, v)): one-workgroup-per-matrix TRSV for batches of small factors
GPK_KNOB(int, g_trsv_sweep, 0);        // tuning knob (gpk_tune(49, v)): the single-column solve of one factor as ONE resident launch (trsv_sweep_kernel).
                                       // OFF: measured SLOWER than the per-block sweep (profiles/r05_ab_trsv_sweep.log: fp64 N = 16384, 1024-blocks 0.584 vs
                                       // 0.470 ms; fp32 N = 32768, 512-blocks 1.96 vs 1.15 ms) -- a grid barrier costs ~7 us (MI355X_MICROARCH.md, barrier-counter)
                                       // against ~1.5 us for a dependent kernel boundary, and the sweep needs 2 n / sb of either
GPK_KNOB(int64_t, g_trsv_sweep_from, 2048);   // tuning knob (gpk_tune(50, v)): ... from this order
void gpk_tune_solve(int key, int64_t value) {
    if (key == 17) GPK_KNOB_SET(g_trsv_batched = (int)value;);
    if (key == 49) GPK_KNOB_SET(g_trsv_sweep = (int)value;);
    if (key == 50) GPK_KNOB_SET(g_trsv_sweep_from = value;);
}

namespace {

// ---------------------------------------------------------------------------
// place the 128-blocks on the diagonal of the SB-blocks, identity-pad, zero rest
// ---------------------------------------------------------------------------
template <typename T>
__global__ void place_blocks_kernel(const T* __restrict__ d128, int64_t s128, int nblk128,
                                    T* __restrict__ dsb, int64_t ssb, int sb, int nsb) {
    const int64_t b = blockIdx.z;
    const int q = blockIdx.y;
    const int64_t per = (int64_t)sb * sb;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / sb), c = (int)(idx % sb);
        const int br = r / GPK_DB, bc = c / GPK_DB;
        T v = T(0);
        if (br == bc) {
            const int g = q * (sb / GPK_DB) + br;
            if (g < nblk128)
                v = d128[b * s128 + (int64_t)g * GPK_DB * GPK_DB + (r % GPK_DB) * GPK_DB + (c % GPK_DB)];
            else
                v = (r == c) ? T(1) : T(0);
        }
        dsb[b * ssb + q * per + idx] = v;
    }
}

// ---------------------------------------------------------------------------
// GEMV with a few right-hand sides:
//   y[i][c] = beta * y[i][c] + alpha * sum_k A[i][k] x[k][c],   i in [0, M)
// plus an optional prefix copy  ycopy[i][c] = x[i][c], i in [0, ncopy)  (writes a
// solved block back while the rows below are updated).
// One wave per row, lanes stride over K with 16-byte loads, x staged in LDS.
// ---------------------------------------------------------------------------
template <typename T>
struct GemvArgs {
    const T* A;
    int64_t lda, sA;
    const T* x;
    int64_t ldx, sx;
    T* y;
    int64_t ldy, sy;
    T* ycopy;          // nullable
    int64_t ncopy;
    int M, K, nrhs;
    T alpha, beta;
    int vec_ok;
};

constexpr int GEMV_ROWS = 16;    // rows per workgroup
// columns of x staged per pass: one right-hand side gets 4096 (32 KB in fp64) -- a long row (the pseudo-point path's V y: 200 000 columns)
// was 390 passes of two barriers with eight loads per lane between them, latency-bound at 3.4 TB/s; per-lane summation order unchanged
template <int NR>
struct GemvChunk { static constexpr int value = NR == 1 ? 4096 : (NR == 2 ? 2048 : 512); };

template <typename T, int NR>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs<T> p) {
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    constexpr int GEMV_MAXK = GemvChunk<NR>::value;
    __shared__ T xs[GEMV_MAXK * NR];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b = blockIdx.y;
    const T* __restrict__ A = p.A + b * p.sA;
    const T* __restrict__ x = p.x + b * p.sx;
    T* __restrict__ y = p.y + b * p.sy;

    const int row0 = blockIdx.x * GEMV_ROWS;
    constexpr int RPW = GEMV_ROWS / 4;   // rows per wave: wave, wave + 4, ...
    T acc[RPW][NR];
#pragma unroll
    for (int q = 0; q < RPW; ++q)
#pragma unroll
        for (int c = 0; c < NR; ++c) acc[q][c] = T(0);

    for (int kc = 0; kc < p.K || kc == 0; kc += GEMV_MAXK) {
        const int kb = (p.K - kc < GEMV_MAXK) ? p.K - kc : GEMV_MAXK;
        __syncthreads();
        for (int idx = tid; idx < kb * NR; idx += 256) {
            const int k = idx / NR, c = idx % NR;
            xs[idx] = (c < p.nrhs) ? x[(int64_t)(kc + k) * p.ldx + c] : T(0);
        }
        __syncthreads();

        // prefix copy (the rows of x in this chunk written back to ycopy)
        if (p.ycopy != nullptr && kc < p.ncopy) {
            T* __restrict__ yc = p.ycopy + b * p.sy;
            const int64_t nc = (p.ncopy - kc < kb) ? p.ncopy - kc : kb;
            for (int64_t idx = (int64_t)blockIdx.x * 256 + tid; idx < nc * p.nrhs;
                 idx += (int64_t)gridDim.x * 256) {
                const int64_t i = idx / p.nrhs;
                const int c = (int)(idx % p.nrhs);
                yc[(kc + i) * p.ldy + c] = xs[i * NR + c];
            }
        }

        // the wave's RPW rows advance together: RPW independent 16-byte loads in flight per lane
        const T* __restrict__ ar[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            int row = row0 + wave + 4 * q;
            if (row >= p.M) row = p.M > 0 ? p.M - 1 : 0;   // clamp: result discarded below
            ar[q] = A + (int64_t)row * p.lda + kc;
        }
        if (p.M > 0) {
            if (p.vec_ok) {
                for (int k = lane * VEC; k < kb; k += 64 * VEC) {
                    vec_t av[RPW];
#pragma unroll
                    for (int q = 0; q < RPW; ++q) av[q] = *reinterpret_cast<const vec_t*>(ar[q] + k);
#pragma unroll
                    for (int q = 0; q < RPW; ++q)
#pragma unroll
                        for (int v = 0; v < VEC; ++v)
#pragma unroll
                            for (int c = 0; c < NR; ++c) acc[q][c] += av[q][v] * xs[(k + v) * NR + c];
                }
            } else {
                for (int k = lane; k < kb; k += 64) {
#pragma unroll
                    for (int q = 0; q < RPW; ++q) {
                        const T av = ar[q][k];
#pragma unroll
                        for (int c = 0; c < NR; ++c) acc[q][c] += av * xs[k * NR + c];
                    }
                }
            }
        }
        if (p.K == 0) break;
    }

#pragma unroll
    for (int q = 0; q < RPW; ++q) {
        const int row = row0 + wave + 4 * q;
        if (row >= p.M) continue;
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            T v = acc[q][c];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            acc[q][c] = v;
        }
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                if (c < p.nrhs) {
                    T* yp = y + (int64_t)row * p.ldy + c;
                    T r = p.alpha * acc[q][c];
                    if (p.beta != T(0)) r += p.beta * (*yp);
                    *yp = r;
                }
            }
        }
    }
}

template <typename T>
int gemv_launch(int64_t M, int64_t K, int nrhs, T alpha, const T* A, int64_t lda, int64_t sA,
                const T* x, int64_t ldx, int64_t sx, T beta, T* y, int64_t ldy, int64_t sy, T* ycopy,
                int64_t ncopy, int64_t batch, hipStream_t stream) {
    if (nrhs > 8 || nrhs < 1) return GPK_ERR_ARG(3);
    if (M <= 0 && ncopy <= 0) return GPK_OK;
    if (batch > 65535) return GPK_ERR_ARG(17);
    constexpr int VEC = Traits<T>::VEC;
    GemvArgs<T> g;
    g.A = A; g.lda = lda; g.sA = sA;
    g.x = x; g.ldx = ldx; g.sx = sx;
    g.y = y; g.ldy = ldy; g.sy = sy;
    g.ycopy = ycopy; g.ncopy = ncopy;
    g.M = (int)(M > 0 ? M : 0); g.K = (int)K; g.nrhs = nrhs;
    g.alpha = alpha; g.beta = beta;
    g.vec_ok = ((uintptr_t)A % 16 == 0) && (lda % VEC == 0) && (sA % VEC == 0) && (K % VEC == 0);
    int64_t gx = gpk_cdiv(g.M, GEMV_ROWS);
    if (gx < 1) gx = 1;
    dim3 grid((unsigned)gx, (unsigned)batch);
    if (nrhs == 1)
        hipLaunchKernelGGL((gemv_kernel<T, 1>), grid, dim3(256), 0, stream, g);
    else if (nrhs == 2)
        hipLaunchKernelGGL((gemv_kernel<T, 2>), grid, dim3(256), 0, stream, g);
    else if (nrhs <= 4)
        hipLaunchKernelGGL((gemv_kernel<T, 4>), grid, dim3(256), 0, stream, g);
    else
        hipLaunchKernelGGL((gemv_kernel<T, 8>), grid, dim3(256), 0, stream, g);
    GPK_CHECK_LAUNCH();
    return GPK_OK;
}


// ---------------------------------------------------------------------------
// Batched TRSV, one workgroup per matrix (many small independent factors: stheno's batched computation): the whole solve of
// one right-hand-side set runs inside ONE workgroup -- no inter-workgroup dependency, one launch instead of 2 x n/128 --
// left-looking over 128-blocks:   r_q -= L[q, 0:q] x[0:q]  (rows read once, contiguously),   x_q = inv(L_qq) r_q.
// x lives in LDS; a wave owns eight rows at a time (lanes across the columns, 16-byte loads, wave reduction).  HBM-bound: the
// lower triangle and the inverted diagonal blocks are read once.  nrhs <= NR.
// ---------------------------------------------------------------------------
template <typename T, int NR>
__global__ __launch_bounds__(256) void trsv_batched_kernel(const T* __restrict__ L, int64_t ld, int64_t sL, const T* __restrict__ dinv,
                                                            int64_t sD, T* __restrict__ B, int64_t ldb, int64_t sB, int n, int nrhs) {
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) char trsv_smem[];
    T* xs = reinterpret_cast<T*>(trsv_smem);                 // [n_pad][NR]: the right-hand sides, overwritten by the solution
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t b = blockIdx.x;
    const T* __restrict__ Lb = L + b * sL;
    const T* __restrict__ Wb = dinv + b * sD;
    T* __restrict__ Bb = B + b * sB;
    const int nblk = (n + GPK_DB - 1) / GPK_DB;
    for (int idx = tid; idx < nblk * GPK_DB * NR; idx += 256) {
        const int i = idx / NR, c = idx % NR;
        xs[idx] = (i < n && c < nrhs) ? Bb[(int64_t)i * ldb + c] : T(0);
    }
    __syncthreads();
    const bool vec_ok = ((uintptr_t)Lb % 16 == 0) && (ld % VEC == 0);
    for (int q = 0; q < nblk; ++q) {
        const int r0 = q * GPK_DB;
        // 1. r_q -= L[q-block rows, 0:r0] x[0:r0]
        if (q > 0) {
            // RU rows per wave at a time: one WG streams its matrix alone (two WGs per CU at a batch of 512), so what it gets out of
            // the memory system is what it keeps in flight -- RU 16-byte loads per lane instead of one (round 4: 512 x 2048^2 fp32
            // 1.53 -> see profiles/r04_experiments.md section 13)
            constexpr int RU = 8;
            for (int g = 0; g < GPK_DB / (4 * RU); ++g) {
                const int rbase = r0 + (g * 4 + wave) * RU;          // rows rbase .. rbase + RU - 1 (uniform per wave)
                if (rbase >= n) break;
                T acc[RU][NR];
#pragma unroll
                for (int u = 0; u < RU; ++u)
#pragma unroll
                    for (int c = 0; c < NR; ++c) acc[u][c] = T(0);
                if (vec_ok) {
                    for (int k = lane * VEC; k < r0; k += 64 * VEC) {
                        vec_t lv[RU];
#pragma unroll
                        for (int u = 0; u < RU; ++u) {
                            const int row = (rbase + u < n) ? rbase + u : n - 1;      // (past the end: a valid row, result dropped)
                            lv[u] = *reinterpret_cast<const vec_t*>(Lb + (int64_t)row * ld + k);
                        }
#pragma unroll
                        for (int v = 0; v < VEC; ++v)
#pragma unroll
                            for (int c = 0; c < NR; ++c) {
                                const T xv = xs[(k + v) * NR + c];
#pragma unroll
                                for (int u = 0; u < RU; ++u) acc[u][c] += lv[u][v] * xv;
                            }
                    }
                } else {
                    for (int k = lane; k < r0; k += 64)
#pragma unroll
                        for (int u = 0; u < RU; ++u) {
                            const int row = (rbase + u < n) ? rbase + u : n - 1;
                            const T lv = Lb[(int64_t)row * ld + k];
#pragma unroll
                            for (int c = 0; c < NR; ++c) acc[u][c] += lv * xs[k * NR + c];
                        }
                }
#pragma unroll
                for (int u = 0; u < RU; ++u)
#pragma unroll
                    for (int c = 0; c < NR; ++c) {
                        T v = acc[u][c];
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                        acc[u][c] = v;
                    }
                if (lane == 0) {
#pragma unroll
                    for (int u = 0; u < RU; ++u)
                        if (rbase + u < n) {
#pragma unroll
                            for (int c = 0; c < NR; ++c) xs[(rbase + u) * NR + c] -= acc[u][c];
                        }
                }
            }
            __syncthreads();
        }
        // 2. x_q = inv(L_qq) r_q   (128 x 128 block, identity-padded past n): results held until every wave has read r_q
        const T* __restrict__ Wq = Wb + (int64_t)q * GPK_DB * GPK_DB;
        // (two rows per wave at a time, one per half-wave: a 128-element row is 32 lanes of 16 bytes in fp32)
        const int half = lane >> 5, l32 = lane & 31;
        T res[GPK_DB / 8][NR];
#pragma unroll
        for (int j = 0; j < GPK_DB / 8; ++j) {
            const int rr = wave + 4 * (2 * j + half);
            T acc[NR];
#pragma unroll
            for (int c = 0; c < NR; ++c) acc[c] = T(0);
#pragma unroll
            for (int k = l32 * VEC; k < GPK_DB; k += 32 * VEC) {
                const vec_t wv = *reinterpret_cast<const vec_t*>(Wq + (int64_t)rr * GPK_DB + k);
#pragma unroll
                for (int v = 0; v < VEC; ++v)
#pragma unroll
                    for (int c = 0; c < NR; ++c) acc[c] += wv[v] * xs[(r0 + k + v) * NR + c];
            }
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                T v = acc[c];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                res[j][c] = v;
            }
        }
        __syncthreads();
        if (l32 == 0) {
#pragma unroll
            for (int j = 0; j < GPK_DB / 8; ++j)
#pragma unroll
                for (int c = 0; c < NR; ++c) xs[(r0 + wave + 4 * (2 * j + half)) * NR + c] = res[j][c];
        }
        __syncthreads();
    }
    for (int idx = tid; idx < n * nrhs; idx += 256) {
        const int i = idx / nrhs, c = idx % nrhs;
        Bb[(int64_t)i * ldb + c] = xs[i * NR + c];
    }
}

// ---------------------------------------------------------------------------
// ONE right-hand side, one (large) factor: the whole forward solve in ONE launch (round 5; the per-block sweep below is 2 n / sb
// launches of 10-40 us each with ~7 us gaps between them: 0.69 ms for 0.21 ms of HBM work at cfg2).  A resident grid of G
// workgroups (one per CU) walks the sb-blocks LEFT-LOOKING; row i of a block belongs to workgroup i mod G:
//   phase 1   r_i = b_i - L[i, 0:r0] x[0:r0]      every workgroup streams ITS rows of block row q (the four waves split k, RU rows and
//                                                 8 / RU k-slices in flight per lane), r -> R (scratch)
//   phase 2   x_i = W_q[i, 0:i] r[0:i]            the explicit inverse of the diagonal block, same rows, x -> B
// with a grid barrier (one counter, agent scope: stores -> workgroup barrier -> release -> add; poll -> acquire -> barrier) behind
// each phase.  Both phases use the whole chip, so the serial chain is 2 n / sb barriers of a few us.  Fixed summation order
// (lane -> wave -> the four waves in order): deterministic.  A poll that spins for seconds (the grid is not co-resident: a
// CU-masked stream) raises ctrl[1]; everybody leaves and workgroup 0 poisons the result with NaN.
// ---------------------------------------------------------------------------
template <typename T, int RU>
__global__ __launch_bounds__(256) void trsv_sweep_kernel(const T* __restrict__ L, int64_t ld, const T* __restrict__ W, int sb, T* B, T* R,
                                                          unsigned* ctrl, int n) {
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    constexpr int U = 8 / RU;                    // k-slices in flight per lane
    constexpr int S = 256 * VEC;                 // columns one pass of the workgroup covers
    __shared__ T red[4][RU];
    __shared__ int s_abort;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = (int)gridDim.x, w = (int)blockIdx.x;
    const int nsb = (n + sb - 1) / sb;
    unsigned epoch = 0;
    auto grid_barrier = [&]() -> bool {          // false: give up (abort raised)
        ++epoch;
        gpk_barrier_stores_done();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(&ctrl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = epoch * (unsigned)G;
            int ab = 0;
            long long t0 = 0;
            for (unsigned spins = 0;; ++spins) {
                if (__hip_atomic_load(&ctrl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) break;
                if (__hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ab = 1; break; }
                if ((spins & 1023u) == 1023u) {
                    const long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    else if (now - t0 > 400000000ll) {       // ~4 s at 100 MHz
                        __hip_atomic_store(&ctrl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ab = 1;
                        break;
                    }
                }
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_abort = ab;
        }
        __syncthreads();
        return s_abort == 0;
    };
    bool alive = true;
    for (int q = 0; q < nsb && alive; ++q) {
        const int r0 = q * sb;
        const int rq = (n - r0 < sb) ? n - r0 : sb;
        const T* __restrict__ Wq = W + (int64_t)q * sb * sb;
        // ---- phase 1 ----
        for (int g0 = w; g0 < rq; g0 += G * RU) {                 // this workgroup's rows g0, g0 + G, ... of the block, RU at a time
            int rows[RU];
            const T* lrow[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                rows[u] = g0 + u * G;
                const int rc = rows[u] < rq ? rows[u] : g0;        // (past the end: a valid row, result dropped)
                lrow[u] = L + (int64_t)(r0 + rc) * ld;
            }
            T acc[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) acc[u] = T(0);
            for (int k = tid * VEC; k < r0; k += S * U) {
                vec_t lv[U][RU], xv[U];
#pragma unroll
                for (int uu = 0; uu < U; ++uu) {
                    const int kk = k + uu * S;
                    const bool ok = kk < r0;
                    const int kc = ok ? kk : 0;
                    xv[uu] = *reinterpret_cast<const vec_t*>(B + kc);
                    if (!ok) {
#pragma unroll
                        for (int v = 0; v < VEC; ++v) xv[uu][v] = T(0);
                    }
#pragma unroll
                    for (int u = 0; u < RU; ++u) lv[uu][u] = *reinterpret_cast<const vec_t*>(lrow[u] + kc);
                }
#pragma unroll
                for (int uu = 0; uu < U; ++uu)
#pragma unroll
                    for (int v = 0; v < VEC; ++v)
#pragma unroll
                        for (int u = 0; u < RU; ++u) acc[u] += lv[uu][u][v] * xv[uu][v];
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                T v = acc[u];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) red[wave][u] = v;
            }
            __syncthreads();
            if (tid < RU && g0 + tid * G < rq) {
                const int i = g0 + tid * G;
                R[i] = B[r0 + i] - (((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid]);
            }
            __syncthreads();
        }
        alive = grid_barrier();
        if (!alive) break;
        // ---- phase 2 ----
        for (int g0 = w; g0 < rq; g0 += G * RU) {
            int rows[RU];
            const T* wrow[RU];
            int kmax = 0;
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                rows[u] = g0 + u * G;
                const int rc = rows[u] < rq ? rows[u] : g0;
                wrow[u] = Wq + (int64_t)rc * sb;
                if (rows[u] < rq) kmax = rows[u] + 1;
            }
            kmax = (kmax + VEC - 1) / VEC * VEC;
            T acc[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) acc[u] = T(0);
            for (int k = tid * VEC; k < kmax; k += S) {
                const vec_t rv = *reinterpret_cast<const vec_t*>(R + k);
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const vec_t wv = *reinterpret_cast<const vec_t*>(wrow[u] + k);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[u] += (k + v <= rows[u]) ? wv[v] * rv[v] : T(0);     // (the inverse is lower triangular)
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                T v = acc[u];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) red[wave][u] = v;
            }
            __syncthreads();
            if (tid < RU && g0 + tid * G < rq) {
                const int i = g0 + tid * G;
                B[r0 + i] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
            }
            __syncthreads();
        }
        alive = grid_barrier();
    }
    if (!alive && w == 0) {
        T qnan = T(0) / T(0);
        asm volatile("" : "+v"(qnan));
        for (int i = tid; i < n; i += 256) B[i] = qnan;
    }
}

}  // namespace

// y[M x nrhs] = alpha * A[M x K] x[K x nrhs] + beta * y, nrhs <= 8, A row-major (k contiguous)
template <typename T>
int gpk_gemv_launch(int64_t M, int64_t K, int nrhs, T alpha, const T* A, int64_t lda, int64_t sA,
                    const T* x, int64_t ldx, int64_t sx, T beta, T* y, int64_t ldy, int64_t sy,
                    int64_t batch, hipStream_t stream) {
    if (M > INT32_MAX || K > INT32_MAX) return GPK_ERR_ARG(1);
    return gemv_launch<T>(M, K, nrhs, alpha, A, lda, sA, x, ldx, sx, beta, y, ldy, sy, (T*)nullptr, 0, batch,
                          stream);
}

// One 128-block step of the single-column sweep of a BATCH of factors (right-looking, as the loop of gpk_trsv_launch):
//   tmp_b = inv(L_qq) b_q;   b_q = tmp_b;   b[below] -= L[below, q] tmp_b        for every matrix b of the batch
// dinv: the 128-block inverses [batch][nblk][128][128] (stride sD); B: one vector of n entries per matrix (stride sB); tmp: batch * 128.
template <typename T>
int gpk_trsv_batch_step_launch(const T* L, int64_t n, int64_t ld, int64_t sL, const T* dinv, int64_t sD, T* B, int64_t sB, T* tmp,
                               int64_t batch, int q, hipStream_t stream) {
    const int64_t r0 = (int64_t)q * GPK_DB;
    if (r0 >= n) return GPK_OK;
    const int64_t rq = (n - r0 < GPK_DB) ? n - r0 : GPK_DB;
    int st = gemv_launch<T>(rq, rq, 1, T(1), dinv + (int64_t)q * (GPK_DB * GPK_DB), GPK_DB, sD, B + r0, 1, sB, T(0), tmp, 1, GPK_DB,
                            (T*)nullptr, 0, batch, stream);
    if (st) return st;
    const int64_t r1 = r0 + rq;
    return gemv_launch<T>(n - r1, rq, 1, T(-1), L + r1 * ld + r0, ld, sL, tmp, 1, GPK_DB, T(1), B + r1, 1, sB, B + r0, rq, batch, stream);
}

template <typename T>
int gpk_trsv_step_launch(const T* W, int64_t ldw, int64_t rq, const T* Lbelow, int64_t ld, int64_t nbelow, T* bq, T* bbelow, T* tmp,
                         hipStream_t stream) {
    if (rq > INT32_MAX || nbelow > INT32_MAX) return GPK_ERR_ARG(1);
    // tmp = W bq
    int st = gemv_launch<T>(rq, rq, 1, T(1), W, ldw, 0, bq, 1, 0, T(0), tmp, 1, 0, (T*)nullptr, 0, 1, stream);
    if (st) return st;
    // bq = tmp;  bbelow -= Lbelow tmp
    return gemv_launch<T>(nbelow, rq, 1, T(-1), Lbelow, ld, 0, tmp, 1, 0, T(1), bbelow, 1, 0, bq, rq, 1, stream);
}

// ---------------------------------------------------------------------------
// merge inv(L_cc) 128-blocks into inverses of SB x SB diagonal blocks
//   dinv_sb : [batch][nsb][sb][sb], nsb = ceil(n / sb)
//   tmp     : at least nsb * sb * sb / 4 elements
// ---------------------------------------------------------------------------
template <typename T>
int gpk_trtri_merge_launch(const T* L, int64_t n, int64_t ld, int64_t batch, int64_t bstride,
                           const T* dinv128, int sb, T* dinv_sb, T* tmp, hipStream_t stream) {
    if (n <= 0 || batch <= 0) return GPK_OK;
    if (!gpk_valid_sb(sb)) return GPK_ERR_ARG(9);
    const int nblk128 = (int)gpk_cdiv(n, GPK_DB);
    const int64_t s128 = (int64_t)nblk128 * GPK_DB * GPK_DB;
    const int nsb = (int)gpk_cdiv(n, sb);
    const int64_t per = (int64_t)sb * sb;
    const int64_t ssb = (int64_t)nsb * per;
    if (batch > 65535) return GPK_ERR_ARG(4);
    {
        dim3 grid((unsigned)gpk_cdiv(per, 256 * 4), (unsigned)nsb, (unsigned)batch);
        hipLaunchKernelGGL((place_blocks_kernel<T>), grid, dim3(256), 0, stream, dinv128, s128, nblk128,
                           dinv_sb, ssb, sb, nsb);
        GPK_CHECK_LAUNCH();
    }
    const int nfull = (int)(n / sb);
    for (int h = GPK_DB; h < sb; h *= 2) {
        const int ppb = sb / (2 * h);          // pairs per SB-block
        const int64_t hh = (int64_t)h * h;
        for (int64_t b = 0; b < batch; ++b) {
            const T* Lb = L + b * bstride;
            T* Db = dinv_sb + b * ssb;
            if (nfull > 0) {
                // T = L21 * Dinv_lo   (pairs: blockIdx.y, SB-blocks: blockIdx.z)
                int st = gpk_gemm_launch2<T>(true, false, h, h, h, T(1), Lb + (int64_t)h * ld, ld,
                                             2 * h * (ld + 1), (int64_t)sb * (ld + 1), Db, sb,
                                             2 * h * (int64_t)(sb + 1), per, T(0), tmp, h, hh, ppb * hh, ppb,
                                             nfull, false, stream);
                if (st) return st;
                // C' = -Dinv_hi * T
                st = gpk_gemm_launch2<T>(true, false, h, h, h, T(-1), Db + (int64_t)h * (sb + 1), sb,
                                         2 * h * (int64_t)(sb + 1), per, tmp, h, hh, ppb * hh, T(0),
                                         Db + (int64_t)h * sb, sb, 2 * h * (int64_t)(sb + 1), per, ppb, nfull,
                                         false, stream);
                if (st) return st;
            }
            if (nfull < nsb) {   // ragged last SB-block: pair by pair, rows clipped to n
                const int q = nfull;
                T* tq = tmp + (int64_t)nfull * ppb * hh;
                for (int pr = 0; pr < ppb; ++pr) {
                    const int64_t o = (int64_t)q * sb + (int64_t)pr * 2 * h;   // global row/col of the pair
                    const int64_t m = n - (o + h);
                    if (m <= 0) break;
                    const int64_t mv = m < h ? m : h;
                    const int64_t oo = (int64_t)pr * 2 * h;                    // offset inside the SB-block
                    T* Dq = Db + q * per;
                    if (hipMemsetAsync(tq, 0, hh * sizeof(T), stream) != hipSuccess) return GPK_ERR_LAUNCH;
                    int st = gpk_gemm_launch<T>(true, false, mv, h, h, T(1), Lb + (o + h) * ld + o, ld, 0,
                                                Dq + oo * (sb + 1), sb, 0, T(0), tq, h, 0, 1, false, stream);
                    if (st) return st;
                    st = gpk_gemm_launch<T>(true, false, h, h, h, T(-1), Dq + (oo + h) * (sb + 1), sb, 0, tq,
                                            h, 0, T(0), Dq + (oo + h) * sb + oo, sb, 0, 1, false, stream);
                    if (st) return st;
                }
            }
        }
    }
    return GPK_OK;
}

// ---------------------------------------------------------------------------
// X = L^{-1} B, many right-hand sides: RECURSIVE blocked solve over the sb-blocks of L.
//   solve(q0, q1):  one block:  X_q = inv(L_qq) B_q                       (GEMM with the merged inverse, k clipped to its triangle)
//                   else:       solve(q0, m);  B[m:q1] -= L[m:q1, q0:m] X[q0:m]  (ONE GEMM, K = all of q0..m);  solve(m, q1)
// Against the right-looking sweep (one update of everything below per block, K = sb) the flops are the same, but half of them
// sit in a single GEMM with K = n/2 and a quarter in two with K = n/4: every C tile is read and written log2(n/sb) times instead of
// n/sb times, launches have exact tile counts more often (cfg2: 8192 x 2048 x 8192 = exactly two rounds of 128-tiles; the sweep's
// 1664-tile updates paid 4 rounds for 3.25), and the long k loops run at the large-K rate of the tile kernel.
//   dinv_sb: [batch][nsb][sb][sb].
//   X == nullptr: in place -- B is overwritten with the solution; tmp: [batch][sb][nrhs] (a solved block is formed there, then copied).
//   X != nullptr: out of place -- solved blocks go to X ([batch] n x nrhs, ldx, stride sX), B is used up as workspace; no copies.
// ---------------------------------------------------------------------------
namespace {
template <typename T>
struct TrsmCtx {
    const T* L; int64_t n, ld, sL;
    const T* dinv; int sb; int64_t per, ssb;
    T* B; int64_t nrhs, ldb, sB;
    T* X; int64_t ldx, sX;         // where solved blocks live (== B, ldb, sB in place)
    T* tmp; int64_t st_tmp;
    int64_t batch; hipStream_t stream;
};
template <typename T>
int trsm_rec(const TrsmCtx<T>& c, int q0, int q1) {
    const int64_t r0 = (int64_t)q0 * c.sb;
    if (q1 - q0 == 1) {
        const int64_t rq = (c.n - r0 < c.sb) ? c.n - r0 : c.sb;
        const bool inplace = (c.X == c.B);
        T* dst = inplace ? c.tmp : c.X + r0 * c.ldx;
        int st = gpk_gemm_launch<T>(true, false, rq, c.nrhs, rq, T(1), c.dinv + q0 * c.per, c.sb, c.ssb, c.B + r0 * c.ldb, c.ldb, c.sB, T(0),
                                    dst, inplace ? c.nrhs : c.ldx, inplace ? c.st_tmp : c.sX, c.batch, 4, c.stream);   // inv(L_qq) is lower triangular
        if (st || !inplace) return st;
        return gpk_copy2d_launch<T>(c.tmp, c.nrhs, c.st_tmp, c.B + r0 * c.ldb, c.ldb, c.sB, rq, c.nrhs, c.batch, c.stream);
    }
    int h = 1;                                   // split at the largest power of two below the count: the big GEMMs get power-of-two K
    while (2 * h < q1 - q0) h *= 2;
    const int m = q0 + h;
    int st = trsm_rec<T>(c, q0, m);
    if (st) return st;
    const int64_t rm = (int64_t)m * c.sb;
    const int64_t r1 = ((int64_t)q1 * c.sb < c.n) ? (int64_t)q1 * c.sb : c.n;
    st = gpk_gemm_launch<T>(true, false, r1 - rm, c.nrhs, rm - r0, T(-1), c.L + rm * c.ld + r0, c.ld, c.sL, c.X + r0 * c.ldx, c.ldx, c.sX, T(1),
                            c.B + rm * c.ldb, c.ldb, c.sB, c.batch, 0, c.stream);
    if (st) return st;
    return trsm_rec<T>(c, m, q1);
}
}  // namespace

template <typename T>
int gpk_trsm_launch(const T* L, int64_t n, int64_t ld, int64_t sL, const T* dinv_sb, int sb, T* B,
                    int64_t nrhs, int64_t ldb, int64_t sB, T* tmp, int64_t batch, hipStream_t stream, T* X, int64_t ldx, int64_t sX) {
    if (n <= 0 || nrhs <= 0 || batch <= 0) return GPK_OK;
    if (!gpk_valid_sb(sb)) return GPK_ERR_ARG(6);
    if (X == nullptr && tmp == nullptr) return GPK_ERR_ARG(11);
    if (X != nullptr && ldx < nrhs) return GPK_ERR_ARG(15);
    const int nsb = (int)gpk_cdiv(n, sb);
    TrsmCtx<T> c;
    c.L = L; c.n = n; c.ld = ld; c.sL = sL;
    c.dinv = dinv_sb; c.sb = sb; c.per = (int64_t)sb * sb; c.ssb = (int64_t)nsb * c.per;
    c.B = B; c.nrhs = nrhs; c.ldb = ldb; c.sB = sB;
    c.X = X ? X : B; c.ldx = X ? ldx : ldb; c.sX = X ? sX : sB;
    c.tmp = tmp; c.st_tmp = (int64_t)sb * nrhs;
    c.batch = batch; c.stream = stream;
    return trsm_rec<T>(c, 0, nsb);
}

// ---------------------------------------------------------------------------
// B <- L^{-1} B, nrhs <= 8 (GEMV sweep; HBM-bound: reads the lower triangle once)
//   tmp: [batch][sb][nrhs]
// ---------------------------------------------------------------------------
template <typename T>
int gpk_trsv_launch(const T* L, int64_t n, int64_t ld, int64_t sL, const T* dinv_sb, int sb, T* B,
                    int nrhs, int64_t ldb, int64_t sB, T* tmp, int64_t batch, hipStream_t stream) {
    if (n <= 0 || nrhs <= 0 || batch <= 0) return GPK_OK;
    if (!gpk_valid_sb(sb)) return GPK_ERR_ARG(6);
    if (nrhs > 8) return GPK_ERR_ARG(8);
    if (batch > 65535) return GPK_ERR_ARG(12);
    const int nsb = (int)gpk_cdiv(n, sb);
    const int64_t per = (int64_t)sb * sb, ssb = (int64_t)nsb * per;
    const int64_t st_tmp = (int64_t)sb * nrhs;
    // many small factors with the 128-block inverses: one workgroup per matrix, one launch
    if (g_trsv_batched && sb == GPK_DB && batch >= 64 && nrhs <= (sizeof(T) == 8 ? 2 : 4)) {      // (fp64 x 4 columns would spill)
        const int NRv = nrhs == 1 ? 1 : (nrhs == 2 ? 2 : 4);
        const size_t lds = (size_t)nsb * GPK_DB * NRv * sizeof(T);
        if (lds <= 64 * 1024) {
            dim3 grid((unsigned)batch);
            if (NRv == 1)
                hipLaunchKernelGGL((trsv_batched_kernel<T, 1>), grid, dim3(256), lds, stream, L, ld, sL, dinv_sb, ssb, B, ldb, sB, (int)n, nrhs);
            else if (NRv == 2)
                hipLaunchKernelGGL((trsv_batched_kernel<T, 2>), grid, dim3(256), lds, stream, L, ld, sL, dinv_sb, ssb, B, ldb, sB, (int)n, nrhs);
            else
                hipLaunchKernelGGL((trsv_batched_kernel<T, 4>), grid, dim3(256), lds, stream, L, ld, sL, dinv_sb, ssb, B, ldb, sB, (int)n, nrhs);
            GPK_CHECK_LAUNCH();
            return GPK_OK;
        }
    }
    // one right-hand side, one factor of several blocks: the whole sweep in one resident launch (trsv_sweep_kernel)
    constexpr int VEC = Traits<T>::VEC;
    if (g_trsv_sweep && batch == 1 && nrhs == 1 && ldb == 1 && nsb >= 2 && sb >= 256 && n >= g_trsv_sweep_from &&
        (uintptr_t)L % 16 == 0 && ld % VEC == 0 && (uintptr_t)B % 16 == 0 && (uintptr_t)tmp % 16 == 0 && (uintptr_t)dinv_sb % 16 == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 16) {
            unsigned* ctrl = reinterpret_cast<unsigned*>(tmp + st_tmp);        // (behind the sb elements of r: gpk.h, GPK_TRSV_CTRL_ELEMS)
            if (hipMemsetAsync(ctrl, 0, 2 * sizeof(unsigned), stream) != hipSuccess) return GPK_ERR_LAUNCH;
            const int ru = sb / cus >= 4 ? 4 : (sb / cus >= 2 ? 2 : 1);
            const int G = (sb / ru < cus) ? sb / ru : cus;
            dim3 grid((unsigned)G);
            if (ru == 4) hipLaunchKernelGGL((trsv_sweep_kernel<T, 4>), grid, dim3(256), 0, stream, L, ld, dinv_sb, sb, B, tmp, ctrl, (int)n);
            else if (ru == 2) hipLaunchKernelGGL((trsv_sweep_kernel<T, 2>), grid, dim3(256), 0, stream, L, ld, dinv_sb, sb, B, tmp, ctrl, (int)n);
            else hipLaunchKernelGGL((trsv_sweep_kernel<T, 1>), grid, dim3(256), 0, stream, L, ld, dinv_sb, sb, B, tmp, ctrl, (int)n);
            GPK_CHECK_LAUNCH();
            return GPK_OK;
        }
    }
    for (int q = 0; q < nsb; ++q) {
        const int64_t r0 = (int64_t)q * sb;
        const int64_t rq = (n - r0 < sb) ? n - r0 : sb;
        // tmp = inv(L_qq) b_q
        int st = gemv_launch<T>(rq, rq, nrhs, T(1), dinv_sb + q * per, sb, ssb, B + r0 * ldb, ldb, sB, T(0),
                                tmp, nrhs, st_tmp, (T*)nullptr, 0, batch, stream);
        if (st) return st;
        // b_q = tmp;  b_below -= L[below, q] tmp
        const int64_t r1 = r0 + rq;
        st = gemv_launch<T>(n - r1, rq, nrhs, T(-1), L + r1 * ld + r0, ld, sL, tmp, nrhs, st_tmp, T(1),
                            B + r1 * ldb, ldb, sB, B + r0 * ldb, rq, batch, stream);
        if (st) return st;
    }
    return GPK_OK;
}

// ---------------------------------------------------------------------------
// W <- L^{-1} (full lower-triangular inverse, n x n): the TRSM sweep on the identity,
// restricted at block row q to the columns that can be non-zero (0 .. end of block q),
// i.e. N^3/3 flops instead of N^3.  tmp: sb * n elements.
// ---------------------------------------------------------------------------
template <typename T>
int gpk_trtri_launch(const T* L, int64_t n, int64_t ld, const T* dinv_sb, int sb, T* W, int64_t ldw, T* tmp,
                     hipStream_t stream) {
    if (n <= 0) return GPK_OK;
    if (!gpk_valid_sb(sb)) return GPK_ERR_ARG(6);
    int st = gpk_set_identity_launch<T>(W, n, ldw, 0, 1, stream);
    if (st) return st;
    const int nsb = (int)gpk_cdiv(n, sb);
    const int64_t per = (int64_t)sb * sb;
    for (int q = 0; q < nsb; ++q) {
        const int64_t r0 = (int64_t)q * sb;
        const int64_t rq = (n - r0 < sb) ? n - r0 : sb;
        const int64_t nc = r0 + rq;          // columns that can be non-zero in block row q
        st = gpk_gemm_launch<T>(true, false, rq, nc, rq, T(1), dinv_sb + q * per, sb, 0, W + r0 * ldw, ldw, 0, T(0),
                                tmp, nc, 0, 1, 4, stream);   // inv(L_qq) is lower triangular
        if (st) return st;
        st = gpk_copy2d_launch<T>(tmp, nc, 0, W + r0 * ldw, ldw, 0, rq, nc, 1, stream);
        if (st) return st;
        const int64_t r1 = r0 + rq;
        if (r1 < n) {
            st = gpk_gemm_launch<T>(true, false, n - r1, nc, rq, T(-1), L + r1 * ld + r0, ld, 0, tmp, nc, 0, T(1),
                                    W + r1 * ldw, ldw, 0, 1, 0, stream);
            if (st) return st;
        }
    }
    return GPK_OK;
}

#define GPK_INST(T)                                                                                 \
    template int gpk_trtri_launch<T>(const T*, int64_t, int64_t, const T*, int, T*, int64_t, T*, hipStream_t); \
    template int gpk_trsv_step_launch<T>(const T*, int64_t, int64_t, const T*, int64_t, int64_t, T*, T*, T*, hipStream_t);  \
    template int gpk_trsv_batch_step_launch<T>(const T*, int64_t, int64_t, int64_t, const T*, int64_t, T*, int64_t, T*, int64_t, int, hipStream_t);  \
    template int gpk_gemv_launch<T>(int64_t, int64_t, int, T, const T*, int64_t, int64_t, const T*,  \
                                    int64_t, int64_t, T, T*, int64_t, int64_t, int64_t, hipStream_t); \
    template int gpk_trtri_merge_launch<T>(const T*, int64_t, int64_t, int64_t, int64_t, const T*,  \
                                           int, T*, T*, hipStream_t);                               \
    template int gpk_trsm_launch<T>(const T*, int64_t, int64_t, int64_t, const T*, int, T*, int64_t, \
                                    int64_t, int64_t, T*, int64_t, hipStream_t, T*, int64_t, int64_t); \
    template int gpk_trsv_launch<T>(const T*, int64_t, int64_t, int64_t, const T*, int, T*, int,    \
                                    int64_t, int64_t, T*, int64_t, hipStream_t);
GPK_INST(double)
GPK_INST(float)
