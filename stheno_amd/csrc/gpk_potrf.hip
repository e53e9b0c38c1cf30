// gpk_potrf.hip -- right-looking blocked Cholesky  A = L L^T  (lower, row-major,
// in place), batched, f64/f32, for gfx950.
//
// Replaces: `B.cholesky(B.reg(K))` of the reference's dependency stack, i.e.
// LAPACK dpotrf/spotrf under `B.logdet` / `B.iqf_diag` (stheno/random.py:274-276)
// and under `B.cholesky(K_z)` (stheno/model/observations.py:300).
//
// Structure (gpk_potrf; the look-ahead variant for one large matrix is described further down):
//   ONE matrix -- per outer panel of `nbo` columns (the whole matrix up to n = 4096, 1024 above) ONE launch, potrf_pipe_kernel:
//     a chain workgroup factorises AND inverts the 128x128 diagonal blocks one after the other entirely in LDS (diag3_block:
//     16-wide micro-panels factorised as panels by shuffle-based waves, MFMA rank-16 updates, the inverse grown row block by row
//     block in the shadow of the factorisation); every other workgroup takes the solves below the blocks (GEMMs against the
//     inverted block) and the rank-128 updates inside the panel as tasks from a counter, synchronised through flag words
//     (gpk_potrf_pipe.hpp).  Between panels: the trailing SYRK update (rank nbo, lower tiles only) -- its strip over the next
//     panel's columns as a GEMM launch, the rest as fill tiles inside the next panel's launch.
//   BATCHES (throughput-bound) -- per 128-column block: potrf_diag3_kernel (one workgroup per matrix), the panel TRSM as an MFMA
//     GEMM  A[c+128:, c:c+128] <- A[...] * inv(L_cc)^T, strip updates by recursive halving, trailing SYRK with K = nbo.
//
// inv(L_cc) blocks are kept: gpk_solve.hip turns every triangular solve of the
// path into GEMM/GEMV work with them.
#include "gpk_common.hpp"
#include "gpk_gemm_tile.hpp"
#include "gpk_potrf_pipe.hpp"
#include <vector>

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
    int info_base;     // added to the reported pivot order (the matrix may be a diagonal block of a larger one)
    long long* prof;   // debug: per-phase cycle stamps of workgroup 0 (nullable)
    int zero_next;     // clear the first 4 KiB of the NEXT block's slot in `dinv`: the flag words of the panel step that follows (gpk_panel_step_launch)
    T* rhs = nullptr;          // round 6 (gpk_potrf_rhs): one right-hand side per matrix -- this block's 128 entries become inv(L_jj) b_j
    int64_t rhs_stride = 0;
};

#define PROF_MARK(i)                                                            \
    do {                                                                        \
        if (p.prof != nullptr && blockIdx.x == 0 && threadIdx.x == 0)           \
            p.prof[(p.off / GPK_DB) * 16 + (i)] = (long long)__builtin_readcyclecounter(); \
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

// ---- pieces of the in-LDS factorisation of the 128x128 block (S: [128][LDP]) ----

// tile t of an update list after micro-step s: mode 0 = micro-column s+1 (bi = s+1+t),
// mode 1 = lower triangle of the tiles with s+2 <= bj <= bi <= 7
__device__ __forceinline__ void tile_of(int mode, int s, int t, int& bi, int& bj) {
    if (mode == 0) {
        bi = s + 1 + t;
        bj = s + 1;
    } else {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= t) ++i;
        bi = s + 2 + i;
        bj = s + 2 + (t - i * (i + 1) / 2);
    }
}

// rank-16 MFMA update  S[bi][bj] -= X_bi X_bj^T  (X_b = S[16b.., c0..c0+15]) of the tiles
// start, start + stride, ... of a list; two tiles are advanced together.
template <typename T>
__device__ __forceinline__ void rank16_update(T* S, int c0, int mode, int s, int ntile, int start, int stride,
                                              int lane, int lr, int kq) {
    typedef typename Traits<T>::acc_t acc_t;
    for (int t = start; t < ntile; t += 2 * stride) {
        int bi0, bj0, bi1, bj1;
        tile_of(mode, s, t, bi0, bj0);
        const bool two = (t + stride < ntile);
        tile_of(mode, s, two ? t + stride : t, bi1, bj1);
        acc_t a0, a1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a0[q] = S[(16 * bi0 + Traits<T>::crow(lane, q)) * LDP + 16 * bj0 + lr];
            a1[q] = S[(16 * bi1 + Traits<T>::crow(lane, q)) * LDP + 16 * bj1 + lr];
        }
        T av0[4], bv0[4], av1[4], bv1[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            av0[kk] = -S[(16 * bi0 + lr) * LDP + c0 + 4 * kk + kq];
            bv0[kk] = S[(16 * bj0 + lr) * LDP + c0 + 4 * kk + kq];
            av1[kk] = -S[(16 * bi1 + lr) * LDP + c0 + 4 * kk + kq];
            bv1[kk] = S[(16 * bj1 + lr) * LDP + c0 + 4 * kk + kq];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a0 = Traits<T>::mfma(av0[kk], bv0[kk], a0);
            a1 = Traits<T>::mfma(av1[kk], bv1[kk], a1);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) S[(16 * bi0 + Traits<T>::crow(lane, q)) * LDP + 16 * bj0 + lr] = a0[q];
        if (two) {
#pragma unroll
            for (int q = 0; q < 4; ++q) S[(16 * bi1 + Traits<T>::crow(lane, q)) * LDP + 16 * bj1 + lr] = a1[q];
        }
    }
}

// ---------------------------------------------------------------------------
// potrf_diag3_kernel -- the diagonal-block kernel, built around what a lone wave per SIMD costs on gfx950 (scripts/dev/mfma_latency.hip:
// one wave issues an fp64 v_mfma 16x16x4 every ~140 cycles at best -- 46 % of the pipe -- and ~180 on a dependent chain; two waves per
// SIMD reach the pipe rate between them):
//  * 512 threads = two waves per SIMD: every MFMA phase (rank-16 updates, the three merge levels of the inverse) gets twice the
//    matrix throughput of the 256-thread kernel;
//  * no separate micro-TRSM: the 16-column micro-panel is factorised as a PANEL.  In micro_chol only lanes 0..15 of the wave held rows
//    (the other three 16-lane groups computed the same thing redundantly); here those lane groups hold the rows of the tiles BELOW the
//    diagonal tile, and the scaling + rank-1 updates of the very same loop are their triangular solve -- for free, in the shadow of
//    the broadcast chain.  One wave covers the diagonal tile + 3 tiles; up to three waves (each repeating the diagonal tile's
//    arithmetic, so that they need no communication) cover the whole micro-panel;
//  * per micro-step: (U1) all eight waves update micro-column s+1 by column s (one tile each), then the panel waves factorise
//    micro-panel s+1 WHILE the other waves apply column s to the rest of the trailing tiles (U2).
// (The 256-thread kernel of rounds 1-2 -- micro-Cholesky on one wave, separate micro-TRSM, inversion by recursive doubling after the
// factorisation: 95k cycles per fp64 block against 66.5k -- was removed in round 4; profiles/r03_experiments.md sections 1-3.)
// ---------------------------------------------------------------------------
constexpr int D3_THREADS = 512;
constexpr int D3_WAVES = D3_THREADS / 64;

// waves that factorise micro-panel s (tiles s .. 7: the diagonal tile + 7 - s tiles below it, three per wave)
__device__ __forceinline__ int panel_waves(int s) { return s == 0 ? 3 : (s <= 3 ? 2 : 1); }

// micro-panel s by wave `pw` (0 .. panel_waves(s) - 1): lanes 0..15 = rows of the diagonal tile, lane group g = 1..3 = rows of tile
// s + 3 pw + g (idle past tile 7).  Wave 0 writes the diagonal tile, the reciprocal pivots and reports a non-positive pivot.
// (Writing the finished columns to global memory from here, out of the registers they sit in, was measured: +1.3k cycles per
// micro-step on the critical wave.  A wave with nothing else to do stores them one step later: store_l_columns.)
// Round 5: the rows are READ (panel_load) in front of a workgroup barrier and factorised (panel_chol) behind it.  Every panel wave repeats
// the diagonal tile's arithmetic on its lanes 0..15 and so reads the diagonal tile's rows -- which wave 0 OVERWRITES with the factor
// when it is done.  With the loads inside this function the compiler sank those of the later columns towards their first use, several
// pivots into the loop; a panel wave that fell behind wave 0 by more than that (its SIMD shared with another kernel's waves: 8 concurrent
// streams of batched factorisations, one block in ~1e5) then read factorised values for columns 8..15 and produced a wrong factor from
// local row 80, column 24 on (scripts/dev_stream_race.py).  Nothing in a single stream ever separated the waves that far.
template <typename T>
__device__ __forceinline__ void panel_load(const T* __restrict__ S, int s, int pw, int lane, T (&a)[16]) {
    const int lr = lane & 15, g = lane >> 4;
    const int c0 = 16 * s;
    const int tile = (g == 0) ? s : s + 3 * pw + g;
    const int row = 16 * (tile < 8 ? tile : s) + lr;
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = S[row * LDP + c0 + c];
}

template <typename T>
__device__ __forceinline__ void panel_chol(T* __restrict__ S, T* __restrict__ rdiag, int s, int pw, int lane, int* info, int off, T (&a)[16]) {
    const int lr = lane & 15, g = lane >> 4;
    const int c0 = 16 * s;
    const int tile = (g == 0) ? s : s + 3 * pw + g;
    const bool live = tile < 8;
    const int row = 16 * (live ? tile : s) + lr;
    int bad = 0;
    T myr = T(0);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const T d = lane_bcast(a[j], j);
        bad = (!(d > T(0)) && bad == 0) ? j + 1 : bad;
        T ljj, rinv;
        sqrt_rsqrt(d, ljj, rinv);
        a[j] = (lane == j) ? ljj : a[j] * rinv;       // rows below the diagonal tile: this IS their solve
        myr = (lane == j) ? rinv : myr;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) {
            const T lcj = lane_bcast(a[j], c);
            a[c] -= a[j] * lcj;
        }
    }
    if (live && (g != 0 || pw == 0)) {
#pragma unroll
        for (int c = 0; c < 16; ++c) S[row * LDP + c0 + c] = a[c];
    }
    if (pw == 0) {
        if (lane < 16) rdiag[c0 + lr] = myr;
        if (bad != 0 && lane == 0) atomicCAS(info, 0, off + c0 + bad);
    }
}

// Columns 16 s .. 16 s + 15 of the finished factor (rows 16 s .. 127, lower triangle only) from LDS to global memory, by ONE wave:
// fire-and-forget stores in the shadow of a later micro-panel instead of a write-back pass of the whole block after the
// factorisation (4.4k cycles of the kernel).  Only for full, 16-byte-aligned blocks (the caller keeps the write-back pass for the
// others).  A lone wave retires an instruction every ~5 cycles and a 1 KiB store costs it ~20 (scripts/dev/store_cost.hip): the
// loop is three instructions per 8 (16) rows -- 16 / VEC lanes cover the 16 columns of a row with one 16-byte store each.
template <typename T>
__device__ __forceinline__ void store_l_columns(const T* __restrict__ S, T* __restrict__ Ag, int64_t ld, int s, int lane) {
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    constexpr int LPR = 16 / VEC, RPP = 64 / LPR;
    const int c0 = 16 * s;
    const int sub = lane / LPR, c = c0 + (lane % LPR) * VEC;
    // the diagonal tile: nothing above the diagonal
#pragma unroll
    for (int i = 0; i < 16 / RPP; ++i) {
        const int row = c0 + sub + RPP * i;
        T* __restrict__ dst = Ag + (int64_t)row * ld + c;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const T x = S[row * LDP + c + v];
            if (c + v <= row) dst[v] = x;
        }
    }
    // the rows below it: whole 16-byte vectors
    const T* src = S + (c0 + 16 + sub) * LDP + c;
    T* __restrict__ dst = Ag + (int64_t)(c0 + 16 + sub) * ld + c;
    for (int row = c0 + 16; row < GPK_DB; row += RPP) {      // (uniform trip count)
        vec_t w;
#pragma unroll
        for (int v = 0; v < VEC; ++v) w[v] = src[v];
        *reinterpret_cast<vec_t*>(dst) = w;
        src += RPP * LDP;
        dst += RPP * ld;
    }
}

// the 28 zero tiles of inv(L) above the diagonal in global memory, by one wave (a handful of instructions per tile)
template <typename T>
__device__ __forceinline__ void store_w_zero_tiles(T* __restrict__ Wg, int lane) {
    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    constexpr int LPR = 16 / VEC, RPP = 64 / LPR;
    vec_t z;
#pragma unroll
    for (int v = 0; v < VEC; ++v) z[v] = T(0);
    T* __restrict__ base = Wg + (int64_t)(lane / LPR) * GPK_DB + (lane % LPR) * VEC;
    for (int tr = 0; tr < 7; ++tr)
        for (int tc = tr + 1; tc < 8; ++tc) {
            T* __restrict__ dst = base + (int64_t)(16 * tr) * GPK_DB + 16 * tc;
#pragma unroll
            for (int i = 0; i < 16 / RPP; ++i) *reinterpret_cast<vec_t*>(dst + (int64_t)(RPP * i) * GPK_DB) = z;
        }
}

// ---- the inverse of the block, grown during its factorisation ----
// inv(L) is built ROW BLOCK by row block (16 rows): row block r only needs the rows 0..r of L, which are final as soon as micro-panel
// r has been factorised, and the row blocks of inv(L) above it:
//     W_rr = inv(L_rr)                                     one 16-lane group, substitution (the reciprocal pivots are at hand)
//     W_rc = -W_rr  sum_{k=c}^{r-1} L_rk W_kc      c < r   one wave per tile: a chain of 4 (r - c) + 4 MFMAs
// so row block s-1 is computed in the shadow of the factorisation of micro-panel s+1, by waves that would otherwise wait at the
// barrier (the panel chain is one wave; the rank-16 updates of the others take a fraction of its time).  When the factorisation
// ends, row blocks 6 and 7 are left; they are computed TOGETHER: the sums over k <= 5 of both rows at once (thirteen jobs dealt to
// the eight waves, six MFMA groups each), then row 7 only adds its k = 6 term -- ~6k cycles instead of the ~21k of inverting the
// finished block by recursive doubling (three levels of two dependent products each, eight barriers).
// Storage: the strictly-lower tiles W_rc go where S has nothing -- ABOVE the diagonal, at tile position (c, r), untransposed inside
// the tile -- and the diagonal tiles W_rr into D16 (pitch DP16: conflict-free as MFMA operand).  Every tile also goes straight to
// global memory from the wave that computed it (Wg, 128 x 128 row-major; the zero tiles above the diagonal are written up front): no
// write-back pass.  The long chains are given to waves that do not share a SIMD with the panel wave (wave w runs on SIMD w % 4).
constexpr int DP16 = 18;
constexpr int D3_LDS_ELEMS = GPK_DB * LDP + GPK_DB + 8 * 16 * DP16;

// inv(L)[r][c] from the LDS image (the pipelined panel's chain reads its operand this way)
template <typename T>
__device__ __forceinline__ T inverse_at(const T* __restrict__ S, int r, int c) {
    const int tr = r >> 4, tc = c >> 4;
    const int lower = (16 * tc + (r & 15)) * LDP + 16 * tr + (c & 15);
    const int diag = GPK_DB * LDP + GPK_DB + r * DP16 + (c & 15);       // D16 sits behind the block and the reciprocal pivots
    const T v = S[tr > tc ? lower : diag];
    return tr >= tc ? v : T(0);
}

// W_qq = inv(L_qq) by lanes 0..15 of the calling wave: lane lr solves for column lr of the inverse, right-looking (x_i is final once
// the columns before it have been applied; column i + 1 of L is requested while column i is applied)
template <typename T>
__device__ __forceinline__ void inverse_diag_tile_lanes(const T* __restrict__ S, const T* __restrict__ rdiag, T* __restrict__ D16, int q, int lane) {
    const int c0 = 16 * q, lr = lane;
    T x[16], v[16], rd[16], col[2][16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        v[i] = (i == lr) ? T(1) : T(0);
        rd[i] = rdiag[c0 + i];
    }
#pragma unroll
    for (int r = 1; r < 16; ++r) col[0][r] = S[(c0 + r) * LDP + c0];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i + 1 < 16) {
#pragma unroll
            for (int r = i + 2; r < 16; ++r) col[(i + 1) & 1][r] = S[(c0 + r) * LDP + c0 + i + 1];
        }
        x[i] = v[i] * rd[i];
#pragma unroll
        for (int r = i + 1; r < 16; ++r) v[r] -= col[i & 1][r] * x[i];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) D16[(c0 + i) * DP16 + lr] = x[i];     // (zeros above the diagonal)
}

// a finished diagonal tile of the inverse from D16 to global memory, by the whole wave (4 elements per lane)
template <typename T>
__device__ __forceinline__ void store_w_diag_tile(const T* __restrict__ D16, T* __restrict__ Wg, int q, int lane) {
    const int r = 16 * q + (lane >> 2), c = (lane & 3) * 4;
    T w[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) w[v] = D16[r * DP16 + c + v];
    T* __restrict__ dst = Wg + (int64_t)r * GPK_DB + 16 * q + c;
#pragma unroll
    for (int v = 0; v < 4; ++v) dst[v] = w[v];
}

template <typename T>
__device__ __forceinline__ void inverse_diag_tile(const T* __restrict__ S, const T* __restrict__ rdiag, T* __restrict__ D16,
                                                  T* __restrict__ Wg, int q, int lane) {
    if (lane < 16) inverse_diag_tile_lanes<T>(S, rdiag, D16, q, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    store_w_diag_tile<T>(D16, Wg, q, lane);
}

// sum_{k=c}^{kend-1} L_rk W_kc  (c < kend <= r) by the calling wave, as an accumulator tile
template <typename T>
__device__ __forceinline__ typename Traits<T>::acc_t inverse_partial(const T* __restrict__ S, const T* __restrict__ D16, int r, int c,
                                                                     int kend, int lr, int kq) {
    typedef typename Traits<T>::acc_t acc_t;
    acc_t a0, a1;
    a0[0] = a0[1] = a0[2] = a0[3] = T(0);
    a1 = a0;
    // the operands of step k + 1 are requested before the MFMAs of step k are issued (a lone wave issues an MFMA every ~140 cycles:
    // the LDS latency of a step fits behind the four of the step before)
    T av[2][4], bv[2][4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        av[0][kk] = S[(16 * r + lr) * LDP + 16 * c + 4 * kk + kq];
        bv[0][kk] = D16[(16 * c + 4 * kk + kq) * DP16 + lr];
    }
    for (int k = c; k < kend; k += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (k + h < kend) {
                if (k + h + 1 < kend) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        av[1 - h][kk] = S[(16 * r + lr) * LDP + 16 * (k + h + 1) + 4 * kk + kq];
                        bv[1 - h][kk] = S[(16 * c + 4 * kk + kq) * LDP + 16 * (k + h + 1) + lr];
                    }
                }
                a0 = Traits<T>::mfma(av[h][0], bv[h][0], a0);
                a1 = Traits<T>::mfma(av[h][1], bv[h][1], a1);
                a0 = Traits<T>::mfma(av[h][2], bv[h][2], a0);
                a1 = Traits<T>::mfma(av[h][3], bv[h][3], a1);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) a0[i] += a1[i];
    return a0;
}

// sum += L_rk W_kc for ONE k (c <= k < r)
template <typename T>
__device__ __forceinline__ typename Traits<T>::acc_t inverse_term(const T* __restrict__ S, const T* __restrict__ D16, int r, int c, int k,
                                                                  typename Traits<T>::acc_t sum, int lr, int kq) {
    T av[4], bv[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        av[kk] = S[(16 * r + lr) * LDP + 16 * k + 4 * kk + kq];
        bv[kk] = (k == c) ? D16[(16 * c + 4 * kk + kq) * DP16 + lr] : S[(16 * c + 4 * kk + kq) * LDP + 16 * k + lr];
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) sum = Traits<T>::mfma(av[kk], bv[kk], sum);
    return sum;
}

// W_rc = -W_rr sum, to LDS and to global memory.  The product takes the accumulator AS the B operand: register i of lane group g
// holds row crow(lane, i) of the sum, so MFMA step i contracts over k = crow(lane, i) with A = -W_rr[lr][k] -- no layout change.
template <typename T>
__device__ __forceinline__ void inverse_finish(T* __restrict__ S, const T* __restrict__ D16, T* __restrict__ Wg, int r, int c,
                                               typename Traits<T>::acc_t sum, int lane, int lr) {
    typedef typename Traits<T>::acc_t acc_t;
    acc_t out;
    out[0] = out[1] = out[2] = out[3] = T(0);
    T wr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wr[i] = -D16[(16 * r + lr) * DP16 + Traits<T>::crow(lane, i)];
#pragma unroll
    for (int i = 0; i < 4; ++i) out = Traits<T>::mfma(wr[i], sum[i], out);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int a = Traits<T>::crow(lane, i);
        S[(16 * c + a) * LDP + 16 * r + lr] = out[i];
        Wg[(int64_t)(16 * r + a) * GPK_DB + 16 * c + lr] = out[i];
    }
}

// The calling wave's share of: the tiles of the row blocks row_lo .. row_hi of the inverse (none if row_hi < 1) and the diagonal tiles
// diag_lo .. diag_hi (none if diag_hi < diag_lo).  Column c of the rows is ONE job (its tiles depend on each other downwards), the
// jobs go in order of length -- column 0 first, the diagonal tiles last -- to the waves 7, 3, 6, 2, 5, 1, 4, 0.
template <typename T>
__device__ __forceinline__ void inverse_jobs(T* __restrict__ S, const T* __restrict__ rdiag, T* __restrict__ D16, T* __restrict__ Wg,
                                             int row_lo, int row_hi, int diag_lo, int diag_hi, int wave, int lane, int lr, int kq) {
    const int slot = (wave & 3) == 3 ? (wave == 7 ? 0 : 1) : ((wave & 3) == 2 ? (wave == 6 ? 2 : 3) : ((wave & 3) == 1 ? (wave == 5 ? 4 : 5) : (wave == 4 ? 6 : 7)));
    const int ncol = row_hi > 0 ? row_hi : 0;
    if (slot < ncol) {
        for (int r = (row_lo > slot + 1 ? row_lo : slot + 1); r <= row_hi; ++r) {
            inverse_finish<T>(S, D16, Wg, r, slot, inverse_partial<T>(S, D16, r, slot, r, lr, kq), lane, lr);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the next row reads this tile back
        }
    } else if (diag_lo + (slot - ncol) <= diag_hi) {
        inverse_diag_tile<T>(S, rdiag, D16, Wg, diag_lo + (slot - ncol), lane);
    }
}

// Row blocks 6 and 7 together, all eight waves.  Part A: P6_c = sum_{k=c}^{5} L_6k W_kc and P7_c likewise (c = 0..5; 6 - c MFMA
// groups each), dealt so that every wave has six groups; row 6 is finished at once (W_66 is there), P7 stays in registers.  Barrier.
// Part B: the holder of P7_c adds L_76 W_6c and finishes W_7c; wave 0 computes W_76.
//   wave:   0      1      2           3           4           5           6           7
//   A:     P6_0   P7_0   P6_1 P7_5   P7_1 P6_5   P6_2 P7_4   P7_2 P6_4   P6_3 P7_3   W_77
//   B:     W_76   W_70   W_75        W_71        W_74        W_72        W_73        -
template <typename T>
__device__ __forceinline__ void inverse_last_rows(T* __restrict__ S, const T* __restrict__ rdiag, T* __restrict__ D16, T* __restrict__ Wg,
                                                  int wave, int lane, int lr, int kq) {
    typedef typename Traits<T>::acc_t acc_t;
    const int c6 = wave == 0 ? 0 : (wave == 2 ? 1 : (wave == 3 ? 5 : (wave == 4 ? 2 : (wave == 5 ? 4 : (wave == 6 ? 3 : -1)))));
    const int c7 = wave == 1 ? 0 : (wave == 2 ? 5 : (wave == 3 ? 1 : (wave == 4 ? 4 : (wave == 5 ? 2 : (wave == 6 ? 3 : -1)))));
    acc_t p7;
    p7[0] = p7[1] = p7[2] = p7[3] = T(0);
    if (wave == 7) inverse_diag_tile<T>(S, rdiag, D16, Wg, 7, lane);
    if (c6 >= 0) inverse_finish<T>(S, D16, Wg, 6, c6, inverse_partial<T>(S, D16, 6, c6, 6, lr, kq), lane, lr);
    if (c7 >= 0) p7 = inverse_partial<T>(S, D16, 7, c7, 6, lr, kq);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const int cb = wave == 0 ? 6 : c7;
    if (cb >= 0) inverse_finish<T>(S, D16, Wg, 7, cb, inverse_term<T>(S, D16, 7, cb, 6, p7, lr, kq), lane, lr);
}

// One 128x128 diagonal block by the calling workgroup (D3_THREADS threads): A = the block's first element (leading dimension ld),
// nv = its valid order (identity-padded to 128), S / rdiag = D3_LDS_ELEMS elements of LDS (block, reciprocal pivots, diagonal tiles of the inverse).  LOAD = false: S already
// holds the block (lower triangle, zeros above the diagonal, identity padding) -- the pipelined panel factorisation builds the
// next block there.  W (nullable) receives inv(L); `zero_next` clears the first 4 KiB behind it.  prof: this thread's stamp
// slots (nullptr except for one thread of a profiled workgroup).  On return every thread's global stores have been ISSUED.
template <typename T, bool LOAD>
__device__ __forceinline__ void diag3_block(T* __restrict__ S, T* __restrict__ rdiag, T* __restrict__ A, int64_t ld, int nv,
                                            T* __restrict__ W, int* info, int info_off, int zero_next, long long* prof) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));        // (called from a loop: nothing derived from the thread index is worth keeping across iterations -- hoisted, it spills)
    T* __restrict__ D16 = rdiag + GPK_DB;
    const int lane_fixed = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    typedef typename Traits<T>::vec_t vec_t;
    constexpr int VEC = Traits<T>::VEC;
    constexpr int CPR = GPK_DB / VEC;                    // 16-byte chunks per row
    constexpr int PER = GPK_DB * CPR / D3_THREADS;       // chunks per thread
    const bool vec_io = (nv == GPK_DB) && ((uintptr_t)A % 16 == 0) && (ld % VEC == 0);
    if (prof) prof[0] = (long long)__builtin_readcyclecounter();

    // ---- phase 0: load the lower triangle; pad with identity; zeros above the diagonal ----
    if (!LOAD) {
    } else if (vec_io) {
        // all global loads of a thread are in flight before its first LDS store (one latency, not one per chunk)
        vec_t buf[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int id = tid + D3_THREADS * i;
            const int r = id / CPR, c = (id % CPR) * VEC;
            const int cc = (c <= r) ? c : 0;             // (above the diagonal: a harmless in-bounds address, the value is dropped below)
            buf[i] = *reinterpret_cast<const vec_t*>(A + (int64_t)r * ld + cc);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int id = tid + D3_THREADS * i;
            const int r = id / CPR, c = (id % CPR) * VEC;
#pragma unroll
            for (int v = 0; v < VEC; ++v) S[r * LDP + c + v] = (c + v <= r) ? buf[i][v] : T(0);
        }
    } else {
#pragma unroll 1
        for (int base = 0; base < GPK_DB * GPK_DB; base += D3_THREADS * 8) {
            T buf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = base + tid + D3_THREADS * i;
                const int r = idx >> 7, c = idx & 127;
                buf[i] = (r < nv && c <= r) ? A[(int64_t)r * ld + c] : ((r == c) ? T(1) : T(0));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = base + tid + D3_THREADS * i;
                S[(idx >> 7) * LDP + (idx & 127)] = buf[i];
            }
        }
    }
    if (LOAD) __syncthreads();
    if (prof) prof[1] = (long long)__builtin_readcyclecounter();

    // ---- phase 1: factorise; every finished micro-panel leaves for global memory at once; the inverse grows row block by row
    //      block in the shadow of the panel factorisations ----
    {
        const int np = panel_waves(0);
        T pa[16];
        if (wave < np) panel_load<T>(S, 0, wave, lane_fixed, pa);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // every panel wave holds its rows: wave 0 may overwrite the diagonal tile (see panel_load)
        if (wave < np) panel_chol<T>(S, rdiag, 0, wave, lane_fixed, info, info_off, pa);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (LDS only, here and below: global stores are in flight and nobody waits for them)
    if (prof) prof[2] = (long long)__builtin_readcyclecounter();
    for (int s = 0; s < 7; ++s) {
        const int c0 = 16 * s;
        // (the per-lane LDS offsets and global addresses of the jobs below are loop-invariant expressions of the lane index: hoisted
        // out of this loop they spill -- 376 bytes of scratch per lane and a scratch reload + vmcnt(0) before every store, measured)
        int lane = lane_fixed;
        asm volatile("" : "+v"(lane));
        const int lr = lane & 15, kq = lane >> 4;
        // (U1) micro-column s+1 by column s: 7 - s tiles, one per wave
        rank16_update<T>(S, c0, 0, s, 7 - s, wave, D3_WAVES, lane, lr, kq);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (prof) prof[21 + s] = (long long)__builtin_readcyclecounter();
        // panel s+1 on its waves  ||  (U2) column s applied to the remaining tiles on the others, then their share of the inverse
        const int np = panel_waves(s + 1);
        T pa[16];
        if (wave < np) panel_load<T>(S, s + 1, wave, lane, pa);
        if (np > 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (see panel_load; one panel wave: nobody else reads the diagonal tile's rows)
        if (wave < np) {
            panel_chol<T>(S, rdiag, s + 1, wave, lane, info, info_off, pa);
        } else {
            rank16_update<T>(S, c0, 1, s, (6 - s) * (7 - s) / 2, wave - np, D3_WAVES - np, lane, lr, kq);
            // Steps 0 and 1 are bound by their 21 / 15 trailing tiles, not by the panel: the inverse starts in step 2 (diagonal tiles
            // 0..2), step 3 takes row blocks 1 and 2 (+ diagonal tile 3), step s >= 4 row block s - 1 (+ diagonal tile s).  The
            // finished columns of L leave from step 3 on, through waves without a share of the inverse.
            if (W != nullptr) {
                if (s == 2) inverse_jobs<T>(S, rdiag, D16, W, 0, 0, 0, 2, wave, lane, lr, kq);
                else if (s == 3) inverse_jobs<T>(S, rdiag, D16, W, 1, 2, 3, 3, wave, lane, lr, kq);
                else if (s >= 4) inverse_jobs<T>(S, rdiag, D16, W, s - 1, s - 1, s, s, wave, lane, lr, kq);
            }
            if (vec_io) {
                int col = -1;
                if (s == 3) col = wave == 4 ? 0 : (wave == 5 ? 1 : (wave == 1 ? 2 : (wave == 2 ? 3 : -1)));
                else if (s >= 4 && wave == 4) col = s;
                if (col >= 0) store_l_columns<T>(S, A, ld, col, lane);
            }
            if (W != nullptr && s == 4 && wave == 1) {     // the zero tiles of inv(L) above the diagonal (the diagonal tiles bring their own zeros)
                store_w_zero_tiles<T>(W, lane);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (prof) prof[3 + s] = (long long)__builtin_readcyclecounter();
    }
    int lane = lane_fixed;
    asm volatile("" : "+v"(lane));
    if (!vec_io) {
        // ragged or unaligned block: the write-back pass (lower triangle only; the upper triangle is never touched)
        for (int idx = tid; idx < GPK_DB * GPK_DB; idx += D3_THREADS) {
            const int r = idx >> 7, c = idx & 127;
            if (r < nv && c <= r) A[(int64_t)r * ld + c] = S[r * LDP + c];
        }
    }
    if (W == nullptr) {
        if (vec_io && wave == 4) store_l_columns<T>(S, A, ld, 7, lane);
        return;
    }
    if (prof) prof[10] = (long long)__builtin_readcyclecounter();

    // ---- phase 2: the last two row blocks of the inverse; the last diagonal tile of L ----
    inverse_last_rows<T>(S, rdiag, D16, W, wave, lane, lane & 15, lane >> 4);
    if (vec_io && wave == 7) store_l_columns<T>(S, A, ld, 7, lane);
    if (prof) prof[12] = (long long)__builtin_readcyclecounter();
    if (zero_next && tid < 256) {
        uint4* z = reinterpret_cast<uint4*>(W + (int64_t)GPK_DB * GPK_DB);
        z[tid] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (prof) prof[13] = (long long)__builtin_readcyclecounter();
}


// WPE: waves per SIMD the kernel is compiled for.  2: one 512-thread workgroup per CU, 189 registers (fp32) -- the latency of ONE block
// counts (single matrices).  4 (fp32 only: the fp64 block fills the LDS): two workgroups per CU at 128 registers and two dozen spills --
// batches of more matrices than CUs, where a launch is rounds of blocks (512 matrices: 15.15 -> 15.06 ms per factorisation in lockstep,
// 14.98 -> 14.65 with the mixed-phase steps; profiles/r06_experiments.md).
template <typename T, int WPE = 2>
__global__ __launch_bounds__(D3_THREADS, WPE) void potrf_diag3_kernel(DiagArgs<T> p) {
    __shared__ __attribute__((aligned(16))) T S[D3_LDS_ELEMS];
    const int64_t b = blockIdx.x;
    T* A = p.A + b * p.bstride + p.off * p.ld + p.off;
    const int rem = p.n - (int)p.off;
    T* W = p.dinv == nullptr ? nullptr : p.dinv + b * p.dinv_bstride + (p.off / GPK_DB) * (int64_t)(GPK_DB * GPK_DB);
    long long* prof = (p.prof != nullptr && blockIdx.x == 0 && threadIdx.x == 0) ? p.prof + (p.off / GPK_DB) * 32 : nullptr;
    diag3_block<T, true>(S, S + GPK_DB * LDP, A, p.ld, rem < GPK_DB ? rem : GPK_DB, W, p.info + b, (int)p.off + p.info_base, p.zero_next, prof);
    if (p.rhs != nullptr && W != nullptr) {
        // the right-hand side that is solved along: b_j <- inv(L_jj) b_j (every earlier block column has been applied to b_j by the
        // solve tiles of the steps before, batch_mix_kernel) -- four threads per row of the inverse this workgroup has just stored
        gpk_barrier_stores_done();
        const int tid = threadIdx.x;
        const int nv = rem < GPK_DB ? rem : GPK_DB;
        T* yb = p.rhs + b * p.rhs_stride + p.off;
        if (tid < GPK_DB) S[tid] = tid < nv ? yb[tid] : T(0);
        __syncthreads();
        const int row = tid >> 2, q = tid & 3;
        const T* wr = W + (int64_t)row * GPK_DB + q * 32;
        T acc = T(0);
#pragma unroll 8
        for (int k = 0; k < 32; ++k) acc += wr[k] * S[q * 32 + k];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (q == 0 && row < nv) yb[row] = acc;
    }
}

// ---------------------------------------------------------------------------
// potrf_pipe_kernel -- a whole panel of the factorisation (its diagonal blocks, the solves below them, the rank-128 updates
// inside the panel) in ONE launch, as a pipeline between workgroups (task list and dependency rules: gpk_potrf_pipe.hpp).
//
// Against one diagonal-block launch + one panel-step launch per 128 columns the chain loses, per step: two kernel boundaries
// (drain, dispatch, ramp: ~3 us each between dependent launches) and everything of the step that the next diagonal block does not
// depend on -- the solves of the rows further down and their updates run on the worker workgroups WHILE the chain workgroup
// factorises; what the next diagonal block does depend on (the 128 rows below the block, and their update of it) is cut into ten
// tasks of 32 rows that as many workers run side by side, eight waves each.
//
//   workgroup 0 (the chain):      for every block j: wait for the tiles of block j (counter), factorise + invert it in LDS
//                                 (diag3_block), publish inv(L_jj).
//   workgroups 1.. (workers):     tasks from an atomic counter in list order, on the GEMM tile (eight waves; 64 x 128 per task,
//                                 32 x 128 / 32 x 64 for the critical ones); progress words / counters say when a piece is ready.
// A first version let the chain workgroup do the critical solve and update itself, operands in LDS and registers (no flags, no
// round trip of the next block through global memory): 2304 MFMAs on one CU between two factorisations, 28 us -- slower than the
// two launches it replaced (profiles/r03_experiments.md).
// Flag protocol as in panel_step_kernel: data stores -> barrier (vmcnt drained) -> agent-scope release -> flag / counter; consumers
// poll with relaxed agent-scope loads, then agent-scope acquire -> barrier.  A poll that exceeds PIPE_SPIN_LIMIT iterations
// (seconds; never seen) raises the abort word: every workgroup leaves and `info` reports -1 instead of hanging the device.
// ---------------------------------------------------------------------------
constexpr unsigned PIPE_SPIN_LIMIT = 1u << 22;

template <typename T>
struct PipeArgs {
    T* A;              // element (c0, c0)
    int64_t ld;
    int m;             // rows from c0 to the end of the matrix
    int w;             // columns of the panel
    PipeShape sh;
    T* dinv;           // slot of the panel's first diagonal block
    unsigned* ctrl;    // pipe_ctrl_words(sh) zeroed words
    int* info;
    int info_off;      // added to the pivot orders (columns left of the panel)
    int vec_ok;
    int ntasks, nseg;
    int segoff[GPK_PIPE_MAX_SEGS + 1];   // first task of segment k (gpk_potrf_pipe.hpp)
    long long* prof;   // 32 stamps per diagonal block (nullable)
    // FILL: the part of the PREVIOUS panel's rank-K update that this panel does not touch (the lower triangle right of it), as
    // 128 x 128 tiles from a second counter (ctrl[2]) for the workers the chain-bound panel leaves idle; fill_tiles == 0: none
    unsigned* wait_word;   // see PanelCtx (nullable)
    unsigned wait_value;
    GemmArgs<T> fill;
    int fill_tiles;
    int panel_wgs;     // workgroups 1 .. panel_wgs take panel tasks first and fill tiles afterwards, the others the other way round
};

// wave 0 of the workgroup: lane i < count waits for *w_i == want.  false = aborted.
__device__ __forceinline__ bool pipe_poll(unsigned* w, bool mine, unsigned want, unsigned* abort_word) {
    unsigned it = 0;
    for (;;) {
        const bool ok = !mine || __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want;
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(2);
        if ((++it & 255u) == 0) {
            if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;      // (reported by pipe_wait's caller)
            if (it > PIPE_SPIN_LIMIT) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

// all threads: wait for up to four words (w[i] == want[i], i < count), then acquire.  false = aborted (uniform).
__device__ __forceinline__ bool pipe_wait(unsigned* w0, unsigned* w1, unsigned* w2, unsigned* w3, unsigned v0, unsigned v1,
                                          unsigned v2, unsigned v3, int count, unsigned* abort_word, int* s_ctl) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        unsigned* w = tid == 0 ? w0 : (tid == 1 ? w1 : (tid == 2 ? w2 : w3));
        const unsigned v = tid == 0 ? v0 : (tid == 1 ? v1 : (tid == 2 ? v2 : v3));
        const bool mine = tid < count;
        const bool ok = pipe_poll(mine ? w : w0, mine, v, abort_word);
        if (tid == 0) {
            s_ctl[1] = ok ? 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    return s_ctl[1] != 0;
}

// all threads: this workgroup's global stores are visible agent-wide before the word changes.  counter == nullptr: *word = v.
// Otherwise *counter += 1 and, if that made it `full`, *word = v (word may be nullptr: the counter is what consumers poll).
__device__ __forceinline__ void pipe_publish(unsigned* word, unsigned v, unsigned* counter = nullptr, unsigned full = 0u) {
    gpk_barrier_stores_done();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (counter == nullptr) {
            __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned before = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (word != nullptr && before + 1u == full) __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <typename T>
__device__ __forceinline__ void pipe_chain(const PipeArgs<T>& p, T* __restrict__ S, int* s_ctl) {
    T* rdiag = S + GPK_DB * LDP;
    const int tid = threadIdx.x;
    const int npb = p.sh.npb, nd = p.sh.nd;
    unsigned* abort_word = p.ctrl + 1;
    unsigned* cnt = p.ctrl + GPK_PIPE_CTRL_HEAD + (int64_t)p.sh.R * npb;
    int64_t ld = p.ld;
    for (int j = 0; j < nd; ++j) {
        asm volatile("" : "+s"(ld));       // (per-thread offsets are loop-invariant: hoisted out of this loop they spill)
        T* Ab = p.A + (int64_t)GPK_DB * j * ld + GPK_DB * j;
        const int rem = p.m - GPK_DB * j;
        T* W = p.dinv + (int64_t)j * (GPK_DB * GPK_DB);
        long long* prof = (p.prof != nullptr && tid == 0) ? p.prof + j * 32 : nullptr;
        if (prof) prof[14] = (long long)__builtin_readcyclecounter();
        if (j > 0) {                       // every tile of the block has received the update of step j - 1
            unsigned* c = cnt + (int64_t)(2 * j) * npb + j;
            if (!pipe_wait(c, c, c, c, (unsigned)pipe_xupdates(pipe_fine_strips(p.sh, j - 1)), 0u, 0u, 0u, 1, abort_word, s_ctl)) break;
        } else if (p.wait_word != nullptr) {   // the launch that updates this panel's first block is still running beside this one
            if (!pipe_wait(p.wait_word, p.wait_word, p.wait_word, p.wait_word, p.wait_value, 0u, 0u, 0u, 1, abort_word, s_ctl)) break;
        }
        if (prof && j > 0) prof[30 - 32] = wall_clock64();          // (development stamps, 100 MHz: the critical path of step j - 1 ends here)
        diag3_block<T, true>(S, rdiag, Ab, ld, rem < GPK_DB ? rem : GPK_DB, W, p.info, p.info_off + GPK_DB * j, 0, prof);
        pipe_publish(p.ctrl + 16 + j, 1u);
        if (prof) {
            prof[15] = (long long)__builtin_readcyclecounter();
            prof[16] = wall_clock64();
        }
    }
    if (tid == 0 && __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) atomicExch(p.info, -1);
}

template <typename T, bool EDGE>
__device__ __forceinline__ void pipe_worker(const PipeArgs<T>& p, char* smem, int* s_ctl, int role) {
    const int tid = threadIdx.x;
    const int npb = p.sh.npb, R = p.sh.R;
    unsigned* abort_word = p.ctrl + 1;
    unsigned* prog = p.ctrl + GPK_PIPE_CTRL_HEAD;
    unsigned* cnt = prog + (int64_t)R * npb;
    GemmArgs<T> g;
    g.lda = p.ld; g.ldc = p.ld; g.ldcin = p.ld;
    g.sA = g.sB = g.sC = g.sA2 = g.sB2 = g.sC2 = 0;
    g.M = p.m; g.K = GPK_DB;
    g.tiles_m = R;
    g.lower_only = 0; g.tri_k = 0; g.tri_k_lo = 0; g.tri_k_lo_b = 0; g.pair_cols = 0; g.colmask = 0; g.grp_tiles = 0;
    g.split_from = INT32_MAX;
    g.colscale = nullptr; g.colss = nullptr; g.ldss = 0; g.xcd_batch = 0; g.xcd_tiles = 0;
    g.vec_ok = p.vec_ok;
    int k = 0;
    const bool fill_first = role > p.panel_wgs;
    bool fill_left = p.fill_tiles > 0, panel_left = true;
    for (;;) {
        if (fill_left && (fill_first || !panel_left)) {
            // a tile of the previous panel's trailing update: nothing in this launch depends on it or feeds it
            if (tid == 0) s_ctl[0] = (int)__hip_atomic_fetch_add(p.ctrl + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const int ft = s_ctl[0];
            __syncthreads();
            if (ft >= p.fill_tiles) {
                fill_left = false;
                continue;
            }
            int ti = 0, tj;
            const int ftn = p.fill.tiles_n, ftri = ftn * (ftn + 1) / 2;
            if (ft < ftri) {
                while ((ti + 1) * (ti + 2) / 2 <= ft) ++ti;
                tj = ft - ti * (ti + 1) / 2;
            } else {                       // the tile rows under the square part (rows under the matrix)
                ti = ftn + (ft - ftri) / ftn;
                tj = (ft - ftri) % ftn;
            }
            gemm_tile<T, 128, true, true, EDGE, 1, D3_WAVES>(p.fill, ti, tj, 0, 0, smem);
            continue;
        }
        if (!panel_left) return;
        if (tid == 0) s_ctl[0] = (int)__hip_atomic_fetch_add(p.ctrl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int t = s_ctl[0];
        __syncthreads();
        if (t >= p.ntasks) {
            panel_left = false;
            continue;
        }
        while (t >= p.segoff[k + 1]) ++k;
        const PipeTask tk = pipe_decode(p.sh, k, t - p.segoff[k]);
        const int j = tk.j;
        const T* Pj = p.A + GPK_DB * j;                        // column block j, rows from the top of the panel
        const bool solve = tk.kind == PIPE_SOLVE || tk.kind == PIPE_XSOLVE;
        if (solve) {
            g.A = Pj; g.B = p.dinv + (int64_t)j * (GPK_DB * GPK_DB); g.C = const_cast<T*>(Pj); g.Cin = Pj;
            g.ldb = GPK_DB;
            g.N = GPK_DB;
            g.alpha = T(1); g.beta_over_alpha = T(0); g.has_beta = 0;
            g.tiles_n = 1;
        } else {
            g.A = Pj; g.B = Pj; g.C = p.A; g.Cin = p.A;
            g.ldb = p.ld;
            g.N = p.w;
            g.alpha = T(-1); g.beta_over_alpha = T(-1); g.has_beta = 1;
            g.tiles_n = npb;
        }
        const int s1 = 2 * (j + 1);
        if (tk.kind == PIPE_SOLVE) {
            // the inverse of block j is out, every earlier update of the piece applied
            unsigned* own = prog + (int64_t)tk.s * npb + j;
            if (!pipe_wait(p.ctrl + 16 + j, own, own, own, 1u, (unsigned)j, 0u, 0u, 2, abort_word, s_ctl)) return;
            gemm_tile<T, GPK_PIPE_STRIP, true, true, EDGE, 2, D3_WAVES, true>(g, tk.s, 0, 0, 0, smem);
            pipe_publish(own, (unsigned)(j + 1));
        } else if (tk.kind == PIPE_UPDATE) {
            // own strip and the rows of column block cb solved, the previous update of the piece applied
            unsigned* own = prog + (int64_t)tk.s * npb + j;
            unsigned* b0 = prog + (int64_t)(2 * tk.cb) * npb + j;
            const bool two = 2 * tk.cb + 1 < R;
            unsigned* piece = prog + (int64_t)tk.s * npb + tk.cb;
            if (!pipe_wait(own, b0, piece, two ? b0 + npb : own, (unsigned)(j + 1), (unsigned)(j + 1), (unsigned)j, (unsigned)(j + 1),
                           4, abort_word, s_ctl))
                return;
            gemm_tile<T, GPK_PIPE_STRIP, true, true, EDGE, 2, D3_WAVES, false>(g, tk.s, tk.cb, 0, 0, smem);
            pipe_publish(piece, (unsigned)(j + 1));
        } else if (tk.kind == PIPE_XSOLVE) {
            // fine strip q of block j+1: as a solve; the strip of 64 rows is solved when its fine strips are
            const int s = s1 + (tk.s >> 1);
            unsigned* own = prog + (int64_t)s * npb + j;
            if (!pipe_wait(p.ctrl + 16 + j, own, own, own, 1u, (unsigned)j, 0u, 0u, 2, abort_word, s_ctl)) return;
            long long* pf = (p.prof != nullptr && tid == 0 && tk.s == 0) ? p.prof + j * 32 : nullptr;
            if (pf) pf[17] = wall_clock64();
            gemm_tile<T, GPK_PIPE_FINE, true, true, EDGE, 4, D3_WAVES, true>(g, 4 * (j + 1) + tk.s, 0, 0, 0, smem);
            if (pf) pf[18] = wall_clock64();
            pipe_publish(own, (unsigned)(j + 1), cnt + (int64_t)s * npb + j,
                         (unsigned)pipe_xsolves_in_strip(pipe_fine_strips(p.sh, j), tk.s >> 1));
            if (pf) pf[19] = wall_clock64();
        } else {
            // tile (fine strip q, column half h) of diagonal block j+1: all rows of the block solved, the earlier updates applied
            unsigned* b0 = prog + (int64_t)s1 * npb + j;
            const bool two = s1 + 1 < R;
            unsigned* piece = prog + (int64_t)(s1 + (tk.s >> 1)) * npb + j + 1;
            if (!pipe_wait(b0, two ? b0 + npb : b0, piece, piece, (unsigned)(j + 1), (unsigned)(j + 1), (unsigned)j, 0u, 3, abort_word, s_ctl)) return;
            long long* pf = (p.prof != nullptr && tid == 0 && tk.s == 3 && tk.cb == 1) ? p.prof + j * 32 : nullptr;
            if (pf) pf[20] = wall_clock64();
            gemm_tile<T, GPK_PIPE_FINE, true, true, EDGE, 2, D3_WAVES, false>(g, 4 * (j + 1) + tk.s, 2 * (j + 1) + tk.cb, 0, 0, smem);
            if (pf) pf[28] = wall_clock64();
            pipe_publish(nullptr, 0u, cnt + (int64_t)s1 * npb + j + 1, 0u);
            if (pf) pf[29] = wall_clock64();
        }
    }
}

template <typename T, bool EDGE>
__global__ __launch_bounds__(D3_THREADS, 2) void potrf_pipe_kernel(PipeArgs<T> p) {
    __shared__ __attribute__((aligned(16))) T S[D3_LDS_ELEMS];
    __shared__ int s_ctl[4];
    static_assert(sizeof(T) * D3_LDS_ELEMS >= 2 * 5 * op_bytes(GPK_PIPE_FINE) && sizeof(T) * D3_LDS_ELEMS >= 2 * 3 * op_bytes(GPK_PIPE_STRIP) &&
                      sizeof(T) * D3_LDS_ELEMS >= 2 * 2 * op_bytes(128),
                  "the worker's operand tiles live in the chain's block");
    // Roles by ARRIVAL, not by block index: the first workgroup that gets to run takes the chain, so the chain is
    // resident whenever any worker is -- whatever part of the grid the dispatcher has placed, in whatever order.
    if (threadIdx.x == 0) s_ctl[1] = (int)__hip_atomic_fetch_add(p.ctrl + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int role = s_ctl[1];
    __syncthreads();
    if (role == 0) pipe_chain<T>(p, S, s_ctl);
    else pipe_worker<T, EDGE>(p, reinterpret_cast<char*>(S), s_ctl, role);
    // whoever leaves after an abort reports it (the chain may have finished its blocks and gone before a worker gave up)
    if (threadIdx.x == 0 && __hip_atomic_load(p.ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) atomicExch(p.info, -1);
}

// ---------------------------------------------------------------------------
// batch_mix_kernel (round 6) -- the panel solves AND the update of one 128-column step of the batched factorisation in ONE launch in
// which the two phases MIX.
//
// The batched path of rounds 1-5 is batch-wide launches in lockstep: diagonal blocks (latency-bound, one workgroup per matrix), panel
// solves (half memory-bound: a 128 x 128 tile read, multiplied by the inverted block, written back), update GEMMs (MFMA-bound) -- 47
// launches per 2048-matrix, and while the chip is in a solve launch its matrix pipes idle (cfg4: the solves are 2.3 of 15.2 ms).  Here
// everything step `jn` does below its diagonal blocks is one launch of the 128-tile's own workgroups (256 threads, two per CU), ONE
// TASK EACH, in per-XCD queues: block L is task L / 8 of queue L % 8 -- the dispatcher places block L on XCD L % 8 and starts blocks in
// index order, so the queue order is the start order and matrix m = queue m % 8 keeps its panels in one L2:
//
//   T(m, r)        the panel solve of row tile r below block jn (TRIB tile, in place, against the inverse the diagonal-block launch in
//                  front of this one left in `dinv`); then cnt[m] += 1 behind a barrier that waits for the tile's stores;
//   U(m, ti, tj)   one 128 x 128 tile of the update that block column jn now owes the matrix (depth K = 128, 256, ... as the recursive
//                  halving of potrf_panel has it, or the outer panel's rank-nbo trailing update); pulls its C tile into the L2, waits for
//                  cnt[m] == all solves of m, multiplies.
//
// Queue order: slot s = [the solves of matrix s] [the update tiles of matrix s - lag]: memory-bound solves and MFMA-bound updates of
// DIFFERENT matrices share every CU at every moment.  Every dependency of a task is a block with a LOWER index, which has started
// whenever this one has and never waits for a later one: no deadlock whatever part of the grid is resident.
//
// NO FENCES between the solves and the updates of a matrix: all of them run on ITS XCD, whose L2 is the point of coherence of its
// CUs -- a solve tile's stores are in that L2 when the counter moves (write-through L1, counted after the stores were acknowledged),
// the counter is read past the L1 (agent-scope atomic load), and no CU can hold a stale L1 line of a solved row (it would have had to
// read the row earlier in this launch: only the workgroup that solves a row reads it unsolved).  Agent-scope release / acquire
// fences write back / invalidate the XCD's WHOLE L2: one release per solve tile costs 2.4 ms per 512 x 2048^2 factorisation, a
// release / acquire pair per task 3 ms (profiles/r06_experiments.md).  What this rests on -- every block of a queue on one XCD -- is
// CHECKED: the first block of a queue records its XCC_ID, every other one compares, a mismatch is reported as info = -2 (never seen;
// the host keeps streams with a CU mask on the lockstep path).
//
// Also built and measured: the same queues served by a RESIDENT grid (one workgroup per slot, tasks from atomic counters, the next
// tile's input prefetched, claims inside the tile body as in gemm_persist_kernel) -- 14.9-15.0 ms against 14.5 for this kernel and
// 15.1 for the lockstep launches: a workgroup that goes from one tile to the next must have its own stores acknowledged before it can
// consume the next tile's loads (one in-order counter per wave), a FRESH workgroup's loads return as they arrive
// (profiles/r06_batched_resident_task_loop.patch).  The diagonal blocks stay a launch of their own between two of these (512 threads
// and 189 registers: they do not fit beside the tiles; inlined into a 512-thread task loop they and the 8-wave tile spill) -- which
// also clears the control words: the first 4 KiB of the dinv slot jn + 1 of every matrix (word 0 = cnt[m]; of matrix 0's: word 16 =
// abort, words 24..31 = XCC_ID + 1 of the queues).
// ---------------------------------------------------------------------------
template <typename T>
struct BatchStepArgs {
    T* A;
    int64_t ld, bstride;
    int n, batch;
    T* dinv;
    int64_t dstride;
    int* info;
    int jn;          // the diagonal block whose rows are solved (c = 128 jn); the update starts at row / column c + 128
    int K;           // depth of the update: panel columns [c + 128 - K, c + 128)
    int tn, tm;      // 128-column tiles of the updated region; 128-row tiles below block jn (= solves per matrix)
    int nU;          // update tiles per matrix
    int lag;
    int opts;        // development (knob 56): 1 = an update tile does not pull its C tile into the L2 before it waits, 4 = solve tiles publish behind an agent-scope release
    T* rhs;          // round 6 (gpk_potrf_rhs): one right-hand side per matrix (nullable); block jn of it holds inv(L_jj) b_j (the diagonal-block kernel)
    int64_t srhs;
};

template <typename T>
__device__ __forceinline__ unsigned* batch_ctrl(const BatchStepArgs<T>& p, int m) {
    return reinterpret_cast<unsigned*>(p.dinv + (int64_t)m * p.dstride + (int64_t)(p.jn + 1) * (GPK_DB * GPK_DB));
}

template <typename T>
__global__ __launch_bounds__(256, 2) void batch_mix_kernel(BatchStepArgs<T> p) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * op_bytes(128)];
    __shared__ int s_ctl[4];
    const int tid = threadIdx.x;
    const int L = (int)blockIdx.x, qx = L & 7, q = L >> 3;
    const int nmx = (p.batch - qx + 7) >> 3;
    const int P = p.tm + p.nU;
    if (nmx <= 0 || q >= (nmx + p.lag) * P) return;
    const int c1 = (p.jn + 1) * GPK_DB;
    const int slot = q / P, r = q - slot * P;
    // Every block of a queue on ONE XCD is what the fence-free protocol rests on: each block ORs the bit of the XCD it runs on into the
    // queue's word (fire and forget: no block waits for it at its start), the solve tiles look at the word when they publish -- more than
    // one bit = info -2.  (A first version did a compare-and-swap with a returned value at the top of every block: ~2 us of exposed
    // latency per tile.)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned* const home = batch_ctrl(p, 0) + 24 + qx;
    const unsigned mybit = 1u << (xcc & 7u);
    GemmArgs<T> g;
    g.lda = g.ldb = g.ldc = g.ldcin = p.ld;
    g.sA = g.sB = g.sC = p.bstride;
    g.sA2 = g.sB2 = g.sC2 = 0;
    g.lower_only = 0; g.vec_ok = 1; g.tri_k = g.tri_k_lo = g.tri_k_lo_b = g.pair_cols = 0;
    g.colscale = nullptr; g.colss = nullptr; g.ldss = 0; g.xcd_batch = g.xcd_tiles = 0; g.colmask = 0; g.grp_tiles = 0;
    g.split_from = INT32_MAX;
    if (r < p.tm) {
        if (slot >= nmx) return;
        const int m = qx + 8 * slot;
        g.A = p.A + (int64_t)c1 * p.ld + (c1 - GPK_DB); g.C = const_cast<T*>(g.A); g.Cin = g.A;
        g.B = p.dinv + (int64_t)p.jn * (GPK_DB * GPK_DB); g.ldb = GPK_DB; g.sB = p.dstride;
        g.M = p.n - c1; g.N = GPK_DB; g.K = GPK_DB;
        g.alpha = T(1); g.beta_over_alpha = T(0); g.has_beta = 0;
        g.tiles_m = p.tm; g.tiles_n = 1;
        gemm_tile<T, 128, true, true, false, 1, 4, true>(g, r, 0, m, 0, smem);
        gpk_barrier_stores_done();            // every wave's stores of the tile are acknowledged
        if (tid == 0) {
            if (p.opts & 4) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __hip_atomic_fetch_add(batch_ctrl(p, m), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned seen = __hip_atomic_fetch_or(home, mybit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) | mybit;
            if (seen & (seen - 1u)) atomicExch(p.info + qx, -2);
        }
        if (p.rhs != nullptr) {
            // the right-hand side's share of this tile, behind the publication (nobody in this launch waits for it):
            //   b[rows of the tile] -= X w,   X = the 128 x 128 tile of L this workgroup has just stored (its own stores, acknowledged:
            // read back through the CU's cache), w = block jn of b = inv(L_jj) b_j.  Two threads per row, 64 columns each.
            typedef typename Traits<T>::vec_t vec_t;
            constexpr int VEC = Traits<T>::VEC;
            T* yb = p.rhs + (int64_t)m * p.srhs;
            T* ws = reinterpret_cast<T*>(smem);             // (the tile's LDS is free: every wave is past the barrier above)
            if (tid < GPK_DB) ws[tid] = yb[c1 - GPK_DB + tid];
            __syncthreads();
            const int row = tid >> 1, half = tid & 1;
            const T* xr = p.A + (int64_t)m * p.bstride + (int64_t)(c1 + GPK_DB * r + row) * p.ld + (c1 - GPK_DB) + half * 64;
            T acc = T(0);
#pragma unroll 4
            for (int k = 0; k < 64; k += VEC) {
                const vec_t xv = *reinterpret_cast<const vec_t*>(xr + k);
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc += xv[v] * ws[half * 64 + k + v];
            }
            acc += __shfl_xor(acc, 1, 64);
            if (half == 0) yb[c1 + GPK_DB * r + row] -= acc;
        }
        return;
    }
    const int i = slot - p.lag;
    if (i < 0 || i >= nmx) return;
    const int m = qx + 8 * i;
    const int u = r - p.tm;
    int ti = u, tj = 0;
    if (u >= p.tm) {                          // column 0 from the top, then the other columns row by row
        int rem = u - p.tm;
        ti = 1;
        for (;;) {
            const int cnt = (ti < p.tn - 1) ? ti : p.tn - 1;
            if (rem < cnt) break;
            rem -= cnt;
            ++ti;
        }
        tj = 1 + rem;
    }
    g.A = p.A + (int64_t)c1 * p.ld + (c1 - p.K); g.B = g.A;
    g.C = p.A + (int64_t)c1 * p.ld + c1; g.Cin = g.C;
    g.M = p.n - c1; g.N = p.tn * GPK_DB; g.K = p.K;
    g.alpha = T(-1); g.beta_over_alpha = T(-1); g.has_beta = 1;
    g.tiles_m = p.tm; g.tiles_n = p.tn;
    unsigned first_look = 0u;
    if (tid == 0) {
        // (the counter is asked for FIRST: loads return in order, behind the C lines it would wait for them too)
        first_look = __hip_atomic_load(batch_ctrl(p, m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_or(home, mybit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (result unused: no wait)
    }
    if (!(p.opts & 1)) {
        // the C tile does not depend on the solves: pull it into the L2 while thread 0 looks at the counter (one dword of every 128-byte line)
        constexpr int LPR = 128 * (int)sizeof(T) / 128;
        const T* ct = g.Cin + (int64_t)m * p.bstride + (int64_t)ti * GPK_DB * p.ld + (int64_t)tj * GPK_DB;
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 128 * LPR / 256; ++j) {
            const int id = tid + 256 * j;
            acc ^= *reinterpret_cast<const int*>(reinterpret_cast<const char*>(ct + (int64_t)(id / LPR) * p.ld) + (id % LPR) * 128);
        }
        asm volatile("" ::"v"(acc));
    }
    if (tid == 0) {
        unsigned* cw = batch_ctrl(p, m);
        unsigned* abort_word = batch_ctrl(p, 0) + 16;
        unsigned it = 0;
        int ok = 1;
        while (first_look != (unsigned)p.tm && __hip_atomic_load(cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)p.tm) {
            __builtin_amdgcn_s_sleep(2);
            if ((++it & 255u) == 0 && (it > PIPE_SPIN_LIMIT || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        s_ctl[1] = ok;
    }
    __syncthreads();
    if (s_ctl[1] == 0) {
        if (tid == 0) atomicExch(p.info + m, -1);
        return;
    }
    gemm_tile<T, 128, true, true, false, 1, 4, false, 1>(g, ti, tj, m, 0, smem);
}

long long* g_diag_prof = nullptr;   // development aid, set through gpk_tune_diag_prof

GPK_KNOB(int, g_diag_two_per_cu, 1);         // tuning knob (gpk_tune(57, v)): fp32 batches of more matrices than CUs take the 128-register diagonal-block kernel
template <typename T>
void launch_diag(const DiagArgs<T>& d, unsigned batch, hipStream_t stream) {
    if constexpr (sizeof(T) == 4) {
        if (g_diag_two_per_cu && batch > 256u) {
            hipLaunchKernelGGL((potrf_diag3_kernel<T, 4>), dim3(batch), dim3(D3_THREADS), 0, stream, d);
            return;
        }
    }
    hipLaunchKernelGGL((potrf_diag3_kernel<T, 2>), dim3(batch), dim3(D3_THREADS), 0, stream, d);
}

template <typename T>
struct PanelCtx {
    T* A;
    int64_t n, ld, batch, bstride;
    T* dinv;
    int64_t dstride;
    int* info;
    hipStream_t stream;
    int info_base;
    int max_wgs = 0;   // workgroups a persistent launch on `stream` may count on at once (0: every CU of the device; a CU-masked stream: its CUs)
    unsigned* wait_word = nullptr;   // the first diagonal block of the next panel is ready when *wait_word == wait_value (another
    unsigned wait_value = 0;         // kernel, on another stream, is still writing it when the panel's launch starts); nullptr: it is ready
    // Round 5: rows > n -- the buffer holds `rows - n` more rows under the square matrix (K(x*, x) of the posterior): they are carried
    // through the factorisation like the rows below a diagonal block (panel solves + trailing updates) and come out as
    // K(x*, x) L^{-T}.  One matrix, pipelined panels only; `dinv` then needs one slot more than ceil(n / 128).  0: rows = n.
    int64_t rows = 0;
    int64_t nrows() const { return rows > n ? rows : n; }
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
        d.dinv = x.dinv; d.dinv_bstride = x.dstride; d.info = x.info; d.info_base = x.info_base;
        d.prof = g_diag_prof;
        d.zero_next = 0;
        launch_diag<T>(d, (unsigned)x.batch, x.stream);
        GPK_CHECK_LAUNCH();
        const int64_t r1 = c0 + GPK_DB;   // first row below the diagonal block
        if (r1 >= x.n) return GPK_OK;
        // panel TRSM as a GEMM: A[r1:, c0:c0+128] <- A[r1:, c0:c0+128] * inv(L_cc)^T
        T* P = x.A + r1 * x.ld + c0;
        const T* Wc = x.dinv + (c0 / GPK_DB) * (int64_t)(GPK_DB * GPK_DB);
        return gpk_gemm_launch<T>(true, true, x.n - r1, GPK_DB, GPK_DB, T(1), P, x.ld, x.bstride, Wc, GPK_DB,
                                  x.dstride, T(0), P, x.ld, x.bstride, x.batch, 16, x.stream);     // (16: B is lower triangular from column 0)
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

// The same panel, columns [c0, c0 + w), for ONE matrix with the fused step kernel: per 128 columns the diagonal-block kernel and
// ONE launch that solves the rows below it and applies the rank-128 update to the rest of the panel (gpk_panel_step_launch) --
// two dependent launches per step instead of three, finer strips, no recursion (the rank-128 updates stay inside the panel: at
// most w - 128 columns wide; everything right of the panel waits for the caller's rank-w update as before).  The flag words of
// step c live in the slot of diagonal block c / 128 + 1 of `dinv`, which nothing else touches until that block is factorised;
// the diagonal-block kernel of step c clears them.
template <typename T>
int potrf_panel_fused(const PanelCtx<T>& x, int64_t c0, int64_t w) {
    const int64_t ke = (c0 + w < x.n) ? c0 + w : x.n;
    for (int64_t c = c0; c < ke; c += GPK_DB) {
        const bool below = c + GPK_DB < x.n;
        DiagArgs<T> d;
        d.A = x.A; d.ld = x.ld; d.bstride = x.bstride; d.off = c; d.n = (int)x.n;
        d.dinv = x.dinv; d.dinv_bstride = x.dstride; d.info = x.info; d.info_base = x.info_base;
        d.prof = g_diag_prof;
        d.zero_next = below ? 1 : 0;
        launch_diag<T>(d, 1u, x.stream);
        GPK_CHECK_LAUNCH();
        if (!below) break;
        T* Wc = x.dinv + (c / GPK_DB) * (int64_t)(GPK_DB * GPK_DB);
        const int st = gpk_panel_step_launch<T>(x.A, x.n, x.ld, c, Wc, ke, reinterpret_cast<unsigned*>(Wc + (int64_t)GPK_DB * GPK_DB), x.stream);
        if (st) return st;
    }
    return GPK_OK;
}
GPK_KNOB(int, g_fused_step, 1);              // tuning knob (gpk_tune(32, v)): single matrices take potrf_panel_fused

GPK_KNOB(int, g_pipe, 1);                    // tuning knob (gpk_tune(37, v)): single matrices take potrf_panel_pipe (one launch per panel) where it applies
int g_pipe_cus = 0;                // CUs of the current device (queried once)
GPK_KNOB(int64_t, g_plain_nbo, 1024);        // tuning knob (gpk_tune(52, v)): panel width of the pipelined plain path for single matrices above 4096
GPK_KNOB(int, g_pipe_fill, 1);               // tuning knob (gpk_tune(38, v)): the trailing update right of the NEXT panel rides along in that panel's launch
GPK_KNOB(int, g_pipe_panel_wgs, 0);          // tuning knob (gpk_tune(39, v)): workgroups that take panel tasks first when a launch carries fill tiles (0: a third of the CUs)

// The same panel in ONE launch (potrf_pipe_kernel) -- plus a memset of its control words and, when the panel reaches the last row of
// the matrix, the diagonal-block kernel for the last block (the control words live in the `dinv` slot of the first diagonal block
// the kernel does NOT factorise: the block behind the panel, or that last block).  GPK_OK + *done = false: the shape does not fit
// (more than GPK_PIPE_MAX_BLOCKS blocks, control words beyond one slot), nothing was enqueued.
// fill_k > 0: the columns [c0 - fill_k, c0) are a finished panel whose rank-fill_k update has been applied up to column c0 + w only;
// the rest of it -- the lower triangle from row / column c0 + w on -- rides along in this launch (PipeArgs::fill).
template <typename T>
int potrf_panel_pipe(const PanelCtx<T>& x, int64_t c0, int64_t w, bool* done, int64_t fill_k = 0) {
    *done = false;
    const int64_t ke = (c0 + w < x.n) ? c0 + w : x.n;
    const int64_t m = x.nrows() - c0;
    const bool last = (ke == x.n) && x.nrows() == x.n;      // (rows under the matrix: its last diagonal block has rows to solve too)
    PipeShape sh;
    sh.npb = (int)gpk_cdiv(ke - c0, GPK_DB);
    sh.nd = last ? sh.npb - 1 : sh.npb;
    sh.R = (int)gpk_cdiv(m, GPK_PIPE_STRIP);
    sh.R32 = (int)gpk_cdiv(m, GPK_PIPE_FINE);
    if (sh.nd < 1 || sh.npb > GPK_PIPE_MAX_BLOCKS) return GPK_OK;
    const int64_t words = pipe_ctrl_words(sh);
    if (words * (int64_t)sizeof(unsigned) > (int64_t)GPK_DB * GPK_DB * (int64_t)sizeof(T)) return GPK_OK;
    PipeArgs<T> pa;
    pa.A = x.A + c0 * x.ld + c0;
    pa.ld = x.ld;
    pa.m = (int)m;
    pa.w = (int)(ke - c0);
    pa.sh = sh;
    pa.dinv = x.dinv + (c0 / GPK_DB) * (int64_t)(GPK_DB * GPK_DB);
    pa.ctrl = reinterpret_cast<unsigned*>(pa.dinv + (int64_t)sh.nd * (GPK_DB * GPK_DB));
    pa.info = x.info;
    pa.info_off = (int)c0 + x.info_base;
    constexpr int VEC = Traits<T>::VEC;
    const bool aligned = ((uintptr_t)pa.A % 16 == 0) && ((uintptr_t)pa.dinv % 16 == 0) && (x.ld % VEC == 0);
    pa.vec_ok = aligned ? 1 : 0;
    pa.nseg = pipe_num_segments(sh);
    pa.segoff[0] = 0;
    for (int k = 0; k < pa.nseg; ++k) pa.segoff[k + 1] = pa.segoff[k] + pipe_segment_tasks(sh, k);
    for (int k = pa.nseg + 1; k <= GPK_PIPE_MAX_SEGS; ++k) pa.segoff[k] = pa.segoff[pa.nseg];
    pa.ntasks = pa.segoff[pa.nseg];
    pa.prof = g_diag_prof != nullptr ? g_diag_prof + (c0 / GPK_DB) * 32 : nullptr;
    int cus = x.max_wgs;
    if (cus <= 0) {
        if (g_pipe_cus == 0) {
            int dev = 0, v = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 2)
                return GPK_ERR_LAUNCH;
            g_pipe_cus = v;
        }
        cus = g_pipe_cus;
    }
    // one workgroup per CU (the chain's block fills the LDS); never more workers than the first, largest step has tasks (+ the
    // critical tasks of the next step, which are taken early and wait)
    int64_t workers = (int64_t)(pa.nseg > 4 ? pa.segoff[5] : pa.ntasks) + 4;
    if (workers > cus - 1) workers = cus - 1;
    if (workers < 1) workers = 1;
    if (pa.ntasks == 0) workers = 0;
    pa.wait_word = x.wait_word;
    pa.wait_value = x.wait_value;
    pa.fill_tiles = 0;
    pa.panel_wgs = (int)workers;
    bool fill_edge = false;
    if (fill_k > 0 && ke < x.n) {
        const int64_t mf = x.n - ke, mr = x.nrows() - ke;        // columns / rows of the trailing matrix (mr > mf: rows under the matrix)
        GemmArgs<T>& g = pa.fill;
        const T* P = x.A + ke * x.ld + (c0 - fill_k);
        g.A = P; g.B = P; g.C = x.A + ke * x.ld + ke; g.Cin = g.C;
        g.lda = x.ld; g.ldb = x.ld; g.ldc = x.ld; g.ldcin = x.ld;
        g.sA = g.sB = g.sC = g.sA2 = g.sB2 = g.sC2 = 0;
        g.M = (int)mr; g.N = (int)mf; g.K = (int)fill_k;
        g.alpha = T(-1); g.beta_over_alpha = T(-1); g.has_beta = 1;
        g.tiles_m = (int)gpk_cdiv(mr, 128); g.tiles_n = (int)gpk_cdiv(mf, 128);
        g.lower_only = 1; g.tri_k = 0; g.tri_k_lo = 0; g.tri_k_lo_b = 0; g.pair_cols = 0; g.colmask = 0; g.grp_tiles = 0;
        g.split_from = INT32_MAX;
        g.colscale = nullptr; g.colss = nullptr; g.ldss = 0; g.xcd_batch = 0; g.xcd_tiles = 0;
        g.vec_ok = aligned ? 1 : 0;
        fill_edge = !aligned || (mf % 128) || (mr % 128) || (fill_k % Traits<T>::BK) || g.lda >= GPK_PIPE_LD_MAX;
        pa.fill_tiles = g.tiles_n * (g.tiles_n + 1) / 2 + (g.tiles_m - g.tiles_n) * g.tiles_n;     // lower triangle + the full tile rows under it
        // the panel is chain-bound: a third of the chip keeps its task list moving, the rest starts with the fill tiles
        workers = cus - 1;
        pa.panel_wgs = g_pipe_panel_wgs > 0 ? g_pipe_panel_wgs : (cus - 1) / 3;
        if (pa.panel_wgs > workers) pa.panel_wgs = (int)workers;
    }
    if (hipMemsetAsync(pa.ctrl, 0, (size_t)words * sizeof(unsigned), x.stream) != hipSuccess) return GPK_ERR_LAUNCH;
    const bool edge = !aligned || (m % GPK_PIPE_STRIP) || ((ke - c0) % GPK_DB) || fill_edge;
    if (edge) hipLaunchKernelGGL((potrf_pipe_kernel<T, true>), dim3((unsigned)(1 + workers)), dim3(D3_THREADS), 0, x.stream, pa);
    else hipLaunchKernelGGL((potrf_pipe_kernel<T, false>), dim3((unsigned)(1 + workers)), dim3(D3_THREADS), 0, x.stream, pa);
    GPK_CHECK_LAUNCH();
    if (last) {
        DiagArgs<T> d;
        d.A = x.A; d.ld = x.ld; d.bstride = x.bstride; d.off = c0 + (int64_t)sh.nd * GPK_DB; d.n = (int)x.n;
        d.dinv = x.dinv; d.dinv_bstride = x.dstride; d.info = x.info; d.info_base = x.info_base;
        d.prof = g_diag_prof;
        d.zero_next = 0;
        launch_diag<T>(d, 1u, x.stream);
        GPK_CHECK_LAUNCH();
    }
    *done = true;
    return GPK_OK;
}

// one wave that returns when *word == value (bounded like the polls of the pipelined panel): puts a stream behind a device-side flag
__global__ void wait_word_kernel(unsigned* word, unsigned value) {
    if (threadIdx.x == 0) {
        unsigned it = 0;
        while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != value && ++it < PIPE_SPIN_LIMIT) __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

template <typename T>
int potrf_panel_any(const PanelCtx<T>& x, int64_t c0, int64_t w) {
    if (g_pipe && x.batch == 1 && x.dinv != nullptr && (x.n - c0 > GPK_DB || x.nrows() > x.n)) {
        bool done = false;
        const int st = potrf_panel_pipe<T>(x, c0, w, &done);
        if (st || done) return st;
    }
    if (x.nrows() > x.n) return GPK_ERR_ARG(2);      // rows under the matrix: the pipelined panel or nothing
    if (x.wait_word != nullptr) {      // the paths below start from the diagonal block right away
        hipLaunchKernelGGL(wait_word_kernel, dim3(1), dim3(64), 0, x.stream, x.wait_word, x.wait_value);
        GPK_CHECK_LAUNCH();
    }
    // the flag words of a step (one per strip of 32 rows below it, 1024 at most) need the dinv slot of the next block: n < 32768
    if (g_fused_step && x.batch == 1 && x.dinv != nullptr && x.n - c0 <= 32768) return potrf_panel_fused<T>(x, c0, w);
    return potrf_panel<T>(x, c0, w);
}

}  // namespace

// Development aid (gpk_tune_diag_prof in include/gpk.h): device buffer of 16 int64 per diagonal
// block that receives cycle-counter stamps of the diag kernel's phases.
void gpk_set_diag_prof(long long* dev_buf) { g_diag_prof = dev_buf; }

GPK_KNOB(int, g_batch_mixed, 1);             // tuning knob (gpk_tune(53, v)): batches of aligned matrices take the mixed-phase steps (1: fp32 only -- the fp64 kernel, two tile bodies at 256 registers, spills; 2: fp64 too)
GPK_KNOB(int64_t, g_batch_mixed_min, 64);    // tuning knob (gpk_tune(54, v)): ... from this many matrices on
GPK_KNOB(int, g_batch_left, 0);              // tuning knob (gpk_tune(60, v)): the mixed-phase steps update left-looking inside an outer panel (see potrf_batched_mixed)
GPK_KNOB(int, g_batch_opts, 0);              // tuning knob (gpk_tune(56, v)): development switches of batch_mix_kernel (BatchStepArgs::opts)
GPK_KNOB(int, g_batch_lag, 96);              // tuning knob (gpk_tune(55, v)): tasks between a matrix's solves and its update tiles, at least (32 / 96 / 256 / 512: 14.70 / 14.51 / 14.60 / 14.69 ms)

// A stream created with a CU mask (hipExtStreamCreateWithCUMask) may not see every XCD, and the mixed-phase steps rest on block L
// running on XCD L % 8: such streams keep the lockstep launches.  (No mask set / the query fails: every CU.)
static bool stream_has_all_cus(hipStream_t stream) {
    if (stream == nullptr) return true;                 // (the legacy default stream carries no mask)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1 ||
        cus > 32 * 32)
        return false;
    // (the runtime fills ceil(CUs / 32) words and refuses a shorter array; an unmasked stream reports all ones: measured on MI355X,
    // scripts/dev_probe_cu_mask.py)
    const uint32_t words = (uint32_t)((cus + 31) / 32);
    uint32_t mask[32];
    for (uint32_t i = 0; i < 32; ++i) mask[i] = 0u;
    if (hipExtStreamGetCUMask(stream, words, mask) != hipSuccess) {
        (void)hipGetLastError();
        return false;                                    // nothing vouches for the placement: the lockstep launches
    }
    int set = 0;
    for (uint32_t i = 0; i < words; ++i) set += __builtin_popcount(mask[i]);
    return set >= cus;
}

// One right-hand side per matrix that is solved ALONG with a batched factorisation (gpk_potrf_rhs), inside the mixed-phase steps: the
// diagonal-block kernel of step jn turns block jn of b into inv(L_jj) b_j (it has just stored that inverse), and every solve tile of the
// step's launch, behind its publication, subtracts its 128 x 128 tile of L times that block from its own 128 entries of b -- the tile
// is the workgroup's own store, read back through the cache: the separate sweep's 4.3 GB of HBM reads (0.75 ms for 512 x 2048^2 fp32)
// become ~2 us of a solve tile's slot.  (First form: the sweep's steps as batched matrix-vector launches on a side stream beside the
// launches of step jn + 1 -- 17.6 ms against 16.65 with the sweep behind the factorisation, profiles/r06_ab_batched_rhs_side_stream.json.)
template <typename T>
struct RhsRide {
    T* B;
    int64_t sB;
    T* tmp;          // batch * 128 (+ GPK_TRSV_CTRL_ELEMS) elements
    bool done;       // set by the path that took it along
};
// The batched factorisation with the solves and the update of every 128-column step in one mixed-phase launch (batch_mix_kernel),
// the diagonal blocks in a launch of their own in front of it.  Same arithmetic, same order per entry as the lockstep path below:
// recursive halving inside outer panels of `nbo` columns, rank-nbo trailing updates.
template <typename T>
static int potrf_batched_mixed(T* A, int64_t n, int64_t ld, int64_t batch, int64_t bstride, T* dinv, int64_t dstride, int* info, int nbo,
                               int info_base, hipStream_t stream, RhsRide<T>* ride = nullptr) {
    const int nblk = (int)(n / GPK_DB), npb = nbo / GPK_DB;
    for (int jn = 0; jn < nblk; ++jn) {
        const bool below = jn + 1 < nblk;
        DiagArgs<T> d;
        d.A = A; d.ld = ld; d.bstride = bstride; d.off = (int64_t)jn * GPK_DB; d.n = (int)n;
        d.dinv = dinv; d.dinv_bstride = dstride; d.info = info; d.info_base = info_base;
        d.prof = g_diag_prof;
        d.zero_next = below ? 1 : 0;          // the control words of the step kernel behind it
        if (ride != nullptr) { d.rhs = ride->B; d.rhs_stride = ride->sB; }
        launch_diag<T>(d, (unsigned)batch, stream);
        GPK_CHECK_LAUNCH();
        if (!below) break;
        BatchStepArgs<T> p;
        p.rhs = ride != nullptr ? ride->B : nullptr;
        p.srhs = ride != nullptr ? ride->sB : 0;
        p.A = A; p.ld = ld; p.bstride = bstride; p.n = (int)n; p.batch = (int)batch;
        p.dinv = dinv; p.dstride = dstride; p.info = info;
        p.jn = jn;
        p.tm = nblk - jn - 1;
        const int jl = jn % npb;                          // block jn inside its outer panel
        if (jl == npb - 1) {                              // the panel is complete: its rank-nbo update of everything behind it
            p.K = nbo; p.tn = p.tm;
        } else if (g_batch_left) {
            // LEFT-LOOKING inside the outer panel: only the NEXT block column, with every column of the panel solved so far (128 (jl + 1)
            // deep) -- tm tiles per step instead of the halving's tm / 2 tm - 1 / tm tiles at depths 128 / 256 / 128: the same flops in
            // fewer, deeper visits of the panel's C tiles (a visit costs ~16 us whatever its depth)
            p.K = GPK_DB * (jl + 1); p.tn = 1;
        } else {
            int z = 0;
            while ((jl >> z) & 1) ++z;                    // trailing ones: the halving level that block closes
            p.K = GPK_DB << z;
            p.tn = (p.K / GPK_DB < p.tm) ? p.K / GPK_DB : p.tm;
        }
        p.nU = p.tm;
        for (int ti = 1; ti < p.tm; ++ti) p.nU += (ti < p.tn - 1) ? ti : p.tn - 1;
        p.opts = g_batch_opts;
        p.lag = (g_batch_lag + p.tm + p.nU - 1) / (p.tm + p.nU);
        const int per_xcd = (int)((batch + 7) / 8);
        if (p.lag > per_xcd) p.lag = per_xcd;
        if (p.lag < 1) p.lag = 1;
        const int64_t qlen = (int64_t)(per_xcd + p.lag) * (p.tm + p.nU);
        // algorithmic flops of the step: the solves at the TRSM count (128^3 per tile), the update at the symmetric count (a diagonal tile half)
        const int ndiag = p.tn < p.tm ? p.tn : p.tm;
        const double fl = (double)batch * ((double)p.tm * GPK_DB * GPK_DB * GPK_DB + (2.0 * (p.nU - ndiag) + ndiag) * GPK_DB * GPK_DB * (double)p.K);
        void* slot = gpk_prof_begin(160 + (sizeof(T) == 8 ? 8 : 0), fl, stream);
        hipLaunchKernelGGL((batch_mix_kernel<T>), dim3((unsigned)(8 * qlen)), dim3(256), 0, stream, p);
        gpk_prof_end(slot, stream);
        GPK_CHECK_LAUNCH();
    }
    if (ride != nullptr) ride->done = true;
    return GPK_OK;
}

template <typename T>
static int potrf_plain(T* A, int64_t n, int64_t ld, int64_t batch, int64_t bstride, T* dinv,
                       int* info, int nbo, int info_base, hipStream_t stream, int64_t rows = 0, RhsRide<T>* rhs_ride = nullptr) {
    if (n <= 0 || batch <= 0) return GPK_OK;
    if (rows < n) rows = n;
    if (rows > INT32_MAX) return GPK_ERR_ARG(2);
    if (ld < n) return GPK_ERR_ARG(3);
    if (rows > n && (batch != 1 || dinv == nullptr || !g_pipe)) return GPK_ERR_ARG(2);
    // one matrix: the pipelined panel (potrf_panel_pipe) takes whole matrices up to 4096 as ONE panel; above, the trailing matrix is
    // big enough for the rank-nbo update GEMM to be worth its launch (measured: N = 8192 5.76 ms at 1024, 5.97 at 2048, 6.07 at 512)
    if (nbo <= 0) nbo = (batch == 1 && dinv != nullptr) ? (n <= 4096 ? 4096 : (int)g_plain_nbo) : ((n >= 8192) ? 1024 : (n >= 2048 ? 512 : 256));
    if (nbo < GPK_DB || (nbo & (nbo - 1))) return GPK_ERR_ARG(9);   // 128 * 2^k
    if (info == nullptr) return GPK_ERR_ARG(7);
    if (dinv == nullptr && n > GPK_DB) return GPK_ERR_ARG(6);
    const int64_t nblk = gpk_cdiv(n, GPK_DB);
    const int64_t dstride = nblk * GPK_DB * GPK_DB;

    // batches of aligned matrices: one mixed-phase launch per 128-column step instead of lockstep launches per phase
    if (g_batch_mixed && batch >= g_batch_mixed_min && rows == n && dinv != nullptr && n % GPK_DB == 0 && n >= 4 * GPK_DB && nbo <= n &&
        (sizeof(T) == 4 || g_batch_mixed >= 2) && n <= 64 * 1024 && stream_has_all_cus(stream) && ld % Traits<T>::VEC == 0 && bstride % Traits<T>::VEC == 0 && (uintptr_t)A % 16 == 0 &&
        (uintptr_t)dinv % 16 == 0 && ld < GPK_PIPE_LD_MAX && batch <= INT32_MAX / 8)
        return potrf_batched_mixed<T>(A, n, ld, batch, bstride, dinv, dstride, info, nbo, info_base, stream, rhs_ride);

    PanelCtx<T> ctx{A, n, ld, batch, bstride, dinv, dstride, info, stream, info_base};
    ctx.rows = rows;
    // One matrix, pipelined panels: a panel's launch is chain-bound and leaves most of the chip idle, and all it needs of the
    // previous panel's rank-nbo update are its own columns.  So that update is split: the strip of the next panel's columns is a
    // GEMM launch of its own, the rest (the lower triangle right of the next panel) rides along in the next panel's launch as fill
    // tiles for its idle workers (N = 8192: 5.76 -> 5.4 ms; the last 6144 columns of N = 16384: 3.4 -> 2.9 ms).
    const bool ride = g_pipe && g_pipe_fill && batch == 1 && dinv != nullptr;
    int64_t owed = 0;      // width of the previous panel whose update is still owed to the columns from k0 on (0: nothing owed)
    for (int64_t k0 = 0; k0 < n; k0 += nbo) {
        const int64_t k1 = (k0 + nbo < n) ? k0 + nbo : n;
        bool done = false;
        if (owed > 0) {
            // the strip: columns [k0, k1), rows k0 .. n-1
            const T* P = A + k0 * ld + (k0 - owed);
            int st = gpk_gemm_launch<T>(true, true, rows - k0, k1 - k0, owed, T(-1), P, ld, bstride, P, ld, bstride, T(1), A + k0 * ld + k0, ld,
                                        bstride, batch, true, stream);
            if (st) return st;
            if (k1 < n && n - k0 > GPK_DB) {
                st = potrf_panel_pipe<T>(ctx, k0, nbo, &done, owed);
                if (st) return st;
            }
            if (!done && k1 < n) {      // the panel did not take it along: the rest of the update as a launch of its own
                const T* P2 = A + k1 * ld + (k0 - owed);
                st = gpk_gemm_launch<T>(true, true, rows - k1, n - k1, owed, T(-1), P2, ld, bstride, P2, ld, bstride, T(1), A + k1 * ld + k1, ld,
                                        bstride, batch, true, stream);
                if (st) return st;
            }
            owed = 0;
        }
        if (!done) {
            int pst = potrf_panel_any<T>(ctx, k0, nbo);
            if (pst) return pst;
        }
        if (k1 < n) {
            if (ride && n - k1 > GPK_DB) {
                owed = k1 - k0;
            } else {
                const T* P = A + k1 * ld + k0;
                int st = gpk_gemm_launch<T>(true, true, rows - k1, n - k1, k1 - k0, T(-1), P, ld, bstride, P, ld,
                                            bstride, T(1), A + k1 * ld + k1, ld, bstride, batch, true, stream);
                if (st) return st;
            }
        }
    }
    return GPK_OK;
}


// ---------------------------------------------------------------------------
// Look-ahead variant for ONE large matrix (n >= ~4096; the batched path above keeps every CU busy
// without it).  Outer blocks of nb columns (nb = 1024: fp64, 512: fp32); per outer step j
//
//   chain(j)      factor the nb x nb DIAGONAL block only (the recursion above, restricted to nb rows)
//                 and merge its 128-block inverses into W_j = inv(L_jj)  -- serial, latency-bound,
//                 a few workgroups at a time;
//   solve(j)      A[k1:, k0:k1] = T[k1:, :] W_j^T   ONE full-chip GEMM (k clipped to W's triangle) for all
//                 rows below the diagonal block, instead of ~30 tall-skinny launches per panel; the
//                 operand T is the panel as the previous trailing update left it, in a workspace, so
//                 the GEMM runs out of place straight into A;
//   diag(j+1)     A[k1:k2, k1:k2] -= P P^T  (small), then chain(j+1) is enqueued on a SECOND stream;
//   trail(j)      the persistent two-segment update (gemm_persist_kernel): the next panel's strip
//                 A[k2:, k1:k2] - P P^T  -> T, and the lower triangle A[k2:, k2:] -= P P^T in place,
//                 with one CU per XCD kept empty for chain(j+1) to run on meanwhile.
//
// So the serial chain of step j+1 runs under the trailing update of step j (depth-1 look-ahead),
// and what is left on the main stream is GEMM work at full-chip size.  W_j doubles as the merged
// diagonal-block inverse that the triangular solves need afterwards (`dinv_big`).
// ---------------------------------------------------------------------------
#include <mutex>

namespace {

struct LaDevice {
    hipStream_t aux = nullptr;
    hipStream_t priv = nullptr;      // stands in for the legacy default stream (see gpk_potrf_la_launch)
    std::vector<hipEvent_t> events;
};
std::mutex g_la_mutex;
LaDevice g_la_dev[64];
GPK_KNOB(int64_t, g_la_min_rows, 2048);     // tuning knob (gpk_tune(6, v)): overlap while the trailing matrix has >= this many rows
GPK_KNOB(int64_t, g_la_tail_rows, 0);       // tuning knob (gpk_tune(9, v)): the last this-many rows (and matrices up to this order) take the plain path;
                                  // 0 = 6144 (with the pipelined plain panels: fp64 N = 16384 26.6 ms at 6144, 27.6 at 4096, 27.0 at 8192; fp32 nb = 512 N = 16384 15.5 / 15.7)
GPK_KNOB(int, g_la_ps_mode, 0);             // tuning knob (gpk_tune(10, v)): panel GEMM as 0 = plain launch, 1 = persistent, 2 = persistent with paired column tiles
GPK_KNOB(int, g_la_strip_last, 1);          // tuning knob (gpk_tune(11, v))
GPK_KNOB(int, g_la_rejoin, 1);              // tuning knob (gpk_tune(18, v)): reserved CUs rejoin the trailing update after the chain
GPK_KNOB(int, g_la_mode, 1);                // tuning knob (gpk_tune(7, v)): 0 = same algorithm on one stream (no overlap), 1 = overlap
GPK_KNOB(int64_t, g_la_fuse_diag_rows, 9216);   // tuning knob (gpk_tune(40, v)): the update of the next diagonal block rides in the trailing update while that has >= this many rows (0: never)
GPK_KNOB(int, g_la_fuse_diag_nb, 512);          // tuning knob (gpk_tune(41, v)): ... and only for outer blocks up to this width
// Round 5: AGGREGATED trailing updates.  Column block c of the trailing matrix only has to be up to date when its own panel is
// factorised, so outer step j updates -- besides the next panel's block j + 1, always -- only the column blocks c = j + 2, j + 2 + m,
// j + 2 + 2m, ... (c == j mod m), each with ALL the panels it has not seen yet: m nb deep instead of nb.  Same flops, every C tile
// of the trailing matrix visited N / (m nb) times instead of N / nb times.  m = 1: the classic right-looking update.  Measured
// (profiles/r05_experiments.md section 1): m = 2 -0.9 % fp64 N = 16384, -2.3 % fp32 N = 32768; m = 3, 4 slower than m = 1 -- most of
// what a visit spends outside its k loop is covered by the CU's other workgroup already, and longer tiles cost at the end of a launch.
GPK_KNOB(int, g_la_agg, 2);                     // tuning knob (gpk_tune(47, v)): m
GPK_KNOB(int64_t, g_la_agg_min_rows, 0);        // tuning knob (gpk_tune(48, v)): aggregate only while the trailing matrix has >= this many rows (below: every column block, every step)

LaDevice* la_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    LaDevice& d = g_la_dev[dev];
    if (d.aux == nullptr) {
        unsigned keys[8];
        // no CU-masked stream on this device (refused by the runtime): the same algorithm runs on one stream, without the overlap
        if (gpk_helper_stream(&d.aux, keys) != GPK_OK) d.aux = nullptr;
    }
    return &d;
}

hipEvent_t la_event(LaDevice& d, size_t i) {
    while (d.events.size() <= i) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        d.events.push_back(e);
    }
    return d.events[i];
}

template <typename T>
int la_chain(T* A, int64_t n, int64_t ld, T* dinv128, T* dinv_big, int nb, int sb, T* tmp, int* info, int64_t j,
             hipStream_t s, int max_wgs = 0, unsigned* wait_word = nullptr, unsigned wait_value = 0) {
    const int64_t k0 = j * nb;
    const int64_t w = (n - k0 < nb) ? n - k0 : nb;
    T* Ab = A + k0 * ld + k0;
    T* d128 = dinv128 + (k0 / GPK_DB) * (int64_t)(GPK_DB * GPK_DB);
    PanelCtx<T> sub{Ab, w, ld, 1, 0, d128, 0, info, s, (int)k0, max_wgs, wait_word, wait_value};
    int st = potrf_panel_any<T>(sub, 0, nb);
    if (st) return st;
    // the explicit inverses of the sb x sb diagonal blocks of this nb-block (sb = nb: of the whole block)
    return gpk_trtri_merge_launch<T>(Ab, w, ld, 1, 0, d128, sb, dinv_big + (k0 / sb) * (int64_t)sb * sb, tmp, s);
}

}  // namespace

void gpk_potrf_shutdown() {
    std::lock_guard<std::mutex> lock(g_la_mutex);
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (int dev = 0; dev < 64; ++dev) {
        LaDevice& d = g_la_dev[dev];
        if (d.aux == nullptr && d.priv == nullptr && d.events.empty()) continue;
        (void)hipSetDevice(dev);
        if (d.priv != nullptr) {
            (void)hipStreamSynchronize(d.priv);
            (void)hipStreamDestroy(d.priv);
        }
        for (hipEvent_t e : d.events) (void)hipEventDestroy(e);
        d = LaDevice();             // (the helper stream itself belongs to gpk_helper_shutdown)
    }
    (void)hipSetDevice(cur);
}

void gpk_tune_potrf(int key, int64_t value) {
    if (key == 6) GPK_KNOB_SET(g_la_min_rows = value;);
    if (key == 7) GPK_KNOB_SET(g_la_mode = (int)value;);
    if (key == 9) GPK_KNOB_SET(g_la_tail_rows = value;);
    if (key == 10) GPK_KNOB_SET(g_la_ps_mode = (int)value;);
    if (key == 18) GPK_KNOB_SET(g_la_rejoin = (int)value;);
    if (key == 11) GPK_KNOB_SET(g_la_strip_last = (int)value;);
    if (key == 32) GPK_KNOB_SET(g_fused_step = (int)value;);
    if (key == 37) GPK_KNOB_SET(g_pipe = (int)value;);
    if (key == 40) GPK_KNOB_SET(g_la_fuse_diag_rows = value;);
    if (key == 41) GPK_KNOB_SET(g_la_fuse_diag_nb = (int)value;);
    if (key == 47) GPK_KNOB_SET(g_la_agg = (int)value;);
    if (key == 48) GPK_KNOB_SET(g_la_agg_min_rows = value;);
    if (key == 52) GPK_KNOB_SET(g_plain_nbo = value;);
    if (key == 53) GPK_KNOB_SET(g_batch_mixed = (int)value;);
    if (key == 54) GPK_KNOB_SET(g_batch_mixed_min = value;);
    if (key == 55) GPK_KNOB_SET(g_batch_lag = (int)value;);
    if (key == 56) GPK_KNOB_SET(g_batch_opts = (int)value;);
    if (key == 60) GPK_KNOB_SET(g_batch_left = (int)value;);
    if (key == 57) GPK_KNOB_SET(g_diag_two_per_cu = (int)value;);
    if (key == 38) GPK_KNOB_SET(g_pipe_fill = (int)value;);
    if (key == 39) GPK_KNOB_SET(g_pipe_panel_wgs = (int)value;);
}

#define GPK_LA_PAD 16
#ifndef GPK_LA_CTRL_PINGPONG
#define GPK_LA_CTRL_PINGPONG 0       // 1: the next step's control words zeroed on the helper stream instead of by a memset on the main stream in front of every
                                     // fork (`make ab ABFLAGS=-DGPK_LA_CTRL_PINGPONG=1`).  Measured flat on one box (fp64 N = 16384 + 2048 rows 33.11 against
                                     // 33.16 ms, fp32 N = 32768 101.45 against 101.56: the memset sits in a dependent-launch gap that is there anyway): OFF
                                     // (profiles/r06_ab_control_words_pingpong.log)
#endif
int64_t gpk_potrf_la_ws_elems_impl(int64_t n, int nb) {
    return (n > nb ? n : nb) * (int64_t)(nb + GPK_LA_PAD) + (int64_t)nb * nb / 4 + 16 + 128 + nb;   // panel, merge scratch, 128 elements >= two sets of control words, one block of the right-hand side
}

template <typename T>
static int potrf_la_body(LaDevice* dev, T* A, int64_t n, int64_t ld, T* dinv128, T* dinv_big, int nb, int sb, T* ws, int* info,
                         hipStream_t stream, int64_t rows, int flags);

template <typename T>
int gpk_potrf_la_launch(T* A, int64_t n, int64_t ld, T* dinv128, T* dinv_big, int nb, T* ws, int* info,
                        hipStream_t stream, int sb, int64_t rows, int flags) {
    if (n <= 0) return GPK_OK;
    if (rows < n) rows = n;
    if ((flags & GPK_ROWS_RHS) && rows < n + GPK_ROWS_RHS_STRIP) return GPK_ERR_ARG(2);
    if (rows > INT32_MAX) return GPK_ERR_ARG(2);
    if (rows > n && n % GPK_DB != 0) return GPK_ERR_ARG(2);      // rows under the matrix: whole 128-blocks only
    if (rows > n && gpk_cdiv(n, nb) > 64) return GPK_ERR_ARG(2);  // ... and at most 64 outer blocks (their column groups are a 64-bit mask)
    if (ld < n) return GPK_ERR_ARG(3);
    if (nb < 256 || nb > 4096 || (nb & (nb - 1))) return GPK_ERR_ARG(6);
    if (sb <= 0) sb = nb;
    if (sb < 256 || sb > nb || (sb & (sb - 1))) return GPK_ERR_ARG(10);
    if (dinv128 == nullptr || dinv_big == nullptr || ws == nullptr || info == nullptr) return GPK_ERR_ARG(4);

    std::lock_guard<std::mutex> lock(g_la_mutex);
    LaDevice* dev = la_device();
    if (dev == nullptr) return GPK_ERR_LAUNCH;
    if (stream != nullptr) return potrf_la_body<T>(dev, A, n, ld, dinv128, dinv_big, nb, sb, ws, info, stream, rows, flags);

    // The legacy default stream (what torch's default stream is) synchronises implicitly with every BLOCKING
    // stream -- and the CU-masked helper stream is one (hipExtStreamCreateWithCUMask takes no flags): each
    // launch on either would wait for the other, the overlap would be gone and then some (measured: 56 instead
    // of 41 ms per eval).  So a private non-blocking stream stands in for it: forked from the default stream
    // here, joined to it at the end; nothing else is enqueued on the default stream in between.
    if (dev->priv == nullptr && hipStreamCreateWithFlags(&dev->priv, hipStreamNonBlocking) != hipSuccess) {
        dev->priv = nullptr;
        return GPK_ERR_LAUNCH;
    }
    hipEvent_t e_in = la_event(*dev, 0), e_out = la_event(*dev, 1);
    if (e_in == nullptr || e_out == nullptr) return GPK_ERR_LAUNCH;
    if (hipEventRecord(e_in, nullptr) != hipSuccess || hipStreamWaitEvent(dev->priv, e_in, 0) != hipSuccess) return GPK_ERR_LAUNCH;
    const int st = potrf_la_body<T>(dev, A, n, ld, dinv128, dinv_big, nb, sb, ws, info, dev->priv, rows, flags);
    // joined whatever `st` is: what was enqueued on the private stream before a failure still uses the caller's buffers
    if (hipEventRecord(e_out, dev->priv) != hipSuccess || hipStreamWaitEvent(nullptr, e_out, 0) != hipSuccess) return st ? st : GPK_ERR_LAUNCH;
    return st;
}

// rows > n: `rows - n` more rows under the square matrix (PanelCtx::rows): every panel solve and trailing update carries them.
// flags & GPK_ROWS_RHS (round 6): the last GPK_ROWS_RHS_STRIP (64) of those rows are a strip whose FIRST row is one right-hand side b
// (n entries, contiguous; the other 63 are padding, so that the plain tail keeps whole 64-row strips -- one ragged row put its six
// panels and five updates on the bounds-checked kernels: +160 us at N = 16384): b comes out as L^{-1} b like
// every other row -- but through the look-ahead steps it is not a row of the GEMMs (a 129th tile row for one vector) but a vector of
// the single-column sweep (gpk_trsv_lower's two matrix-vector products per diagonal block), enqueued as soon as a panel is final on a THIRD stream with the
// helper stream's CU mask: memory-bound work beside the serial chain on the CUs the trailing update keeps empty, ordered by nothing but its own panels.
// The plain tail then carries it as the row it is.
template <typename T>
static int potrf_la_body(LaDevice* dev, T* A, int64_t n, int64_t ld, T* dinv128, T* dinv_big, int nb, int sb, T* ws, int* info,
                         hipStream_t stream, int64_t rows, int flags) {
    const bool rhs = (flags & GPK_ROWS_RHS) != 0;
    const int64_t Rall = rows;           // rows of the buffer (>= n), the right-hand side among them
    const int64_t R = rhs ? rows - GPK_ROWS_RHS_STRIP : rows;     // ... the rows the look-ahead's GEMMs carry
    T* bvec = rhs ? A + R * ld : nullptr;
    // 128 elements reserved: TWO sets of control words.  Step j's persistent update (and the helper's rejoin launch) count in set j & 1;
    // the other set is zeroed on the HELPER stream during the step (nobody uses it then: its last users were joined before the fork), so
    // that the next step finds its counters at zero without a memset -- one launch and one dependent-launch gap less on the main
    // stream per outer step.  Launches of a step in front of its fork (rare: a change of aggregation policy) count in the other set.
    unsigned* const ctrl_sets[2] = {reinterpret_cast<unsigned*>(ws), reinterpret_cast<unsigned*>(ws) + GPK_PERSIST_CTRL_WORDS};
    unsigned* ctrl = ctrl_sets[0];
    bool next_set_zeroed = false;        // the set of the step about to run has been zeroed by the previous step's helper-stream memset
    // n x nb, leading dimension nb + 16: with a power-of-two pitch the rows of a tile sit on a few memory channels and
    // the panel GEMM, which streams this buffer once, crawls (measured 2x)
    const int64_t ldt = nb + GPK_LA_PAD;
    T* Tp = ws + 128;
    T* tmp = Tp + (Rall > nb ? Rall : nb) * ldt;
    T* vtmp = tmp + (int64_t)nb * nb / 4 + 16;       // one block of the right-hand side (side stream only)
    const int64_t nblk = gpk_cdiv(n, nb);
    const int64_t per = (int64_t)sb * sb;      // one explicit inverse
    size_t ev = 2;                       // (events 0 and 1 belong to the default-stream stand-in)

    // Small matrices, and the tail of big ones, are chain-bound: there the plain right-looking recursion
    // (no explicit block inverse, no separate panel GEMM) is faster -- measured crossover ~5000 rows.
    auto finish_plain = [&](int64_t k) -> int {       // factor A[k:, k:] (all earlier panels applied) the plain way
        int s = potrf_plain<T>(A + k * ld + k, n - k, ld, 1, 0, dinv128 + (k / GPK_DB) * (int64_t)(GPK_DB * GPK_DB), info,
                               0, (int)k, stream, Rall - k);
        if (s) return s;
        if (flags & GPK_ROWS_NO_TAIL_INVERSES) return GPK_OK;       // (nobody is going to solve with the sb-wide inverses of the tail)
        return gpk_trtri_merge_launch<T>(A + k * ld + k, n - k, ld, 1, 0, dinv128 + (k / GPK_DB) * (int64_t)(GPK_DB * GPK_DB),
                                         sb, dinv_big + (k / sb) * per, Tp, stream);   // the panel workspace is free by now
    };
    int64_t tail_rows = g_la_tail_rows > 0 ? g_la_tail_rows : 6144;
    if (Rall > n && tail_rows < nb) tail_rows = nb;      // (rows under the matrix: the last column block's rows are solved by the plain tail)
    if (n <= tail_rows) return finish_plain(0);

    // the right-hand side's share of panel j (final on `stream` at the point of the call): for every sb-wide diagonal block of the
    // panel  b_q <- W_q b_q,  b[below, within n] -= L[below, q] b_q  -- on the side stream, behind an event of `stream`
    hipEvent_t e_side = nullptr;
    bool side_used = false;
    hipStream_t side = nullptr;          // the helper stream's sibling (same CU mask); none: the products run in line on `stream`
    if (rhs && g_la_mode == 1 && dev->aux != nullptr && gpk_helper_side_stream(&side) != GPK_OK) side = nullptr;
    auto rhs_panel = [&](int64_t j) -> int {
        if (!rhs) return GPK_OK;
        hipStream_t s = stream;
        if (side != nullptr) {
            hipEvent_t e = la_event(*dev, ev++);
            if (e == nullptr || hipEventRecord(e, stream) != hipSuccess || hipStreamWaitEvent(side, e, 0) != hipSuccess) return GPK_ERR_LAUNCH;
            side_used = true;
            s = side;
        }
        const int64_t k0 = j * nb;
        for (int64_t c = 0; c < nb; c += sb) {
            const int64_t r0 = k0 + c, r1 = r0 + sb;
            const int st1 = gpk_trsv_step_launch<T>(dinv_big + (r0 / sb) * per, sb, sb, A + r1 * ld + r0, ld, n - r1, bvec + r0, bvec + r1, vtmp, s);
            if (st1) return st1;
        }
        return GPK_OK;
    };
    // ... and `stream` behind everything the side stream has been given (before the plain tail reads the vector as a row, and on
    // every way out: the side stream's work references the caller's buffers)
    auto rhs_join = [&]() -> int {
        if (!side_used) return GPK_OK;
        if (e_side == nullptr) e_side = la_event(*dev, ev++);
        if (e_side == nullptr || hipEventRecord(e_side, side) != hipSuccess || hipStreamWaitEvent(stream, e_side, 0) != hipSuccess) return GPK_ERR_LAUNCH;
        return GPK_OK;
    };
    // (a lambda, so that the side stream is joined on every way out)
    auto steps = [&]() -> int {
        int st = la_chain<T>(A, n, ld, dinv128, dinv_big, nb, sb, tmp, info, 0, stream);
        if (st) return st;
        if (nblk > 1) {   // the first panel enters the workspace as it is
            st = gpk_copy2d_launch<T>(A + (int64_t)nb * ld, ld, 0, Tp + (int64_t)nb * ldt, ldt, 0, R - nb, nb, 1, stream);
            if (st) return st;
        }
        // applied[c]: panels 0 .. applied[c] - 1 have been subtracted from column block c (columns c nb ..., the rows from its diagonal
        // block down).  Column block c is updated with its pending panels  L[:, applied[c] nb : k1]  -- contiguous columns of the factor.
        const int agg = (g_la_agg > 1 && nblk <= 64) ? (int)g_la_agg : 1;      // (the column groups of a segment are a 64-bit mask)
        std::vector<int> applied((size_t)nblk, 0);
        struct ColGroup { int from; uint64_t mask; };          // column blocks (bit c - c_first) that are `from` panels deep
        auto group_cols = [&](int64_t c_first, int64_t j, bool all, std::vector<ColGroup>& out) {
            out.clear();
            // more column blocks than a mask has bits (N / nb > 64): agg is 1 then, every block is updated in every step and they all
            // have the same depth -- ONE group that means "every column" (all-ones; col_segment turns it into the plain triangle)
            const bool wide = nblk - c_first > 64;
            for (int64_t c = c_first; c < nblk; ++c) {
                if (!all && (c - j) % agg != 0) continue;
                size_t g = 0;
                while (g < out.size() && out[g].from != applied[(size_t)c]) ++g;
                if (g == out.size()) out.push_back(ColGroup{applied[(size_t)c], 0});
                if (wide) out[g].mask = ~(uint64_t)0;
                else out[g].mask |= (uint64_t)1 << (c - c_first);
                applied[(size_t)c] = (int)(j + 1);
            }
        };
        // column blocks c_first ... of the trailing matrix A[kc:, kc:] (kc = c_first nb), one segment per depth
        auto col_segment = [&](const ColGroup& cg, int64_t c_first, int64_t k1) -> GpkSeg<T> {
            const int64_t kc = c_first * nb, ka = (int64_t)cg.from * nb;
            const T* P = A + kc * ld + ka;
            const uint64_t every = (nblk - c_first >= 64) ? ~(uint64_t)0 : (((uint64_t)1 << (nblk - c_first)) - 1);
            return GpkSeg<T>{R - kc, n - kc, k1 - ka, P, ld, P, ld, A + kc * ld + kc, ld, A + kc * ld + kc, ld, 1, 0, 0,
                             (cg.mask == every && R == n) ? 0 : cg.mask, nb};
        };
        std::vector<ColGroup> groups;
        for (int64_t j = 0; j + 1 < nblk; ++j) {
            const int64_t k0 = j * nb, k1 = k0 + nb;
            const int64_t k2 = (k1 + nb < n) ? k1 + nb : n;
            unsigned* const ctrl_step = ctrl_sets[j & 1];       // this step's persistent update / rejoin
            ctrl = ctrl_sets[(j + 1) & 1];                      // every other launch of the step (they zero it themselves)
            const bool step_set_zeroed = next_set_zeroed;
            next_set_zeroed = false;
            // solve(j): rows k1.. of panel j  (k clipped to W's triangle)
            if (sb < nb) {
                // the explicit inverses are sb wide (fp32: the error of the posterior mean grows with the width of an explicit inverse): block
                // substitution over the nb / sb column blocks of the panel -- X_i = (T_i - sum_{p<i} X_p L_ip^T) W_ii^T, same flops as the
                // single product, 2 nb / sb - 1 launches
                for (int64_t i = 0; i * sb < nb && st == GPK_OK; ++i) {
                    const int64_t c = i * sb;
                    if (i > 0)
                        st = gpk_gemm_launch<T>(true, true, R - k1, sb, c, T(-1), A + k1 * ld + k0, ld, 0, A + (k0 + c) * ld + k0, ld, 0, T(1),
                                                Tp + k1 * ldt + c, ldt, 0, 1, 0, stream);
                    if (st == GPK_OK)
                        st = gpk_gemm_launch<T>(true, true, R - k1, sb, sb, T(1), Tp + k1 * ldt + c, ldt, 0, dinv_big + (k0 / sb + i) * per, sb, 0, T(0),
                                                A + k1 * ld + k0 + c, ld, 0, 1, 8, stream);
                }
            } else if (g_la_ps_mode == 0) {
                st = gpk_gemm_launch<T>(true, true, R - k1, nb, nb, T(1), Tp + k1 * ldt, ldt, 0, dinv_big + j * per, nb, 0, T(0),
                                        A + k1 * ld + k0, ld, 0, 1, 8, stream);
            } else {
                GpkSeg<T> ps{R - k1, nb, nb, Tp + k1 * ldt, ldt, dinv_big + j * per, nb, nullptr, 0, A + k1 * ld + k0, ld, 0, g_la_ps_mode};
                st = gpk_gemm_persist_launch<T>(&ps, 1, T(1), ctrl, 0, stream);
            }
            if (st) return st;
            st = rhs_panel(j);
            if (st) return st;
            if (n - k1 <= tail_rows) {
                // last look-ahead step: the whole trailing matrix is brought up to date in place (every column block with the panels it
                // has not seen: one segment per depth), the rest is factorised the plain way
                group_cols(j + 1, j, true, groups);
                for (size_t g0 = 0; g0 < groups.size(); g0 += GPK_PERSIST_MAX_SEG) {
                    GpkSeg<T> sg[GPK_PERSIST_MAX_SEG];
                    int ns = 0;
                    for (size_t g = g0; g < groups.size() && ns < GPK_PERSIST_MAX_SEG; ++g) sg[ns++] = col_segment(groups[g], j + 1, k1);
                    st = gpk_gemm_persist_launch<T>(sg, ns, T(-1), ctrl, 0, stream);
                    if (st) return st;
                }
                st = rhs_join();          // (the tail carries the right-hand side as a row: all look-ahead panels applied to it first)
                if (st) return st;
                return finish_plain(k1);
            }
            // the next panel's column block (diagonal block + strip): the panels it has not seen yet (one with m <= 2, m - 1 in general)
            const int64_t ka1 = (int64_t)applied[(size_t)(j + 1)] * nb, kd1 = k1 - ka1;
            applied[(size_t)(j + 1)] = (int)(j + 1);
            const T* P1 = A + k1 * ld + ka1;
            const bool overlap = g_la_mode == 1 && dev->aux != nullptr && (n - k2) >= g_la_min_rows;
            // diag(j+1): a launch of its own -- or, while the trailing update is long enough to hide a chain that starts a tile later, the
            // FIRST tiles of that update: the chain's kernel waits for them through a counter word instead of a kernel boundary
            // (one dependent launch less on the main stream per step)
            // (measured: fp32 N = 32768 with 512-blocks, 60 steps: cfg3 127.8 -> 127.0 ms; fp64 N = 16384 with 1024-blocks, 10 steps whose
            // 36 diagonal tiles take ~300 us as 128-tiles of the persistent kernel against 46 us as a launch of 64-tiles: 26.7 -> 26.8 ms.
            // So: blocks up to 512 only.)
            const bool fuse_diag = overlap && nb <= g_la_fuse_diag_nb && g_la_fuse_diag_rows > 0 && (n - k2) >= g_la_fuse_diag_rows && g_pipe &&
                                   (k2 - k1) > GPK_DB;
            if (!fuse_diag) {
                st = gpk_gemm_launch<T>(true, true, k2 - k1, k2 - k1, kd1, T(-1), P1, ld, 0, P1, ld, 0, T(1),
                                        A + k1 * ld + k1, ld, 0, 1, 1, stream);
                if (st) return st;
            }
            // which column blocks behind the next panel are updated in this step, and how deep
            if (k2 < n) group_cols(j + 2, j, agg == 1 || (n - k2) < g_la_agg_min_rows, groups);
            else groups.clear();
            {   // more depths than the launch below has segments for (only at a change of policy): in-place launches of their own, up front
                const size_t room = (size_t)(GPK_PERSIST_MAX_SEG - 1 - (fuse_diag ? 1 : 0));
                while (groups.size() > room) {
                    GpkSeg<T> sg = col_segment(groups.back(), j + 2, k1);
                    groups.pop_back();
                    st = gpk_gemm_persist_launch<T>(&sg, 1, T(-1), ctrl, 0, stream);
                    if (st) return st;
                }
            }
            hipEvent_t e_fork = nullptr, e_join = nullptr;
            if (overlap) {
                e_fork = la_event(*dev, ev++);
                e_join = la_event(*dev, ev++);
                if (e_fork == nullptr || e_join == nullptr) return GPK_ERR_LAUNCH;
                // the tile counter is zeroed BEFORE the fork: the helper stream's rejoin launch reads it, so it must be ordered behind
                // (by the previous step's helper-stream memset, joined since; the first overlapped step zeroes its own)
                if (!step_set_zeroed && hipMemsetAsync(ctrl_step, 0, GPK_PERSIST_CTRL_WORDS * sizeof(unsigned), stream) != hipSuccess) return GPK_ERR_LAUNCH;
                if (hipEventRecord(e_fork, stream) != hipSuccess) return GPK_ERR_LAUNCH;
            } else {
                st = la_chain<T>(A, n, ld, dinv128, dinv_big, nb, sb, tmp, info, j + 1, stream);
                if (st) return st;
            }
            GpkPersistSaved saved;
            saved.valid = 0;
            // Everything between the fork and the join: on ANY failure in here the helper stream is still joined to `stream` below --
            // work already enqueued on it references `ws`, `dinv128` and `A`, which the caller releases as soon as it sees the error
            // (a caching allocator would hand that memory to later work on the caller's stream).
            auto forked = [&]() -> int {
                // trail(j) goes to the device BEFORE the ~45 launches of the chain are enqueued: the host needs
                // ~0.3 ms for those, which the main stream would otherwise spend idle
                if (k2 < n) {
                    const T* P2 = A + k2 * ld + ka1;
                    GpkSeg<T> seg[GPK_PERSIST_MAX_SEG];
                    int ns = 0;
                    if (fuse_diag) seg[ns++] = GpkSeg<T>{k2 - k1, k2 - k1, kd1, P1, ld, P1, ld, A + k1 * ld + k1, ld, A + k1 * ld + k1, ld, 1, 0, 1};
                    const GpkSeg<T> strip{R - k2, k2 - k1, kd1, P2, ld, P1, ld, A + k2 * ld + k1, ld, Tp + k2 * ldt, ldt, 0, 0};
                    // the strip is what the next panel GEMM streams: written last, it is still in the Infinity Cache
                    if (!g_la_strip_last) seg[ns++] = strip;
                    for (const ColGroup& cg : groups) seg[ns++] = col_segment(cg, j + 2, k1);
                    if (g_la_strip_last) seg[ns++] = strip;
                    const int s2 = gpk_gemm_persist_launch<T>(seg, ns, T(-1), ctrl_step, overlap ? 1 : 0, stream, &saved, overlap);
                    if (s2) return s2;
                }
                if (overlap) {
                    if (hipStreamWaitEvent(dev->aux, e_fork, 0) != hipSuccess) return GPK_ERR_LAUNCH;
                    // the NEXT step's set (free since the join in front of this fork), off the main stream
                    if (GPK_LA_CTRL_PINGPONG) {
                        if (hipMemsetAsync(ctrl, 0, GPK_PERSIST_CTRL_WORDS * sizeof(unsigned), dev->aux) != hipSuccess) return GPK_ERR_LAUNCH;
                        next_set_zeroed = true;
                    }
                    const int s2 = fuse_diag ? la_chain<T>(A, n, ld, dinv128, dinv_big, nb, sb, tmp, info, j + 1, dev->aux, 8, ctrl_step + 2, (unsigned)saved.signal_tiles)
                                             : la_chain<T>(A, n, ld, dinv128, dinv_big, nb, sb, tmp, info, j + 1, dev->aux, 8);
                    if (s2) return s2;
                    if (g_la_rejoin && saved.valid)         // chain done: the reserved CUs take tiles of the update that is still running
                        return gpk_gemm_persist_rejoin<T>(&saved, dev->aux);
                }
                return GPK_OK;
            };
            st = forked();
            if (overlap) {
                const bool joined = hipEventRecord(e_join, dev->aux) == hipSuccess && hipStreamWaitEvent(stream, e_join, 0) == hipSuccess;
                if (!joined && st == GPK_OK) st = GPK_ERR_LAUNCH;
            }
            if (st) return st;
        }
        return GPK_OK;
    };
    const int st_all = steps();
    const int st_join = rhs_join();
    return st_all ? st_all : st_join;
}

template int gpk_potrf_la_launch<double>(double*, int64_t, int64_t, double*, double*, int, double*, int*, hipStream_t, int, int64_t, int);
template int gpk_potrf_la_launch<float>(float*, int64_t, int64_t, float*, float*, int, float*, int*, hipStream_t, int, int64_t, int);

template <typename T>
int gpk_potrf_launch(T* A, int64_t n, int64_t ld, int64_t batch, int64_t bstride, T* dinv,
                     int* info, int nbo, hipStream_t stream) {
    return potrf_plain<T>(A, n, ld, batch, bstride, dinv, info, nbo, 0, stream);
}

template <typename T>
int gpk_potrf_rhs_launch(T* A, int64_t n, int64_t ld, int64_t batch, int64_t bstride, T* dinv, int* info, int nbo, T* B, int64_t sB, T* tmp,
                         hipStream_t stream) {
    if (B == nullptr || tmp == nullptr || dinv == nullptr) return GPK_ERR_ARG(9);
    if (batch > 1 && sB < n) return GPK_ERR_ARG(10);
    RhsRide<T> ride{B, sB, tmp, false};
    int st = potrf_plain<T>(A, n, ld, batch, bstride, dinv, info, nbo, 0, stream, 0, &ride);
    if (st) return st;
    // the paths that do not take it along (lockstep launches, one matrix): the sweep behind the factorisation, as the caller would
    if (!ride.done) st = gpk_trsv_launch<T>(A, n, ld, bstride, dinv, GPK_DB, B, 1, 1, sB, tmp, batch, stream);
    return st;
}
template int gpk_potrf_rhs_launch<double>(double*, int64_t, int64_t, int64_t, int64_t, double*, int*, int, double*, int64_t, double*, hipStream_t);
template int gpk_potrf_rhs_launch<float>(float*, int64_t, int64_t, int64_t, int64_t, float*, int*, int, float*, int64_t, float*, hipStream_t);

template <typename T>
int gpk_potrf_rows_launch(T* A, int64_t n, int64_t rows, int64_t ld, T* dinv, int* info, hipStream_t stream) {
    return potrf_plain<T>(A, n, ld, 1, 0, dinv, info, 0, 0, stream, rows);
}
template int gpk_potrf_rows_launch<double>(double*, int64_t, int64_t, int64_t, double*, int*, hipStream_t);
template int gpk_potrf_rows_launch<float>(float*, int64_t, int64_t, int64_t, float*, int*, hipStream_t);

template int gpk_potrf_launch<double>(double*, int64_t, int64_t, int64_t, int64_t, double*, int*,
                                      int, hipStream_t);
template int gpk_potrf_launch<float>(float*, int64_t, int64_t, int64_t, int64_t, float*, int*, int,
                                     hipStream_t);
