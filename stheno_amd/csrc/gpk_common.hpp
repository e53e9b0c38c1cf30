// gpk_common.hpp -- shared device/host helpers for the gpk (GP kernels) library.
//
// Target: AMD Instinct MI355X (gfx950, CDNA4) only.  64-wide wavefronts, MFMA
// 16x16x4 in f64/f32, 160 KiB LDS per CU, 8 XCDs x 32 CUs.
//
// Storage convention everywhere in this library: ROW-MAJOR matrices with an
// explicit leading dimension in ELEMENTS, an optional batch dimension with a
// batch stride in ELEMENTS, raw device pointers, a hipStream_t.  No function
// allocates, frees, retains pointers or synchronises (see include/gpk.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Tuning knobs.  A RELEASE build (the default) has none: every "knob" is a compile-time constant at its measured optimum, the
// branches it guards are folded away and gpk_tune() does nothing.  `make dev` (-DGPK_DEV_KNOBS, into dev/) builds the library with
// mutable knobs for A/B measurements through `gpk_selftest --set KEY VALUE` / GPK_DEV=1 GPK_TUNE=... .
#ifdef GPK_DEV_KNOBS
#define GPK_KNOB(type, name, value) type name = (value)
#define GPK_KNOB_SET(stmt) stmt
#else
#define GPK_KNOB(type, name, value) constexpr type name = (value)
#define GPK_KNOB_SET(stmt) ((void)0)
#endif

#define GPK_WAVE 64
#define GPK_TILE 128      // GEMM block tile (rows and cols)
#define GPK_DB 128        // diagonal block factorised in LDS by one workgroup
#define GPK_MAX_TERMS 8

// status codes (mirrored in include/gpk.h)
#define GPK_OK 0
#define GPK_ERR_ARG(i) (-(i))
#define GPK_ERR_LAUNCH (-100)

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename T>
struct Traits;

// f64: v_mfma_f64_16x16x4_f64.  A/B: one f64 per lane, lane l holds
// A[row = l & 15][k = l >> 4] and B[k = l >> 4][col = l & 15].  C/D: 4 f64 per
// lane, reg i of lane l is C[row = (l >> 4) + 4 i][col = l & 15]  (NOT the f32
// map -- see cdna_hip_programming.md section 3).
template <>
struct Traits<double> {
    typedef f64x4 acc_t;
    typedef f64x2 vec_t;                 // 16-byte global/LDS vector
    static constexpr int VEC = 2;        // elements per 16 bytes
    static constexpr int BK = 16;        // K-chunk staged per LDS stage = 128 bytes
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int lane, int i) { return (lane >> 4) + 4 * i; }
    // row j of a 16x16 C/D tile lives in accumulator register rowi(j) of the 16 lanes of lane group rowq(j)
    static __host__ __device__ constexpr int rowq(int j) { return j & 3; }
    static __host__ __device__ constexpr int rowi(int j) { return j >> 2; }
};

// f32: v_mfma_f32_16x16x4_f32 (exact f32 FMA chain, f32 vector rate).  A/B as
// above; C/D reg i of lane l is C[row = 4 (l >> 4) + i][col = l & 15].
template <>
struct Traits<float> {
    typedef f32x4 acc_t;
    typedef f32x4 vec_t;
    static constexpr int VEC = 4;
    static constexpr int BK = 32;        // 128 bytes
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int lane, int i) { return (lane >> 4) * 4 + i; }
    static __host__ __device__ constexpr int rowq(int j) { return j >> 2; }
    static __host__ __device__ constexpr int rowi(int j) { return j & 3; }
};

static inline int64_t gpk_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// The workgroup barrier in front of a PUBLICATION (a flag / counter another workgroup waits for): every wave's global stores are
// acknowledged by the L2 before any thread passes it.  __syncthreads() alone does not do that on gfx950 -- its workgroup-scope
// release compiles to `s_waitcnt vmcnt(63) ... ; s_barrier`, the stores of the other waves may still be in flight when thread 0
// goes on to its agent-scope release (which waits for thread 0's wave only) and sets the flag.  Round 6 found it with a batched fp64
// factorisation whose update tiles read a panel row before the solve tile's last stores had landed; the publications of rounds 2-5
// (pipelined panel, panel step, persistent update's signal, the resident sweep) had the same hole behind a write-back that happened
// to cover it.
#if defined(__HIPCC__)
__device__ __forceinline__ void gpk_barrier_stores_done() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
#endif

#define GPK_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return GPK_ERR_LAUNCH;        \
    } while (0)

// ---------------------------------------------------------------------------
// Internal (C++) entry points shared between translation units.  All enqueue
// on `stream` and return a status; none synchronise.
// ---------------------------------------------------------------------------

// op(A) is M x K, op(B) is N x K ("B transposed" convention):
//   C[m][n] = alpha * sum_k a(m,k) b(n,k) + beta * C[m][n]
// a_kmaj: A stored M x K row-major (k contiguous), else stored K x M (m contiguous).
// b_kmaj: B stored N x K row-major (k contiguous), else stored K x N (n contiguous).
// flags bit 0 (lower_only): compute only tiles with tile_col <= tile_row (SYRK-style).
// flags bit 1 (tri_k): both operands vanish for k < their row index (lower-triangular
//   factors stored K x M): the k loop of tile row m0 starts at k = m0.
// flags bit 2 (tri_k_lower): A (M x K) is lower triangular: the k loop of tile row m0 stops at m0 + tile.
// flags bit 3 (tri_k_lower_b): B (N x K) is lower triangular: the k loop of tile column n0 stops at n0 + tile.
// merged diagonal-block sizes of the solves: 128 * 2^k up to 4096
static inline bool gpk_valid_sb(int sb) { return sb >= 128 && sb <= 4096 && (sb & (sb - 1)) == 0; }

template <typename T>
int gpk_gemm_launch(bool a_kmaj, bool b_kmaj, int64_t M, int64_t N, int64_t K, T alpha,
                    const T* A, int64_t lda, int64_t sA, const T* B, int64_t ldb, int64_t sB,
                    T beta, T* C, int64_t ldc, int64_t sC, int64_t batch, int flags,
                    hipStream_t stream);

// Same with a second batch level (blockIdx.z) -- used to batch over regularly strided
// sub-blocks of one matrix.
template <typename T>
int gpk_gemm_launch2(bool a_kmaj, bool b_kmaj, int64_t M, int64_t N, int64_t K, T alpha,
                     const T* A, int64_t lda, int64_t sA, int64_t sA2, const T* B, int64_t ldb,
                     int64_t sB, int64_t sB2, T beta, T* C, int64_t ldc, int64_t sC, int64_t sC2,
                     int64_t batch, int64_t batch2, int flags, hipStream_t stream,
                     const T* colscale = nullptr, T* colss = nullptr, int64_t ldss = 0);     // fused column scaling / sums of squares (gpk_gemm_colscale)

// Persistent two-problem update (see gemm_persist_kernel in gpk_gemm.hip): each segment is
//   C[m][n] = Cin[m][n] + alpha * sum_k A[m][k] B[n][k]     (A: M x K, B: N x K, both k contiguous)
// lower_only: tiles on/below the diagonal only.  ctrl: GPK_PERSIST_CTRL_WORDS unsigned words of device
// scratch (zeroed by the launch).  reserve: keep one CU per XCD free of this kernel's workgroups.
#define GPK_PERSIST_CTRL_WORDS 64
// The library's helper stream (one per device, created on first use): masked to one CU per XCD; keys[x] = the
// HW_ID key (+1) of that CU on XCD x, 0 if nothing is reserved.
int gpk_helper_stream(hipStream_t* aux, unsigned keys[8]);
int gpk_helper_side_stream(hipStream_t* side);      // a second stream on the helper stream's CUs; nullptr where there is no CU mask
void gpk_helper_shutdown();
void gpk_potrf_shutdown();
template <typename T>
struct GpkSeg {
    int64_t M, N, K;
    const T* A; int64_t lda;
    const T* B; int64_t ldb;
    const T* Cin; int64_t ldcin;
    T* C; int64_t ldc;
    int lower_only;
    int tri_b;      // B (N x K) is lower triangular: k stops at the column tile's last column (2: pair column tiles c, n-1-c)
    int signal;     // every finished tile of this segment is announced: ctrl[2] += 1 behind an agent-scope release (somebody polls it)
    // Round 5 (aggregated trailing updates of the look-ahead Cholesky): a square lower-only segment restricted to COLUMN GROUPS --
    // group g = columns [g grp, (g + 1) grp) and the rows from g grp down; bit g of colmask set = the group is part of the segment
    // (0 = every column: the whole lower triangle).  grp: a multiple of 128.
    uint64_t colmask;
    int64_t grp;
};
// Cin == nullptr: C = alpha * A B^T (nothing is read from C).  Up to GPK_PERSIST_MAX_SEG segments per launch, handed out in order.
#define GPK_PERSIST_MAX_SEG 4
struct GpkPersistSaved {       // what a reserving launch was made of, for gpk_gemm_persist_rejoin
    alignas(16) char bytes[2048];
    int ts, edge, per_cu, valid;
    int signal_tiles;          // tiles of the segments with `signal` set: what ctrl[2] counts up to (set whenever `saved` is given)
};
template <typename T>
int gpk_gemm_persist_launch(const GpkSeg<T>* segs, int nseg, T alpha, unsigned* ctrl, int reserve,
                            hipStream_t stream, GpkPersistSaved* saved = nullptr, bool ctrl_zeroed = false);
template <typename T>
int gpk_gemm_persist_rejoin(const GpkPersistSaved* saved, hipStream_t helper_stream);

// One 128-column step below a factorised diagonal block: panel solve + rank-128 update of the rest of the outer panel, one launch
// (panel_step_kernel in gpk_gemm.hip).  flags: ceil((n - c - 128) / 32) zeroed words of device scratch.
template <typename T>
int gpk_panel_step_launch(T* A, int64_t n, int64_t ld, int64_t c, const T* W, int64_t ke, unsigned* flags, hipStream_t stream);

template <typename T>
int gpk_potrf_launch(T* A, int64_t n, int64_t ld, int64_t batch, int64_t bstride, T* dinv,
                     int* info, int nbo, hipStream_t stream);

// Look-ahead Cholesky of one large matrix (gpk_potrf.hip).  dinv_big: [ceil(n/nb)][nb][nb] (receives the
// inverses of the nb x nb diagonal blocks of L); ws: gpk_potrf_la_ws_elems_impl(n, nb) elements.
int64_t gpk_potrf_la_ws_elems_impl(int64_t n, int nb);
template <typename T>
int gpk_potrf_la_launch(T* A, int64_t n, int64_t ld, T* dinv128, T* dinv_big, int nb, T* ws, int* info,
                        hipStream_t stream, int sb = 0, int64_t rows = 0, int flags = 0);     // sb: width of the explicit inverses (0 = nb); dinv_big: [ceil(n/sb)][sb][sb]
                                                                                 // rows > n: rows under the matrix (gpk_potrf_la_rows in gpk.h)

// One matrix with `rows - n` more rows under it (gpk_potrf_rows in gpk.h), the plain path: pipelined panels.
template <typename T>
int gpk_potrf_rows_launch(T* A, int64_t n, int64_t rows, int64_t ld, T* dinv, int* info, hipStream_t stream);

// measurement hooks (gpk_prof_* in gpk.h) for launches outside gpk_gemm.hip: HIP events around a launch, filed under `variant`
void* gpk_prof_begin(int variant, double flops, hipStream_t stream);
void gpk_prof_end(void* slot, hipStream_t stream);

void gpk_tune_gemm(int key, int64_t value);
void gpk_tune_potrf(int key, int64_t value);
void gpk_tune_kmat(int key, int64_t value);
void gpk_tune_solve(int key, int64_t value);
void gpk_set_diag_prof(long long* dev_buf);
void gpk_set_tile_prof(long long* dev_buf);

template <typename T>
int gpk_copy2d_launch(const T* src, int64_t lds, int64_t ss, T* dst, int64_t ldd, int64_t sd,
                      int64_t rows, int64_t cols, int64_t batch, hipStream_t stream);

template <typename T>
int gpk_set_identity_launch(T* dst, int64_t n, int64_t ld, int64_t sd, int64_t batch,
                            hipStream_t stream);

template <typename T>
int gpk_trtri_merge_launch(const T* L, int64_t n, int64_t ld, int64_t batch, int64_t bstride,
                           const T* dinv128, int sb, T* dinv_sb, T* tmp, hipStream_t stream);
template <typename T>
int gpk_trsm_launch(const T* L, int64_t n, int64_t ld, int64_t sL, const T* dinv_sb, int sb, T* B,
                    int64_t nrhs, int64_t ldb, int64_t sB, T* tmp, int64_t batch, hipStream_t stream,
                    T* X = nullptr, int64_t ldx = 0, int64_t sX = 0);
template <typename T>
int gpk_trsv_launch(const T* L, int64_t n, int64_t ld, int64_t sL, const T* dinv_sb, int sb, T* B,
                    int nrhs, int64_t ldb, int64_t sB, T* tmp, int64_t batch, hipStream_t stream);
template <typename T>
int gpk_logdet_launch(const T* L, int64_t n, int64_t ld, int64_t sL, int64_t batch, T* out,
                      hipStream_t stream);
int64_t gpk_colreduce_nchunks_impl(int64_t rows);
template <typename T>
int gpk_colreduce_launch(const T* V, int64_t rows, int64_t cols, int64_t ld, int64_t sV, const T* w,
                         int64_t sw, T* odot, T* oss, T* ws, int64_t batch, hipStream_t stream);
template <typename T>
int gpk_rowreduce_launch(const T* Z, int64_t rows, int64_t n, int64_t ld, const T* w, T* odot, T* oss, hipStream_t stream);
template <typename T>
int gpk_tril_launch(T* A, int64_t n, int64_t ld, int64_t sA, int64_t batch, hipStream_t stream);
template <typename T>
int gpk_sum_lower_launch(const T* parts, int64_t nparts, int64_t n, int64_t ldp, int64_t sP, T* out, int64_t ldo, hipStream_t stream);
template <typename T>
int gpk_add_diag_launch(T* A, int64_t n, int64_t ld, int64_t sA, T s, const T* v, int64_t sv,
                        int64_t batch, hipStream_t stream);
template <typename T>
int gpk_kmat_launch(const int* kinds, const double* variances, const double* inv_ls, int nterms,
                    const T* X, int64_t n, int64_t ldx, int64_t sX, const T* Y, int64_t m, int64_t ldy,
                    int64_t sY, int d, T* out, int64_t ld, int64_t sO, int64_t batch, int lower_only,
                    int symmetric, double diag_add, const T* diag_vec, int64_t sDiag, int accumulate,
                    hipStream_t stream);
template <typename T>
int gpk_kdiag_launch(const int* kinds, const double* variances, const double* inv_ls, int nterms,
                     const T* X, int64_t n, int64_t ldx, int64_t sX, int d, T* out, int64_t sO,
                     int64_t batch, hipStream_t stream);
template <typename T>
int gpk_scale_cols_launch(T* V, int64_t rows, int64_t cols, int64_t ld, int64_t sV, const T* s, int64_t ss,
                          int64_t batch, hipStream_t stream);
template <typename T>
int gpk_symmetrize_launch(T* A, int64_t n, int64_t ld, int64_t sA, int64_t batch, hipStream_t stream);
template <typename T>
int gpk_gemv_launch(int64_t M, int64_t K, int nrhs, T alpha, const T* A, int64_t lda, int64_t sA,
                    const T* x, int64_t ldx, int64_t sx, T beta, T* y, int64_t ldy, int64_t sy,
                    int64_t batch, hipStream_t stream);
// one diagonal block of the single-column sweep (gpk_trsv_lower's inner step) on a contiguous vector:
//   bq (rq entries) <- W bq,   bbelow (nbelow entries) -= Lbelow (nbelow x rq, ld) bq;   tmp: rq elements
template <typename T>
int gpk_trsv_step_launch(const T* W, int64_t ldw, int64_t rq, const T* Lbelow, int64_t ld, int64_t nbelow, T* bq, T* bbelow, T* tmp,
                         hipStream_t stream);
template <typename T>
int gpk_trsv_batch_step_launch(const T* L, int64_t n, int64_t ld, int64_t sL, const T* dinv, int64_t sD, T* B, int64_t sB, T* tmp,
                               int64_t batch, int q, hipStream_t stream);
// gpk_potrf + one right-hand side per matrix solved along (gpk_potrf_rhs in gpk.h)
template <typename T>
int gpk_potrf_rhs_launch(T* A, int64_t n, int64_t ld, int64_t batch, int64_t bstride, T* dinv, int* info, int nbo, T* B, int64_t sB, T* tmp,
                         hipStream_t stream);
// (flags of gpk_potrf_rows_rhs, as gpk.h defines them)
#ifndef GPK_ROWS_RHS
#define GPK_ROWS_RHS 1
#define GPK_ROWS_NO_TAIL_INVERSES 2
#define GPK_ROWS_RHS_STRIP 64
#endif
int64_t gpk_kmat_vjp_blocks_impl(int64_t n);
template <typename T>
int gpk_kmat_vjp_launch(const int* kinds, const double* inv_ls, int nterms, const T* X, int64_t n, int64_t ldx,
                        int d, const T* Kinv, int64_t ldk, const T* A, int C, int64_t lda, const double* g,
                        T* partial, T* diag_g, hipStream_t stream);
void gpk_kmat_vjp_dense_grid_impl(int64_t n, int64_t m, int64_t* rowtiles, int64_t* nchunks, int64_t* tiles_per_chunk);
template <typename T>
int gpk_kmat_vjp_dense_launch(const int* kinds, const double* variances, const double* inv_ls, int nterms,
                              const T* X, int64_t n, int64_t ldx, const T* Y, int64_t m, int64_t ldy, int d,
                              const T* G, int64_t ldg, const T* colscale, const T* w, const T* b, T* partial,
                              T* colsum, T* gradx, hipStream_t stream);
template <typename T>
int gpk_trtri_launch(const T* L, int64_t n, int64_t ld, const T* dinv_sb, int sb, T* W, int64_t ldw, T* tmp,
                     hipStream_t stream);
