/* gpk.h -- C ABI of libgpk.so: the MI355X (gfx950) kernels behind the dense
 * Gaussian-process inference hot path of wesselb/stheno.
 *
 * The reference is pure Python; its arithmetic for this path lives in the
 * un-vendored packages `mlkernels` (kernel evaluation), `matrix` and `lab`
 * (Cholesky / triangular solves / reductions, dispatched to LAPACK).  Each
 * entry point below replaces the op named in its comment at the reference
 * call site given as file:line (paths relative to the reference repository).
 * INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *  - All matrices are ROW-MAJOR with a leading dimension `ld*` in ELEMENTS.
 *  - `batch` independent problems are laid out with a stride `s*` in ELEMENTS
 *    (stheno's batched computation: README.md:744-766, random.py:261,274).
 *  - Every pointer named x/y/a/b/l/out/... is a DEVICE pointer owned by the
 *    caller; term descriptors (`kinds`, `variances`, `inv_ls`) are HOST arrays.
 *  - `dtype`: GPK_F32 or GPK_F64; all device buffers of a call share it.
 *  - `stream` is a hipStream_t passed as void*.  Calls only ENQUEUE work: no
 *    device allocation, no free, no host synchronisation, no retained pointers.
 *    Process-wide state is limited to: one helper stream (CU-masked to one CU per XCD) +
 *    events per device that gpk_potrf_la / gpk_init create on first use (fork/join around
 *    the caller's stream, serialised by a mutex; that first use allocates 36 bytes for a
 *    moment and synchronises once), the cached CU count per device, the tuning knobs of gpk_tune and the
 *    opt-in measurement hooks gpk_prof_*.  Everything else is thread-safe for distinct
 *    streams/buffers.
 *  - Return value: 0 on success; -k if argument k (1-based) is invalid;
 *    GPK_ERR_LAUNCH if a kernel launch failed.  No C++ exception crosses the ABI.
 */
#ifndef GPK_H
#define GPK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPK_F32 0
#define GPK_F64 1

#define GPK_OK 0
#define GPK_ERR_LAUNCH (-100)

/* kernel term kinds; one term = variance * kappa(dist(x, y) * inv_ls) */
#define GPK_K_EQ 0        /* exp(-r^2/2)                        mlkernels.EQ        */
#define GPK_K_MATERN12 1  /* exp(-r)                            mlkernels.Exp       */
#define GPK_K_MATERN32 2  /* (1+sqrt3 r) exp(-sqrt3 r)          mlkernels.Matern32  */
#define GPK_K_MATERN52 3  /* (1+sqrt5 r+5r^2/3) exp(-sqrt5 r)   mlkernels.Matern52  */
#define GPK_K_LINEAR 4    /* <x, y>                             mlkernels.Linear    */
#define GPK_K_CONST 5     /* 1                                  mlkernels.OneKernel */
#define GPK_MAX_TERMS 8

#define GPK_GEMM_LOWER 1
#define GPK_GEMM_TRI_K 2
#define GPK_GEMM_TRI_K_LOWER 4
#define GPK_GEMM_TRI_K_LOWER_B 8

#define GPK_DIAG_BLOCK 128 /* order of the diagonal blocks whose inverses gpk_potrf leaves in `dinv` */

int gpk_version(void);

/* Optional: create the per-device helper stream of gpk_potrf_la now (current device) instead of at its first
 * call -- it costs one stream creation, a 64-workgroup census kernel and ONE host synchronisation, which must
 * not happen inside a stream capture. */
int gpk_init(void);

/* Destroy the helper streams and events again (idempotent; they are re-created on demand).  Call it before the process
 * exits (the Python binding registers it with atexit): a CU-masked stream that is still alive at exit can crash the
 * teardown of the runtime / of a profiler wrapped around the process. */
void gpk_shutdown(void);

/* Kernel matrix  out[i][j] (+)= sum_t variances[t] * kappa_kinds[t](x_i, y_j; inv_ls[t])
 * and, if `symmetric` (x and y are the same points), + diag_add + diag_vec[i] on i == j.
 * Replaces mlkernels `pairwise` + `B.add(K, noise)`:  stheno/model/fdd.py:79,
 * stheno/model/observations.py:139 (K_x), :285-286 (K_zx, K_z).
 *   x: n x d (ldx), y: m x d (ldy), out: n x m (ld).  lower_only: skip tiles above the diagonal.
 *   diag_vec: nullable, n values per batch (stride s_diag).  accumulate: out += instead of out =.
 *   Numerics: squared distances from direct differences (no |x|^2 + |y|^2 - 2 x.y cancellation); exp and sqrt are the library's own
 *   branch-free device routines -- every value within ~2.5 eps (1 + |argument of the exponential|) of the exactly rounded one for the
 *   given inputs (measured element by element against an 80-bit reference: csrc/selftest.cpp, `kmat_* elementwise`); NaN inputs give
 *   NaN values; the distance of a point to itself is exactly 0 (sqrt(0) is evaluated as 1e-140 in fp64). */
int gpk_kmat(int dtype, const int* kinds, const double* variances, const double* inv_ls, int nterms,
             const void* x, int64_t n, int64_t ldx, int64_t sx, const void* y, int64_t m, int64_t ldy,
             int64_t sy, int d, void* out, int64_t ld, int64_t so, int64_t batch, int lower_only,
             int symmetric, double diag_add, const void* diag_vec, int64_t s_diag, int accumulate,
             void* stream);

/* Kernel diagonal  out[i] = k(x_i, x_i).  Replaces mlkernels `elwise`:
 * stheno/model/fdd.py:66, stheno/model/observations.py:304. */
int gpk_kdiag(int dtype, const int* kinds, const double* variances, const double* inv_ls, int nterms,
              const void* x, int64_t n, int64_t ldx, int64_t sx, int d, void* out, int64_t so,
              int64_t batch, void* stream);

/* Number of elements of the `dinv` workspace gpk_potrf needs per batch entry. */
int64_t gpk_dinv_elems(int64_t n);

/* In-place lower Cholesky  A = L L^T  (only the lower triangle is read and written; the
 * strict upper triangle is left untouched: use gpk_tril for a clean factor).  `dinv` receives inv(L_cc) of every 128x128
 * diagonal block ([batch][ceil(n/128)][128][128], identity-padded); `info`
 * (int per batch entry, MUST be zeroed by the caller) receives the LAPACK-style
 * order of the first non-positive pivot, 0 if none.  nbo: outer block of the right-looking
 * sweep (128 * 2^k; <= 0 selects the default -- one matrix: the whole matrix up to n = 4096, else 1024; batches: 1024 for
 * n >= 8192, 512 for n >= 2048, else 256).
 * ONE matrix (batch == 1) is factorised panel by panel with ONE launch per panel of nbo columns (+ a memset of its control words,
 * which live in a `dinv` slot the panel does not write): a chain workgroup factorises and inverts the diagonal blocks, the other
 * workgroups take the solves and updates of the panel as tasks and wait for each other through flag words in device memory
 * (gpk_potrf.hip, potrf_pipe_kernel; the task list is deadlock-free whatever part of the grid is resident).  A workgroup that
 * waited for seconds gives up and makes all others leave: `info` = -1 then (never observed; the factor is unusable).
 * BATCHES of at least 64 fp32 matrices whose order is a multiple of 128 (from 512 on; 16-byte aligned, on a stream without a CU
 * mask) take one launch for the diagonal blocks and ONE mixed-phase launch for everything below them per 128-column step
 * (gpk_potrf.hip, batch_mix_kernel: panel solves and update tiles of different matrices share the CUs; per-matrix counters in the
 * next `dinv` slot order them, without fences -- every task of a matrix runs on the XCD the matrix is pinned to).  That placement is
 * checked on the device: a violation is reported as `info` = -2 for one of the batch's matrices (never observed; the factors are
 * unusable).  Same arithmetic in the same order per entry as the lockstep launches other batches take: bit-identical factors.
 * Replaces `B.cholesky(B.reg(K))` (LAPACK potrf): implicit under B.logdet / B.iqf_diag at
 * stheno/random.py:274-276, explicit at stheno/model/observations.py:300. */
int gpk_potrf(int dtype, void* a, int64_t n, int64_t ld, int64_t sa, int64_t batch, void* dinv,
              int* info, int nbo, void* stream);

/* The same factorisation for ONE large matrix with LOOK-AHEAD: outer blocks of `nb` columns (256 ... 4096,
 * a power of two); the serial panel chain of outer step j+1 (diagonal-block factorisations) runs on a helper
 * stream, on CUs the persistent trailing-update kernel of step j leaves empty, and the rows below each diagonal
 * block are solved by ONE GEMM with the explicit inverse of the nb x nb diagonal block.  Results as gpk_potrf, plus
 * `dinv_nb` = [ceil(n/nb)][nb][nb], the inverses of the nb x nb diagonal blocks of L -- exactly what
 * gpk_trtri_merge(sb = nb) would produce, so the solves below can take them as they are.
 * `ws`: gpk_potrf_la_ws_elems(n, nb) elements of scratch.  `info` as for gpk_potrf (one int, zeroed by the caller).
 * The helper stream is forked from / joined to `stream` with events; under stream capture it joins the capture. */
/* gpk_potrf with ONE right-hand side per matrix solved along (round 6): `b` holds batch vectors of n entries (unit stride, sb elements
 * apart), overwritten by L^{-1} b -- what gpk_trsv_lower(l, ..., sb = 128, nrhs = 1) would compute behind the factorisation.
 * Batches that take the mixed-phase steps (fp32, >= 64 aligned matrices) run each 128-column step's share of the sweep on a side
 * stream as soon as that block column is final, beside the MFMA-bound launches of the next step: the sweep's HBM traffic (the factor
 * read once: 4.3 GB and 0.75 ms for 512 x 2048^2 fp32) hides under the factorisation.  Every other shape factorises, then sweeps.
 * `tmp`: batch * 128 + GPK_TRSV_CTRL_ELEMS elements; dinv as for gpk_potrf (required).
 * Replaces `B.cholesky` + the solve inside `B.iqf_diag(K, y - m)`: stheno/random.py:272-279 (batched: tests/model/test_cases.py:134-176). */
int gpk_potrf_rhs(int dtype, void* a, int64_t n, int64_t ld, int64_t sa, int64_t batch, void* dinv, int* info, int nbo, void* b, int64_t sb,
                  void* tmp, void* stream);

int64_t gpk_potrf_la_ws_elems(int64_t n, int nb);
int gpk_potrf_la(int dtype, void* a, int64_t n, int64_t ld, void* dinv, void* dinv_nb, int nb, void* ws, int* info,
                 void* stream);
/* The same with explicit inverses NARROWER than the outer blocks: `sb` (256 ... nb, a power of two) wide.  The rows below an nb-block
 * are then solved by block substitution over its nb / sb column blocks (same flops, 2 nb / sb - 1 GEMMs instead of one) and `dinv_sb` =
 * [ceil(n/sb)][sb][sb] is what gpk_trtri_merge(sb) would produce.  For fp32: the error of everything solved against an explicit inverse
 * grows with its width (posterior mean of configs[2] at N = 32768 against fp64: 1.0e-3 with 1024-wide inverses, 7e-4 with 512), while
 * the factorisation is fastest with 1024-column outer blocks. */
int gpk_potrf_la_split(int dtype, void* a, int64_t n, int64_t ld, void* dinv, void* dinv_sb, int nb, int sb, void* ws, int* info,
                       void* stream);

/* Factorisation WITH ROWS UNDER THE MATRIX (round 5): `a` holds `rows` >= n rows of n columns -- the symmetric matrix (lower
 * triangle read) in the first n, anything else below, typically K(x*, x) of the posterior.  The extra rows are carried through
 * the factorisation like the rows below a diagonal block: every panel solve and every trailing update includes them, and they
 * come out as  a[n:, :] L^{-T}  -- the TRANSPOSE of the whitened cross-covariance L^{-1} K(x, x*) that `gpk_trsm_lower` would
 * have to compute afterwards.  Their flops ride in the factorisation's own GEMM launches and, in its chain-bound tail, on
 * workgroups that would otherwise wait: the separate many-column solve (and its ~30 launches) disappears from the posterior path.
 * nb = 0: pipelined panels (any n up to the look-ahead threshold; dinv_sb / ws unused);  nb > 0: look-ahead as gpk_potrf_la /
 * gpk_potrf_la_split (sb = 0: sb = nb), `ws`: gpk_potrf_la_ws_elems(rows, nb).
 * n must be a multiple of 128 when rows > n; `dinv`: gpk_dinv_elems(n) + 128 * 128 elements (one slot more: control words).
 * Replaces `B.cholesky` + the `B.solve(L, K_zx)` of mlkernels' PosteriorKernel / PosteriorMean (observations.py:148-168) in one call. */
int gpk_potrf_rows(int dtype, void* a, int64_t n, int64_t rows, int64_t ld, void* dinv, void* dinv_sb, int nb, int sb, void* ws, int* info,
                   void* stream);

/* The same with flags (round 6):
 *   GPK_ROWS_RHS               the last GPK_ROWS_RHS_STRIP (64) rows are a strip whose FIRST row, a[rows - 64][0 .. n), is one right-hand
 *                              side b (contiguous); the other 63 are padding (initialise them -- zeros; unspecified on return): whole
 *                              64-row strips keep the plain tail on its fast kernels.  b comes out as L^{-1} b like every
 *                              other row does, but through the look-ahead steps it is treated as the vector it is -- the two matrix-vector
 *                              products per diagonal block of gpk_trsv_lower, enqueued as soon as a panel is final on a third stream (CU mask of the helper stream), so
 *                              that they run beside the trailing updates instead of as ~30 dependent launches behind the factorisation;
 *                              the plain tail (and the pipelined panels, nb = 0) carry it as a row.  The observations y - m(x) of the
 *                              posterior mean / log-density: the single-column solve disappears from the step.
 *   GPK_ROWS_NO_TAIL_INVERSES  the sb-wide explicit inverses of the plain tail's diagonal blocks (the last <= 6144 columns) are NOT
 *                              written to dinv_sb (nobody is going to solve against this factor with them; gpk_trtri_merge computes
 *                              them from `dinv` on demand).  The look-ahead panels' own inverses are always there.
 * Replaces `B.cholesky` + `B.solve(L, K_zx)` + the solve inside `B.iqf_diag(K, y - m)`: stheno/model/observations.py:148-168,
 * stheno/random.py:272-279. */
#define GPK_ROWS_RHS 1
#define GPK_ROWS_NO_TAIL_INVERSES 2
#define GPK_ROWS_RHS_STRIP 64
int gpk_potrf_rows_rhs(int dtype, void* a, int64_t n, int64_t rows, int64_t ld, void* dinv, void* dinv_sb, int nb, int sb, void* ws, int* info,
                       int flags, void* stream);

/* Merge the 128-block inverses into inverses of sb x sb diagonal blocks
 * (sb = 128 * 2^k <= 4096); dinv_sb: [batch][ceil(n/sb)][sb][sb];
 * tmp: >= ceil(n/sb) * sb * sb / 4 elements.  Part of the blocked TRSM below. */
int gpk_trtri_merge(int dtype, const void* l, int64_t n, int64_t ld, int64_t sl, int64_t batch,
                    const void* dinv128, int sb, void* dinv_sb, void* tmp, void* stream);

/* B <- L^{-1} B, many right-hand sides: recursive blocked solve on the MFMA GEMM (half of the flops in ONE GEMM with K = n/2, a
 * quarter in two with K = n/4, ...; the sb x sb diagonal blocks by their merged inverses).  tmp: batch * sb * nrhs elements.
 * Replaces `B.solve(L, .)` / the solves inside `B.iqf` (LAPACK trsm):
 * stheno/model/observations.py:301,322,327,329; mlkernels.PosteriorMean/PosteriorKernel
 * constructed at observations.py:148-168. */
int gpk_trsm_lower(int dtype, const void* l, int64_t n, int64_t ld, int64_t sl, const void* dinv_sb,
                   int sb, void* b, int64_t nrhs, int64_t ldb, int64_t sb_stride, void* tmp,
                   int64_t batch, void* stream);

/* The same solve OUT OF PLACE: x = L^{-1} b into a second n x nrhs buffer (ldx >= nrhs), `b` is used up as workspace.  Saves the
 * copy of every solved block (the in-place form builds it in `tmp` first): the form the posterior path uses for its
 * 2048-column solve.  Replaces the same reference call sites as gpk_trsm_lower. */
int gpk_trsm_lower_to(int dtype, const void* l, int64_t n, int64_t ld, int64_t sl, const void* dinv_sb,
                      int sb, void* b, int64_t nrhs, int64_t ldb, int64_t sb_stride, void* x, int64_t ldx,
                      int64_t sx_stride, int64_t batch, void* stream);

/* B <- L^{-1} B, nrhs <= 8 (GEMV sweep, HBM-bound).  tmp: batch * sb * nrhs + GPK_TRSV_CTRL_ELEMS elements (16-byte aligned;
 * the extra elements hold the control words of the single-launch sweep -- one right-hand side of one factor runs as ONE resident
 * launch whose workgroups meet at grid barriers; the call zeroes them).
 * Replaces the solve inside `B.iqf_diag(var, y - mean)`: stheno/random.py:276,
 * stheno/model/observations.py:335. */
#define GPK_TRSV_CTRL_ELEMS 16
int gpk_trsv_lower(int dtype, const void* l, int64_t n, int64_t ld, int64_t sl, const void* dinv_sb,
                   int sb, void* b, int nrhs, int64_t ldb, int64_t sb_stride, void* tmp, int64_t batch,
                   void* stream);

/* C = alpha * op(A) op(B)^T-style contraction + beta * C on MFMA:
 *   C[m][n] = alpha * sum_k a(m,k) b(n,k) + beta * C[m][n]
 * a_kmajor != 0: A stored M x K (k contiguous); else stored K x M.  Same for B (N x K / K x N).
 * flags: GPK_GEMM_LOWER = only tiles on/below the diagonal (SYRK); GPK_GEMM_TRI_K = both operands
 * vanish for k < their row index (lower-triangular factors stored K x M, e.g. W^T W with W = L^{-1}):
 * all-zero k-chunks are skipped; GPK_GEMM_TRI_K_LOWER = A (M x K) is lower triangular (a(m,k) = 0 for
 * k > m, e.g. an inverted diagonal block times right-hand sides): the k loop stops at the tile's last row;
 * GPK_GEMM_TRI_K_LOWER_B = the same for B (b(n,k) = 0 for k > n: right-hand sides times an inverted block, transposed).
 * Replaces the dense products
 * `B.mm` / `B.matmul` / `B.iqf` outer products: stheno/model/observations.py:322-323,
 * mlkernels.PosteriorKernel (full covariance), `B.sample` (L xi): stheno/random.py:351. */
int gpk_gemm(int dtype, int a_kmajor, int b_kmajor, int64_t m, int64_t n, int64_t k, double alpha,
             const void* a, int64_t lda, int64_t sa, const void* b, int64_t ldb, int64_t sb, double beta,
             void* c, int64_t ldc, int64_t sc, int64_t batch, int flags, void* stream);

/* The same contraction with a column scaling and column statistics folded into the store (unbatched, beta = 0, no GPK_GEMM_LOWER):
 *   c[m][n] = alpha * sum_k a(m,k) b(n,k) * colscale[n]          (colscale NULL: no scaling)
 *   colss[r][n] = sum over the 64 rows of slab r of (alpha * sum_k a(m,k) b(n,k))^2      (colss NULL: not computed)
 * colss: gpk_gemm_colss_rows(m) rows of ldss >= n elements; the column sums of squares of the UNSCALED product are the sums of its
 * rows.  What SURVEY 8(b) proposed as `gpk_syrk_scaled`, cut where the pseudo-point path needs it: V = L_z^{-1} K_zx leaves the
 * kernel as V K_n^{-1/2}, and Q_x_diag = column sums of squares of V (`B.matmul_diag`) is a by-product -- the stand-alone scaling
 * and reduction passes over the M x N matrix are gone.  Replaces `B.solve` + `B.matmul_diag` + the K_n^{-1/2} scalings at
 * stheno/model/observations.py:301, 305, 322, 327. */
int64_t gpk_gemm_colss_rows(int64_t m);
int gpk_gemm_colscale(int dtype, int a_kmajor, int b_kmajor, int64_t m, int64_t n, int64_t k, double alpha,
                      const void* a, int64_t lda, const void* b, int64_t ldb, void* c, int64_t ldc, int flags,
                      const void* colscale, void* colss, int64_t ldss, void* stream);

/* Up to two rank-k updates in ONE persistent launch (a resident set of workgroups pulls 128x128 tiles of
 * both problems from a device-side counter: one ramp, one tail):
 *   c[m][n] = cin[m][n] + alpha * sum_k a[m][k] b[n][k]        a: m x k, b: n x k, both row-major
 * `cin` may differ from `c` (out-of-place update).  lower_only: tiles on/below the diagonal only.
 * ctrl: 256 bytes of device scratch (zeroed by the call).  reserve_cus != 0: the kernel leaves one CU per XCD
 * empty while it runs, for work enqueued on another stream (what gpk_potrf_la does for its panel chain).
 * This is the trailing update of gpk_potrf_la, exposed for measurement and reuse. */
typedef struct {
    int64_t m, n, k;
    const void* a; int64_t lda;
    const void* b; int64_t ldb;
    const void* cin; int64_t ldcin;
    void* c; int64_t ldc;
    int lower_only;
} gpk_update_t;
int gpk_gemm_update2(int dtype, const gpk_update_t* upd, int nupd, double alpha, void* ctrl, int reserve_cus,
                     void* stream);

/* out[b] = 2 * sum_i log L[i][i].  Replaces `B.logdet`: stheno/random.py:274,
 * stheno/model/observations.py:334. */
int gpk_logdet_chol(int dtype, const void* l, int64_t n, int64_t ld, int64_t sl, int64_t batch,
                    void* out, void* stream);

/* Number of row chunks (workspace sizing) used by gpk_colreduce for `rows` rows. */
int64_t gpk_colreduce_chunks(int64_t rows);

/* Row reductions of Z (rows x n, row-major, ld >= n):  out_dot[i] = sum_k Z[i][k] w[k]  (skipped if out_dot is NULL),
 * out_ss[i] = sum_k Z[i][k]^2  (skipped if out_ss is NULL), one pass.  The posterior mean and marginal variance from the rows
 * gpk_potrf_rows leaves under the factor (Z = K(x*, x) L^{-T}): the same quantities as gpk_colreduce on L^{-1} K(x, x*).
 * Replaces the reductions inside mlkernels' PosteriorMean / PosteriorKernel.elwise (observations.py:148-168). */
int gpk_rowreduce(int dtype, const void* z, int64_t rows, int64_t n, int64_t ld, const void* w, void* out_dot, void* out_ss, void* stream);

/* Fused column reductions of V (rows x cols):
 *   out_dot[j] = sum_i V[i][j] w[i]   (skipped if out_dot or w is NULL)
 *   out_ss[j]  = sum_i V[i][j]^2      (skipped if out_ss is NULL)
 * ws: 2 * batch * gpk_colreduce_chunks(rows) * cols elements.  Deterministic (two-stage).
 * Replaces `B.iqf_diag` / `B.matmul_diag(V, V, tr_a=True)` (random.py:276, observations.py:305)
 * and the mean contraction of mlkernels.PosteriorMean. */
int gpk_colreduce(int dtype, const void* v, int64_t rows, int64_t cols, int64_t ld, int64_t sv,
                  const void* w, int64_t sw, void* out_dot, void* out_ss, void* ws, int64_t batch,
                  void* stream);

/* out[i][j] = sum_s parts[s][i][j] for j <= i: the partial products of a SPLIT-K symmetric update (`nparts` n x n matrices, `sp`
 * elements apart, of which only the lower tiles were written -- gpk_gemm with GPK_GEMM_LOWER over a batch of k-slices) added up in a
 * fixed order, lower triangle only (entries above the diagonal of `out` are unspecified).  The pseudo-point path's
 * `V K_n^{-1} V^T` (observations.py:322) with its 200 000-long contraction cut into 25 slices: replaces a zero fill of the 25 partial
 * matrices and a torch reduction over all of them. */
int gpk_sum_lower(int dtype, const void* parts, int64_t nparts, int64_t n, int64_t ldp, int64_t sp, void* out, int64_t ldo, void* stream);

/* Zero the strict upper triangle (clean Cholesky factor for `B.cholesky` consumers). */
int gpk_tril(int dtype, void* a, int64_t n, int64_t ld, int64_t sa, int64_t batch, void* stream);

/* A[i][i] += s + (v ? v[i] : 0):  `B.add(var, Diagonal)` / `B.reg`: stheno/model/fdd.py:79. */
int gpk_add_diag(int dtype, void* a, int64_t n, int64_t ld, int64_t sa, double s, const void* v,
                 int64_t sv, int64_t batch, void* stream);

/* V[i][j] *= s[j]  (column scaling by K_n^{-1/2} in the pseudo-point ELBO: the
 * `B.iqf(K_n, .)` with diagonal K_n of stheno/model/observations.py:322,327). */
int gpk_scale_cols(int dtype, void* v, int64_t rows, int64_t cols, int64_t ld, int64_t sv, const void* s,
                   int64_t ss, int64_t batch, void* stream);

/* Mirror the lower triangle into the upper one (full symmetric matrix after a lower-only SYRK/kmat). */
int gpk_symmetrize(int dtype, void* a, int64_t n, int64_t ld, int64_t sa, int64_t batch, void* stream);

/* y[m x nrhs] = alpha * A[m x k] x[k x nrhs] + beta * y, nrhs <= 8, A row-major; trans must be 0.
 * HBM-bound matrix-vector products of the path (e.g. `B.iqf(K_n, V^T, y)`: observations.py:327). */
int gpk_gemv(int dtype, int trans, int64_t m, int64_t k, int nrhs, double alpha, const void* a, int64_t lda,
             int64_t sa, const void* x, int64_t ldx, int64_t sx, double beta, void* y, int64_t ldy, int64_t sy,
             int64_t batch, void* stream);

/* W <- L^{-1}: the full lower-triangular inverse (n x n, ldw), N^3/3 flops on the MFMA GEMM.
 * dinv_sb as for gpk_trsm_lower (unbatched); tmp: sb * n elements.  With gpk_gemm(K x M storage,
 * GPK_GEMM_LOWER | GPK_GEMM_TRI_K) it gives K^{-1} = W^T W for the log-density gradient. */
int gpk_trtri_lower(int dtype, const void* l, int64_t n, int64_t ld, const void* dinv_sb, int sb, void* w,
                    int64_t ldw, void* tmp, void* stream);

/* Kernel-hyperparameter VJP of the GP log-density (hyper-parameter learning through
 * `f(x, noise).logpdf(y)`: readme_example13_optimisation_torch.py:47-53).  With
 *   G = d logpdf / dK = 1/2 (A diag(g) A^T - sum(g) K^{-1}),  A = K^{-1} (y - m)  (n x ncols <= 8),
 * one pass over the LOWER triangle of `kinv` accumulates per workgroup
 *   partial[b][2t] = sum_ij G_ij kappa_t,  partial[b][2t+1] = sum_ij G_ij kappa_t'(q) q,
 *   partial[b][2*GPK_MAX_TERMS] = trace(G)   (row stride 2*GPK_MAX_TERMS + 1; sum over b),
 * and writes diag_g[i] = G_ii.  d logpdf/d variance_t = S1_t; d logpdf/d scale_t = -2 v_t S2_t / l_t;
 * d logpdf / d noise_i = G_ii.  `g` (host): upstream gradient per column. */
int64_t gpk_kmat_vjp_blocks(int64_t n);
int gpk_kmat_vjp(int dtype, const int* kinds, const double* inv_ls, int nterms, const void* x, int64_t n,
                 int64_t ldx, int d, const void* kinv, int64_t ldk, const void* alpha, int ncols, int64_t lda,
                 const double* g, void* partial, void* diag_g, void* stream);

/* Kernel VJP with an explicit cotangent (gradient of the pseudo-point ELBO w.r.t. kernel
 * hyper-parameters, observation noise and inducing inputs: the learning loop that
 * readme_example10_sparse.py:8-29 / stheno/model/observations.py:279-336 feed when `stheno.torch`
 * tensors carry gradients).  For K = k(x, y) (n x m, never formed) and
 *   Geff_ij = g[i][j] * colscale[j] + w[i] * b[j]      (colscale and the pair (w, b) may be NULL),
 * one pass over g writes, per workgroup (rowtiles x nchunks of them, see gpk_kmat_vjp_dense_grid),
 *   partial[wg][2t] = sum Geff kappa_t,  partial[wg][2t+1] = sum Geff kappa_t'(q) q   (row stride
 *   2*GPK_MAX_TERMS + 1; sum over wg; d/dvariance_t = S1_t, d/dscale_t = -2 v_t S2_t / l_t),
 *   colsum[rowtile][j] = sum_{i in tile} Geff_ij K_ij        (NULL to skip; sum over row tiles),
 *   gradx[chunk][i][0..d) = sum_{j in chunk} Geff_ij dK_ij/dx_i   (NULL to skip; needs d <= 8; sum over chunks).
 * Deterministic (no atomics). */
int gpk_kmat_vjp_dense_grid(int64_t n, int64_t m, int64_t* rowtiles, int64_t* nchunks);
int gpk_kmat_vjp_dense(int dtype, const int* kinds, const double* variances, const double* inv_ls, int nterms,
                       const void* x, int64_t n, int64_t ldx, const void* y, int64_t m, int64_t ldy, int d,
                       const void* g, int64_t ldg, const void* colscale, const void* w, const void* b,
                       void* partial, void* colsum, void* gradx, void* stream);

/* Measurement hooks (bench.py's live roofline figure).  Between gpk_prof_start and
 * gpk_prof_stop every MFMA GEMM launch of this process is bracketed by HIP events on its
 * launch stream; gpk_prof_stop synchronises those events and returns the summed duration,
 * launch count and ALGORITHMIC flops (2mnk; mnk for a lower-only symmetric update) of the
 * launches of one kernel variant: 16*(64x64-tile kernel) + 8*(f64) + 4*(a_kmajor) + 2*(b_kmajor)
 * + 1*(bounds-checked kernel); + 32 = the persistent update (gemm_persist_kernel), 64 + ... = panel_step_kernel,
 * 96 + ... = gemm_trilo_pair_kernel, 128 + ... = gemm_trib_kernel, 160 (+ 8: f64) = batch_mix_kernel (one mixed-phase step of a batched
 * factorisation: its panel solves at the TRSM count + its update tiles at the symmetric count); or -1 for all.  Not thread-safe; off by default; not used
 * by the product path. */
int gpk_prof_start(void);
int gpk_prof_stop(int variant, double* total_ms, int64_t* launches, double* useful_flops);

/* The SUSTAINED rate of the matrix pipes, measured on the device the call runs on ("re-derive on the box from a measured
 * MFMA micro-benchmark and print the value used"): one workgroup per CU, `waves_per_simd` (1 or 2) register-resident waves per SIMD
 * streaming v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 (the instructions every GEMM of this library issues) on pseudo-random
 * operands into 8 independent accumulators, launches repeated until `min_ms` of device time have passed.  Out: *tflops = flops / HIP-event
 * time; *ms = that time; *clock_mhz = the shader clock the stream ran at (s_memtime cycles over the 100 MHz wall clock, last launch,
 * mean over CUs); *issue_eff = tflops / (CUs x clock x 128 (f64) / 256 (f32) flops per CU and cycle) -- the last three may be NULL.
 * A measurement hook like gpk_prof_*: allocates its own 4 KiB of scratch, SYNCHRONISES the stream, not on the product path. */
int gpk_mfma_peak(int dtype, double min_ms, int waves_per_simd, double* tflops, double* ms, double* clock_mhz, double* issue_eff, void* stream);

/* Development aids.  The RELEASE library (stheno_amd/csrc/libgpk.so) has no tuning knobs: every one of them is a compile-time
 * constant at its measured optimum and gpk_tune() does nothing.  The dev build (stheno_amd/csrc/dev/libgpk.so, -DGPK_DEV_KNOBS;
 * what the native self-test links against and what GPK_DEV=1 makes the Python binding load) keeps them mutable for A/B runs
 * (`gpk_selftest --set KEY VALUE`).  key 1: use 64x64 GEMM tiles below this many 128-tiles; 6: look-ahead overlaps while the
 * trailing matrix has at least this many rows; 7: 0 = look-ahead algorithm on one stream, 1 = with the helper stream; 8: the
 * persistent update takes 64x64 tiles below this many 128-tiles; 9: gpk_potrf_la finishes the last this-many rows with the plain
 * algorithm (0 = 6144); 10: panel GEMM of gpk_potrf_la as 0 = plain launch, 1 / 2 = persistent (paired tiles); 11: strip written
 * last; 12: 1 = row-band kernel-matrix kernel, 0 = one tile per workgroup; 17: 1 = one-workgroup-per-matrix TRSV for batches of
 * small factors; 18: 1 = the CUs reserved for the look-ahead chain rejoin the trailing update once the chain is done;
 * 20: gpk_tune_tile_prof stamps only the v-th persistent launch since this knob was set (-1 = every launch); 31: quarter tiles for
 * the last partial round of a 128-tile GEMM launch; 32: fused panel-step kernel (batched / fallback path); 34: compact 1-D grid for
 * lower-triangle kernel matrices; 36: triangular-operand fragment skipping in panel solves; 37: 1 = one pipelined launch per panel
 * for single matrices, 0 = diagonal-block kernel + panel-step kernel per 128 columns (the batched path's steps); 38: the rest of a
 * panel's trailing update rides in the next panel's launch; 39: workgroups that take panel tasks first in such a launch (0 = a
 * third of the CUs); 40 / 41: the look-ahead's update of the next diagonal block is the first segment of the trailing update while
 * that has at least (40) rows and the outer block is at most (41) wide; 42: small products with a lower-triangular A (the leaves of
 * the recursive solve) as pairs of 32-row tiles with equal K per workgroup; 45: batched 128-tile GEMM launches as a 1-D grid with all
 * tiles of a matrix on one XCD; 53: batched factorisations take the mixed-phase steps (0 = lockstep launches, 1 = fp32, 2 = fp64 too);
 * 54: ... from this many matrices on; 55: tasks between a matrix's solves and its update tiles in the queue order; 56: bit 0 = update
 * tiles do not pull their C tile into the L2 before they wait, bit 2 = solve tiles publish behind an agent-scope release; 57: fp32
 * batches of more matrices than CUs take the diagonal-block kernel compiled for two workgroups per CU.  (Removed in round 4 with the code they selected: 2 / 4 /
 * 13 -- XCD super-tile, row-pair and column-major tile orders -- and 30, the 256-thread diagonal-block kernel of rounds 1-2.)
 * gpk_tune_diag_prof: device buffer (32 int64 per diagonal block, or NULL) for cycle / wall-clock stamps of the diagonal-block
 * kernel and of the pipelined panel's chain and critical tasks (read by `gpk_selftest --diagprof`). */
void gpk_tune(int key, int64_t value);
void gpk_tune_diag_prof(long long* dev_buf);
/* device buffer (grid x 8 tiles x 8 int64, or NULL): stamps of the persistent update's first 8 tiles per workgroup:
 * [0..3] wall clock (10 ns ticks) at entry / C tile requested / k loop done / stores retired, [4], [5] shader-cycle
 * counter at the start and the end of the k loop (so the clock the loop ran at can be read off), [6] wall clock when the C values
 * and the first operand chunk have arrived (pipelined 128-tile), [7] unused. */
void gpk_tune_tile_prof(long long* dev_buf);

/* Strided 2-D copy (rows x cols). */
int gpk_copy2d(int dtype, const void* src, int64_t lds, int64_t ss, void* dst, int64_t ldd, int64_t sd,
               int64_t rows, int64_t cols, int64_t batch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPK_H */
