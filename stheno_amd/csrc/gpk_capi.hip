// gpk_capi.hip -- the extern "C" boundary declared in include/gpk.h: dtype
// dispatch onto the templated launchers.  No torch types, no exceptions.
#include "gpk_common.hpp"
#include "../../include/gpk.h"

#define DISPATCH(dtype, CALL_F32, CALL_F64)  \
    do {                                     \
        if ((dtype) == GPK_F32) {            \
            typedef float T;                 \
            return CALL_F32;                 \
        } else if ((dtype) == GPK_F64) {     \
            typedef double T;                \
            return CALL_F64;                 \
        }                                    \
        return -1;                           \
    } while (0)

#define D1(dtype, EXPR) DISPATCH(dtype, EXPR, EXPR)

template <typename T>
static int update2(const gpk_update_t* upd, int nupd, double alpha, void* ctrl, int reserve, hipStream_t stream) {
    if (upd == nullptr || nupd < 1 || nupd > 2) return GPK_ERR_ARG(3);
    GpkSeg<T> seg[2];
    for (int i = 0; i < nupd; ++i)
        seg[i] = GpkSeg<T>{upd[i].m, upd[i].n, upd[i].k, (const T*)upd[i].a, upd[i].lda, (const T*)upd[i].b, upd[i].ldb,
                           (const T*)upd[i].cin, upd[i].ldcin, (T*)upd[i].c, upd[i].ldc, upd[i].lower_only, 0};
    return gpk_gemm_persist_launch<T>(seg, nupd, (T)alpha, (unsigned*)ctrl, reserve, stream);
}

template <typename T>
static int potrf_rows_any(T* a, int64_t n, int64_t rows, int64_t ld, T* dinv, T* dinv_sb, int nb, int sb, T* ws, int* info, int flags, hipStream_t stream) {
    if (rows < n) return GPK_ERR_ARG(4);
    if (n % GPK_DB != 0 && rows > n) return GPK_ERR_ARG(3);
    if (flags & ~(GPK_ROWS_RHS | GPK_ROWS_NO_TAIL_INVERSES)) return GPK_ERR_ARG(12);
    if ((flags & GPK_ROWS_RHS) && rows < n + GPK_ROWS_RHS_STRIP) return GPK_ERR_ARG(4);
    if (nb > 0) return gpk_potrf_la_launch<T>(a, n, ld, dinv, dinv_sb, nb, ws, info, stream, sb, rows, flags);
    return gpk_potrf_rows_launch<T>(a, n, rows, ld, dinv, info, stream);       // (pipelined panels: the right-hand side is a row like the others)
}

extern "C" {

int gpk_version(void) { return 100; }

int64_t gpk_colreduce_chunks(int64_t rows) { return gpk_colreduce_nchunks_impl(rows); }

int64_t gpk_dinv_elems(int64_t n) { return gpk_cdiv(n > 0 ? n : 1, GPK_DB) * GPK_DB * GPK_DB; }

int gpk_kmat(int dtype, const int* kinds, const double* variances, const double* inv_ls, int nterms,
             const void* x, int64_t n, int64_t ldx, int64_t sx, const void* y, int64_t m, int64_t ldy,
             int64_t sy, int d, void* out, int64_t ld, int64_t so, int64_t batch, int lower_only,
             int symmetric, double diag_add, const void* diag_vec, int64_t s_diag, int accumulate,
             void* stream) {
    D1(dtype, gpk_kmat_launch<T>(kinds, variances, inv_ls, nterms, (const T*)x, n, ldx, sx, (const T*)y, m,
                                 ldy, sy, d, (T*)out, ld, so, batch, lower_only, symmetric, diag_add,
                                 (const T*)diag_vec, s_diag, accumulate, (hipStream_t)stream));
}

int gpk_kdiag(int dtype, const int* kinds, const double* variances, const double* inv_ls, int nterms,
              const void* x, int64_t n, int64_t ldx, int64_t sx, int d, void* out, int64_t so,
              int64_t batch, void* stream) {
    D1(dtype, gpk_kdiag_launch<T>(kinds, variances, inv_ls, nterms, (const T*)x, n, ldx, sx, d, (T*)out, so,
                                  batch, (hipStream_t)stream));
}

int gpk_potrf(int dtype, void* a, int64_t n, int64_t ld, int64_t sa, int64_t batch, void* dinv,
              int* info, int nbo, void* stream) {
    D1(dtype, gpk_potrf_launch<T>((T*)a, n, ld, batch, sa, (T*)dinv, info, nbo, (hipStream_t)stream));
}

int gpk_potrf_rhs(int dtype, void* a, int64_t n, int64_t ld, int64_t sa, int64_t batch, void* dinv, int* info, int nbo, void* b, int64_t sb,
                  void* tmp, void* stream) {
    D1(dtype, gpk_potrf_rhs_launch<T>((T*)a, n, ld, batch, sa, (T*)dinv, info, nbo, (T*)b, sb, (T*)tmp, (hipStream_t)stream));
}

int64_t gpk_potrf_la_ws_elems(int64_t n, int nb) { return gpk_potrf_la_ws_elems_impl(n, nb); }

int gpk_potrf_la(int dtype, void* a, int64_t n, int64_t ld, void* dinv, void* dinv_nb, int nb, void* ws, int* info,
                 void* stream) {
    D1(dtype, gpk_potrf_la_launch<T>((T*)a, n, ld, (T*)dinv, (T*)dinv_nb, nb, (T*)ws, info, (hipStream_t)stream));
}

int gpk_potrf_rows(int dtype, void* a, int64_t n, int64_t rows, int64_t ld, void* dinv, void* dinv_sb, int nb, int sb, void* ws, int* info,
                   void* stream) {
    D1(dtype, potrf_rows_any<T>((T*)a, n, rows, ld, (T*)dinv, (T*)dinv_sb, nb, sb, (T*)ws, info, 0, (hipStream_t)stream));
}

int gpk_potrf_rows_rhs(int dtype, void* a, int64_t n, int64_t rows, int64_t ld, void* dinv, void* dinv_sb, int nb, int sb, void* ws, int* info,
                       int flags, void* stream) {
    D1(dtype, potrf_rows_any<T>((T*)a, n, rows, ld, (T*)dinv, (T*)dinv_sb, nb, sb, (T*)ws, info, flags, (hipStream_t)stream));
}

int gpk_potrf_la_split(int dtype, void* a, int64_t n, int64_t ld, void* dinv, void* dinv_sb, int nb, int sb, void* ws, int* info,
                       void* stream) {
    D1(dtype, gpk_potrf_la_launch<T>((T*)a, n, ld, (T*)dinv, (T*)dinv_sb, nb, (T*)ws, info, (hipStream_t)stream, sb));
}

int gpk_gemm_update2(int dtype, const gpk_update_t* upd, int nupd, double alpha, void* ctrl, int reserve_cus,
                     void* stream) {
    D1(dtype, update2<T>(upd, nupd, alpha, ctrl, reserve_cus, (hipStream_t)stream));
}

int gpk_init(void) {
    hipStream_t aux;
    unsigned keys[8];
    return gpk_helper_stream(&aux, keys);
}

void gpk_shutdown(void) {
    gpk_potrf_shutdown();
    gpk_helper_shutdown();
}

void gpk_tune(int key, int64_t value) {
    gpk_tune_gemm(key, value);
    gpk_tune_potrf(key, value);
    gpk_tune_kmat(key, value);
    gpk_tune_solve(key, value);
}

void gpk_tune_diag_prof(long long* dev_buf) { gpk_set_diag_prof(dev_buf); }

void gpk_tune_tile_prof(long long* dev_buf) { gpk_set_tile_prof(dev_buf); }

int gpk_trtri_merge(int dtype, const void* l, int64_t n, int64_t ld, int64_t sl, int64_t batch,
                    const void* dinv128, int sb, void* dinv_sb, void* tmp, void* stream) {
    D1(dtype, gpk_trtri_merge_launch<T>((const T*)l, n, ld, batch, sl, (const T*)dinv128, sb, (T*)dinv_sb,
                                        (T*)tmp, (hipStream_t)stream));
}

int gpk_trsm_lower(int dtype, const void* l, int64_t n, int64_t ld, int64_t sl, const void* dinv_sb,
                   int sb, void* b, int64_t nrhs, int64_t ldb, int64_t sb_stride, void* tmp,
                   int64_t batch, void* stream) {
    D1(dtype, gpk_trsm_launch<T>((const T*)l, n, ld, sl, (const T*)dinv_sb, sb, (T*)b, nrhs, ldb, sb_stride,
                                 (T*)tmp, batch, (hipStream_t)stream));
}

int gpk_trsm_lower_to(int dtype, const void* l, int64_t n, int64_t ld, int64_t sl, const void* dinv_sb,
                      int sb, void* b, int64_t nrhs, int64_t ldb, int64_t sb_stride, void* x, int64_t ldx, int64_t sx_stride,
                      int64_t batch, void* stream) {
    if (x == nullptr) return GPK_ERR_ARG(12);
    D1(dtype, gpk_trsm_launch<T>((const T*)l, n, ld, sl, (const T*)dinv_sb, sb, (T*)b, nrhs, ldb, sb_stride,
                                 (T*)nullptr, batch, (hipStream_t)stream, (T*)x, ldx, sx_stride));
}

int gpk_trsv_lower(int dtype, const void* l, int64_t n, int64_t ld, int64_t sl, const void* dinv_sb,
                   int sb, void* b, int nrhs, int64_t ldb, int64_t sb_stride, void* tmp, int64_t batch,
                   void* stream) {
    D1(dtype, gpk_trsv_launch<T>((const T*)l, n, ld, sl, (const T*)dinv_sb, sb, (T*)b, nrhs, ldb, sb_stride,
                                 (T*)tmp, batch, (hipStream_t)stream));
}

int gpk_gemm(int dtype, int a_kmajor, int b_kmajor, int64_t m, int64_t n, int64_t k, double alpha,
             const void* a, int64_t lda, int64_t sa, const void* b, int64_t ldb, int64_t sb, double beta,
             void* c, int64_t ldc, int64_t sc, int64_t batch, int flags, void* stream) {
    D1(dtype, gpk_gemm_launch<T>(a_kmajor != 0, b_kmajor != 0, m, n, k, (T)alpha, (const T*)a, lda, sa,
                                 (const T*)b, ldb, sb, (T)beta, (T*)c, ldc, sc, batch, flags,
                                 (hipStream_t)stream));
}

int64_t gpk_gemm_colss_rows(int64_t m) { return 2 * gpk_cdiv(m, GPK_TILE); }

int gpk_gemm_colscale(int dtype, int a_kmajor, int b_kmajor, int64_t m, int64_t n, int64_t k, double alpha,
                      const void* a, int64_t lda, const void* b, int64_t ldb, void* c, int64_t ldc, int flags,
                      const void* colscale, void* colss, int64_t ldss, void* stream) {
    if (colss != nullptr && ldss < n) return GPK_ERR_ARG(17);
    // the fused epilogue lives in the plain 128-tile kernel only: bit 0 (lower-only) and bit 4 (the panel-solve kernel) have no such epilogue
    if ((flags & ~(2 | 4 | 8)) != 0) return GPK_ERR_ARG(13);
    D1(dtype, gpk_gemm_launch2<T>(a_kmajor != 0, b_kmajor != 0, m, n, k, (T)alpha, (const T*)a, lda, 0, 0, (const T*)b, ldb, 0, 0,
                                  (T)0, (T*)c, ldc, 0, 0, 1, 1, flags, (hipStream_t)stream, (const T*)colscale, (T*)colss, ldss));
}

int gpk_logdet_chol(int dtype, const void* l, int64_t n, int64_t ld, int64_t sl, int64_t batch,
                    void* out, void* stream) {
    D1(dtype, gpk_logdet_launch<T>((const T*)l, n, ld, sl, batch, (T*)out, (hipStream_t)stream));
}

int gpk_colreduce(int dtype, const void* v, int64_t rows, int64_t cols, int64_t ld, int64_t sv,
                  const void* w, int64_t sw, void* out_dot, void* out_ss, void* ws, int64_t batch,
                  void* stream) {
    D1(dtype, gpk_colreduce_launch<T>((const T*)v, rows, cols, ld, sv, (const T*)w, sw, (T*)out_dot,
                                      (T*)out_ss, (T*)ws, batch, (hipStream_t)stream));
}

int gpk_rowreduce(int dtype, const void* z, int64_t rows, int64_t n, int64_t ld, const void* w, void* out_dot, void* out_ss, void* stream) {
    D1(dtype, gpk_rowreduce_launch<T>((const T*)z, rows, n, ld, (const T*)w, (T*)out_dot, (T*)out_ss, (hipStream_t)stream));
}

int gpk_tril(int dtype, void* a, int64_t n, int64_t ld, int64_t sa, int64_t batch, void* stream) {
    D1(dtype, gpk_tril_launch<T>((T*)a, n, ld, sa, batch, (hipStream_t)stream));
}

int gpk_sum_lower(int dtype, const void* parts, int64_t nparts, int64_t n, int64_t ldp, int64_t sp, void* out, int64_t ldo, void* stream) {
    D1(dtype, gpk_sum_lower_launch<T>((const T*)parts, nparts, n, ldp, sp, (T*)out, ldo, (hipStream_t)stream));
}

int gpk_add_diag(int dtype, void* a, int64_t n, int64_t ld, int64_t sa, double s, const void* v,
                 int64_t sv, int64_t batch, void* stream) {
    D1(dtype, gpk_add_diag_launch<T>((T*)a, n, ld, sa, (T)s, (const T*)v, sv, batch, (hipStream_t)stream));
}

int gpk_copy2d(int dtype, const void* src, int64_t lds, int64_t ss, void* dst, int64_t ldd, int64_t sd,
               int64_t rows, int64_t cols, int64_t batch, void* stream) {
    D1(dtype, gpk_copy2d_launch<T>((const T*)src, lds, ss, (T*)dst, ldd, sd, rows, cols, batch,
                                   (hipStream_t)stream));
}

int gpk_scale_cols(int dtype, void* v, int64_t rows, int64_t cols, int64_t ld, int64_t sv, const void* s,
                   int64_t ss, int64_t batch, void* stream) {
    D1(dtype, gpk_scale_cols_launch<T>((T*)v, rows, cols, ld, sv, (const T*)s, ss, batch, (hipStream_t)stream));
}

int gpk_symmetrize(int dtype, void* a, int64_t n, int64_t ld, int64_t sa, int64_t batch, void* stream) {
    D1(dtype, gpk_symmetrize_launch<T>((T*)a, n, ld, sa, batch, (hipStream_t)stream));
}

int gpk_gemv(int dtype, int trans, int64_t m, int64_t k, int nrhs, double alpha, const void* a, int64_t lda,
             int64_t sa, const void* x, int64_t ldx, int64_t sx, double beta, void* y, int64_t ldy, int64_t sy,
             int64_t batch, void* stream) {
    if (trans != 0) return -2;   // A^T x is gpk_colreduce (nrhs = 1) or gpk_gemm
    D1(dtype, gpk_gemv_launch<T>(m, k, nrhs, (T)alpha, (const T*)a, lda, sa, (const T*)x, ldx, sx, (T)beta,
                                 (T*)y, ldy, sy, batch, (hipStream_t)stream));
}

int gpk_trtri_lower(int dtype, const void* l, int64_t n, int64_t ld, const void* dinv_sb, int sb, void* w,
                    int64_t ldw, void* tmp, void* stream) {
    D1(dtype, gpk_trtri_launch<T>((const T*)l, n, ld, (const T*)dinv_sb, sb, (T*)w, ldw, (T*)tmp, (hipStream_t)stream));
}

int64_t gpk_kmat_vjp_blocks(int64_t n) { return gpk_kmat_vjp_blocks_impl(n); }

int gpk_kmat_vjp(int dtype, const int* kinds, const double* inv_ls, int nterms, const void* x, int64_t n,
                 int64_t ldx, int d, const void* kinv, int64_t ldk, const void* alpha, int ncols, int64_t lda,
                 const double* g, void* partial, void* diag_g, void* stream) {
    D1(dtype, gpk_kmat_vjp_launch<T>(kinds, inv_ls, nterms, (const T*)x, n, ldx, d, (const T*)kinv, ldk,
                                     (const T*)alpha, ncols, lda, g, (T*)partial, (T*)diag_g, (hipStream_t)stream));
}

int gpk_kmat_vjp_dense_grid(int64_t n, int64_t m, int64_t* rowtiles, int64_t* nchunks) {
    if (rowtiles == nullptr || nchunks == nullptr) return GPK_ERR_ARG(3);
    gpk_kmat_vjp_dense_grid_impl(n, m, rowtiles, nchunks, nullptr);
    return GPK_OK;
}

int gpk_kmat_vjp_dense(int dtype, const int* kinds, const double* variances, const double* inv_ls, int nterms,
                       const void* x, int64_t n, int64_t ldx, const void* y, int64_t m, int64_t ldy, int d,
                       const void* g, int64_t ldg, const void* colscale, const void* w, const void* b,
                       void* partial, void* colsum, void* gradx, void* stream) {
    D1(dtype, gpk_kmat_vjp_dense_launch<T>(kinds, variances, inv_ls, nterms, (const T*)x, n, ldx, (const T*)y, m,
                                           ldy, d, (const T*)g, ldg, (const T*)colscale, (const T*)w,
                                           (const T*)b, (T*)partial, (T*)colsum, (T*)gradx, (hipStream_t)stream));
}

}  // extern "C"
