"""ctypes binding of ``libgpk.so`` (the C ABI declared in ``include/gpk.h``).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C stheno_amd/csrc``).
There is NO fallback: if the shared object is missing, loading raises.
"""
import ctypes
import os

_c_int = ctypes.c_int
_c_i64 = ctypes.c_int64
_c_dbl = ctypes.c_double
_c_ptr = ctypes.c_void_p
_p_int = ctypes.POINTER(ctypes.c_int)
_p_dbl = ctypes.POINTER(ctypes.c_double)

GPK_F32 = 0
GPK_F64 = 1

K_EQ, K_MATERN12, K_MATERN32, K_MATERN52, K_LINEAR, K_CONST = range(6)
MAX_TERMS = 8
DIAG_BLOCK = 128

#: name -> (restype, argtypes); mirrors include/gpk.h declaration by declaration.
SIGNATURES = {
    "gpk_version": (_c_int, []),
    "gpk_init": (_c_int, []),
    "gpk_shutdown": (None, []),
    "gpk_dinv_elems": (_c_i64, [_c_i64]),
    "gpk_colreduce_chunks": (_c_i64, [_c_i64]),
    "gpk_kmat": (
        _c_int,
        [_c_int, _p_int, _p_dbl, _p_dbl, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64,
         _c_i64, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_dbl, _c_ptr, _c_i64, _c_int,
         _c_ptr],
    ),
    "gpk_kdiag": (
        _c_int,
        [_c_int, _p_int, _p_dbl, _p_dbl, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr, _c_i64,
         _c_i64, _c_ptr],
    ),
    "gpk_potrf": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_int, _c_ptr]),
    "gpk_potrf_la_ws_elems": (_c_i64, [_c_i64, _c_int]),
    "gpk_potrf_la": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr]),
    "gpk_potrf_la_split": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr]),
    "gpk_potrf_rows": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_int, _c_int, _c_ptr, _c_ptr, _c_ptr]),
    "gpk_potrf_rhs": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_ptr]),
    "gpk_potrf_rows_rhs": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_int, _c_int, _c_ptr, _c_ptr, _c_int, _c_ptr]),
    "gpk_gemm_update2": (_c_int, [_c_int, _c_ptr, _c_int, _c_dbl, _c_ptr, _c_int, _c_ptr]),
    "gpk_tune": (None, [_c_int, _c_i64]),
    "gpk_tune_diag_prof": (None, [_c_ptr]),
    "gpk_tune_tile_prof": (None, [_c_ptr]),
    "gpk_trtri_merge": (
        _c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_int, _c_ptr, _c_ptr, _c_ptr]
    ),
    "gpk_trsm_lower": (
        _c_int,
        [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr,
         _c_i64, _c_ptr],
    ),
    "gpk_trsm_lower_to": (
        _c_int,
        [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64,
         _c_i64, _c_i64, _c_ptr],
    ),
    "gpk_trsv_lower": (
        _c_int,
        [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_int, _c_ptr, _c_int, _c_i64, _c_i64, _c_ptr,
         _c_i64, _c_ptr],
    ),
    "gpk_gemm": (
        _c_int,
        [_c_int, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_dbl, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64,
         _c_i64, _c_dbl, _c_ptr, _c_i64, _c_i64, _c_i64, _c_int, _c_ptr],
    ),
    "gpk_logdet_chol": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr]),
    "gpk_rowreduce": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    "gpk_colreduce": (
        _c_int,
        [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_i64,
         _c_ptr],
    ),
    "gpk_tril": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr]),
    "gpk_add_diag": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_dbl, _c_ptr, _c_i64, _c_i64, _c_ptr]),
    "gpk_scale_cols": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_ptr]),
    "gpk_symmetrize": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr]),
    "gpk_gemv": (
        _c_int,
        [_c_int, _c_int, _c_i64, _c_i64, _c_int, _c_dbl, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64,
         _c_dbl, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr],
    ),
    "gpk_trtri_lower": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_int, _c_ptr, _c_i64, _c_ptr, _c_ptr]),
    "gpk_kmat_vjp_blocks": (_c_i64, [_c_i64]),
    "gpk_kmat_vjp": (
        _c_int,
        [_c_int, _p_int, _p_dbl, _c_int, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr, _c_i64, _c_ptr, _c_int, _c_i64,
         _p_dbl, _c_ptr, _c_ptr, _c_ptr],
    ),
    "gpk_kmat_vjp_dense_grid": (_c_int, [_c_i64, _c_i64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "gpk_kmat_vjp_dense": (
        _c_int,
        [_c_int, _p_int, _p_dbl, _p_dbl, _c_int, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_int, _c_ptr,
         _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr],
    ),
    "gpk_gemm_colss_rows": (_c_i64, [_c_i64]),
    "gpk_gemm_colscale": (
        _c_int,
        [_c_int, _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_dbl, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_ptr, _c_i64, _c_int,
         _c_ptr, _c_ptr, _c_i64, _c_ptr],
    ),
    "gpk_mfma_peak": (_c_int, [_c_int, _c_dbl, _c_int, _p_dbl, _p_dbl, _p_dbl, _p_dbl, _c_ptr]),
    "gpk_sum_lower": (_c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr, _c_i64, _c_ptr]),
    "gpk_prof_start": (_c_int, []),
    "gpk_prof_stop": (_c_int, [_c_int, _p_dbl, ctypes.POINTER(ctypes.c_int64), _p_dbl]),
    "gpk_copy2d": (
        _c_int, [_c_int, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr]
    ),
}

_LIB = None


def lib_path():
    """``csrc/libgpk.so`` -- the release library (no tuning knobs).  ``GPK_DEV=1`` selects ``csrc/dev/libgpk.so``, the same sources
    built with mutable knobs (``make -C stheno_amd/csrc``), for A/B measurements; never set in a product run."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    if os.environ.get("GPK_DEV") == "1":
        return os.path.join(here, "dev", "libgpk.so")
    return os.path.join(here, "libgpk.so")


def load():
    """Load ``libgpk.so`` and attach the prototypes.  Raises if it was not built."""
    global _LIB
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C stheno_amd/csrc`). "
                "stheno_amd has no CPU fallback."
            )
        lib = ctypes.CDLL(path)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        _LIB = lib
        import atexit

        atexit.register(lib.gpk_shutdown)      # helper streams must not outlive the HIP runtime's own teardown
        # Development aid for A/B runs of the library's tuning knobs (gpk_tune in include/gpk.h), e.g.
        # GPK_DEV=1 GPK_TUNE="9=4096,38=0" (GPK_DEV=1 loads the dev build above; in the release library gpk_tune does nothing).
        # Ignored unless GPK_DEV=1 is set as well: a stray GPK_TUNE in a user's environment must not change what the product
        # path runs.
        if os.environ.get("GPK_DEV") == "1":
            for item in filter(None, os.environ.get("GPK_TUNE", "").split(",")):
                key, _, value = item.partition("=")
                lib.gpk_tune(int(key), int(value))
    return _LIB
