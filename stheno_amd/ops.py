"""Tensor-level ops of the GP hot path, backed by ``libgpk.so`` (HIP, gfx950).

Every function takes/returns ``torch.Tensor`` objects that live on a HIP device; the
tensors provide memory and the stream, ``libgpk.so`` does the arithmetic.  There is
no CPU implementation in this package: calling an op with a CPU tensor, or without
the built extension, raises.

``set_backend`` exists so that the host-side model logic can be unit-tested on a
machine without a GPU by injecting a checker backend from ``tests/`` (see
``tests/conftest.py``); nothing in the package ever installs another backend.
"""
import ctypes

import torch

from . import _native

__all__ = ["get_backend", "set_backend", "HipBackend", "KTerms"]

_KIND_IDS = {
    "eq": _native.K_EQ,
    "matern12": _native.K_MATERN12,
    "matern32": _native.K_MATERN32,
    "matern52": _native.K_MATERN52,
    "linear": _native.K_LINEAR,
    "const": _native.K_CONST,
}


class KTerms:
    """A kernel as a sum of ``variance * kind(. / scale)`` terms (host-side descriptor)."""

    def __init__(self, terms):
        terms = list(terms)
        if len(terms) > _native.MAX_TERMS:
            raise ValueError(f"at most {_native.MAX_TERMS} kernel terms are supported")
        self.terms = [(str(k), float(v), float(s)) for k, v, s in terms]
        for k, _, s in self.terms:
            if k not in _KIND_IDS:
                raise ValueError(f"unknown kernel kind {k!r}")
            if not s > 0:
                raise ValueError("length scales must be positive")

    def __len__(self):
        return len(self.terms)

    def c_arrays(self):
        n = len(self.terms)
        kinds = (ctypes.c_int * max(n, 1))(*[_KIND_IDS[k] for k, _, _ in self.terms])
        var = (ctypes.c_double * max(n, 1))(*[v for _, v, _ in self.terms])
        ils = (ctypes.c_double * max(n, 1))(*[1.0 / s for _, _, s in self.terms])
        return kinds, var, ils, n


def _dtype_id(t):
    if t.dtype == torch.float64:
        return _native.GPK_F64
    if t.dtype == torch.float32:
        return _native.GPK_F32
    raise TypeError(f"stheno_amd supports float32/float64 tensors, got {t.dtype}")


def _as3(t):
    """View a (..., R, C) tensor as (B, R, C) with unit inner stride; returns (t3, batch_shape)."""
    if t.dim() < 2:
        raise ValueError("expected a matrix")
    bshape = tuple(t.shape[:-2])
    if t.stride(-1) != 1 and t.shape[-1] > 1:
        t = t.contiguous()
    if t.dim() == 2:
        return t.unsqueeze(0), bshape
    if t.dim() > 3:
        t = t.reshape(-1, t.shape[-2], t.shape[-1])
    if t.dim() == 3 and t.shape[0] > 1 and t.stride(-1) != 1:
        t = t.contiguous()
    return t, bshape


def _as3_out(t):
    """``_as3`` for an OUTPUT: a tensor that would have to be copied to get a unit inner stride cannot
    receive results in place -- refuse it (a transposed view once swallowed a GEMM update silently)."""
    t3, bshape = _as3(t)
    if t3.data_ptr() != t.data_ptr() or (t.dim() >= 2 and t.shape[-1] > 1 and t.stride(-1) != 1):
        raise ValueError("output tensors need a unit inner stride (and a regular batch stride); pass a contiguous buffer")
    return t3, bshape


def alloc_matrix(bshape, rows, cols, dtype, device):
    """An uninitialised (..., rows, cols) matrix whose leading dimension is padded to a multiple
    of 16 elements when ``cols`` is not one: rows stay 16-byte aligned, so the kernels keep their
    vector loads for odd sizes.  Returned as a view (unit inner stride, stride(-2) = padded ld)."""
    if cols >= 64 and cols % 16:
        ldp = (cols + 15) // 16 * 16
        return torch.empty(tuple(bshape) + (rows, ldp), dtype=dtype, device=device)[..., :cols]
    return torch.empty(tuple(bshape) + (rows, cols), dtype=dtype, device=device)


_alloc = alloc_matrix      # (module-internal spelling)


def _ld(t3):
    # leading dimension of the (R, C) slices of a (B, R, C) tensor
    return t3.stride(1) if t3.shape[1] > 1 else max(t3.shape[2], 1)


def _bs(t3):
    return t3.stride(0) if t3.shape[0] > 1 else 0


def _tensors_of(args):
    for a in args:
        if torch.is_tensor(a):
            yield a
        elif isinstance(a, (list, tuple)):
            yield from _tensors_of(a)


def _on_operand_device(fn):
    """Run ``fn`` with the operands' device as the current device: ``libgpk`` launches on the CURRENT HIP
    device and ``torch.cuda.current_stream()`` is per device, so a call made while another device is current
    would run on that device's stream against this device's memory.  Operands on different devices are refused."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        dev = None
        for t in _tensors_of(list(args) + list(kwargs.values())):
            if not t.is_cuda:
                continue
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise RuntimeError(f"operands live on different devices ({dev} and {t.device})")
        if dev is None or dev.index is None or dev.index == torch.cuda.current_device():
            return fn(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)

    return wrapped


class HipBackend:
    """ctypes calls into ``libgpk.so`` on the current torch HIP stream of the operands' device."""

    name = "hip"

    def __init__(self):
        self.lib = _native.load()

    # -- helpers -------------------------------------------------------------
    @staticmethod
    def _check(*tensors):
        for t in tensors:
            if t is None:
                continue
            if not t.is_cuda:
                raise RuntimeError(
                    "stheno_amd ops run on a HIP device only (got a CPU tensor); "
                    "there is no CPU fallback in this package"
                )

    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _ptr(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)

    @staticmethod
    def _st(code, what):
        if code != 0:
            raise RuntimeError(f"libgpk {what} failed with status {code}")

    # -- kernel matrices -----------------------------------------------------
    @_on_operand_device
    def kmat(self, terms, x, y=None, *, lower=False, diag_add=0.0, diag_vec=None, out=None, accumulate=False):
        """``out[b, i, j] (+)= k(x[b, i], y[b, j])``; ``y is None`` means the symmetric case
        (then ``diag_add`` / ``diag_vec`` go on the diagonal)."""
        symmetric = y is None
        x3, bshape = _as3(x)
        y3 = x3 if symmetric else _as3(y)[0]
        self._check(x3, y3, diag_vec, out)
        B, n, d = x3.shape
        m = y3.shape[1]
        if y3.shape[0] != B or y3.shape[2] != d:
            raise ValueError("x and y must agree in batch size and input dimension")
        if out is None:
            out = _alloc(bshape, n, m, x.dtype, x.device)
        o3, _ = _as3_out(out)
        dv = None
        if diag_vec is not None:
            dv = diag_vec.reshape(B, n).contiguous()
        kinds, var, ils, nt = terms.c_arrays()
        code = self.lib.gpk_kmat(
            _dtype_id(x3), kinds, var, ils, nt, self._ptr(x3), n, _ld(x3), _bs(x3), self._ptr(y3), m, _ld(y3),
            _bs(y3), d, self._ptr(o3), _ld(o3), _bs(o3), B, int(lower), int(symmetric), float(diag_add),
            self._ptr(dv), n if dv is not None else 0, int(accumulate), self._stream(),
        )
        self._st(code, "gpk_kmat")
        return out

    @_on_operand_device
    def kdiag(self, terms, x):
        x3, bshape = _as3(x)
        self._check(x3)
        B, n, d = x3.shape
        out = torch.empty(bshape + (n,), dtype=x.dtype, device=x.device)
        kinds, var, ils, nt = terms.c_arrays()
        code = self.lib.gpk_kdiag(_dtype_id(x3), kinds, var, ils, nt, self._ptr(x3), n, _ld(x3), _bs(x3), d,
                                  self._ptr(out), n, B, self._stream())
        self._st(code, "gpk_kdiag")
        return out

    # -- factorisation -------------------------------------------------------
    supports_potrf_rhs = True

    @_on_operand_device
    def potrf_(self, a, nbo=0, lookahead_nb=0, lookahead_sb=0, rhs=None):
        """In-place lower Cholesky of ``a`` (..., n, n).  ``rhs`` (plain path only): a contiguous (B, n) tensor, one right-hand side per
        matrix, overwritten by ``L^{-1} rhs`` (``gpk_potrf_rhs``: for large fp32 batches the sweep runs beside the factorisation).
        Returns ``(dinv, info)``, or
        ``(dinv, info, dinv_sb)`` when ``lookahead_nb`` (256 ... 4096) selects the look-ahead
        factorisation of ONE large matrix: ``dinv_sb`` are the inverses of the ``sb x sb`` diagonal
        blocks of the factor (what ``trtri_merge(l, dinv, sb)`` would compute), ``sb = lookahead_sb`` or ``lookahead_nb``."""
        a3, _ = _as3(a)
        if a3.data_ptr() != a.data_ptr():
            raise ValueError("potrf_ needs a tensor with unit inner stride (it factorises in place)")
        self._check(a3)
        B, n, _ = a3.shape
        nblk = (max(n, 1) + 127) // 128
        dinv = torch.empty((B, nblk, 128, 128), dtype=a.dtype, device=a.device)
        info = torch.zeros((B,), dtype=torch.int32, device=a.device)
        if lookahead_nb:
            if B != 1 or rhs is not None:
                raise ValueError("the look-ahead factorisation takes one matrix (and no right-hand side: gpk_potrf_rows_rhs)")
            nb = int(lookahead_nb)
            sb = int(lookahead_sb) or nb
            dnb = torch.empty((1, (n + sb - 1) // sb, sb, sb), dtype=a.dtype, device=a.device)
            ws = torch.empty((int(self.lib.gpk_potrf_la_ws_elems(n, nb)),), dtype=a.dtype, device=a.device)
            if sb == nb:
                code = self.lib.gpk_potrf_la(_dtype_id(a3), self._ptr(a3), n, _ld(a3), self._ptr(dinv), self._ptr(dnb), nb,
                                             self._ptr(ws), self._ptr(info), self._stream())
            else:
                code = self.lib.gpk_potrf_la_split(_dtype_id(a3), self._ptr(a3), n, _ld(a3), self._ptr(dinv), self._ptr(dnb), nb, sb,
                                                   self._ptr(ws), self._ptr(info), self._stream())
            self._st(code, "gpk_potrf_la")
            # `ws` is freed here while the factorisation may still be running: torch's caching allocator only reuses the
            # block for work enqueued later on this same stream, and the helper stream has joined it by then
            return dinv, info, dnb
        if rhs is not None:
            if tuple(rhs.shape) != (B, n) or not rhs.is_contiguous() or rhs.dtype != a.dtype or rhs.device != a.device:
                raise ValueError("potrf_: rhs must be a contiguous (batch, n) tensor of the matrices' dtype and device")
            tmp = torch.empty((B * 128 + 16,), dtype=a.dtype, device=a.device)
            code = self.lib.gpk_potrf_rhs(_dtype_id(a3), self._ptr(a3), n, _ld(a3), _bs(a3), B, self._ptr(dinv), self._ptr(info), int(nbo),
                                          self._ptr(rhs), n, self._ptr(tmp), self._stream())
            self._st(code, "gpk_potrf_rhs")
            # (`tmp` is released here: the caching allocator hands it to later work of THIS stream only, which the side stream has joined)
            return dinv, info
        code = self.lib.gpk_potrf(_dtype_id(a3), self._ptr(a3), n, _ld(a3), _bs(a3), B, self._ptr(dinv),
                                  self._ptr(info), int(nbo), self._stream())
        self._st(code, "gpk_potrf")
        return dinv, info

    @_on_operand_device
    def potrf_rows_(self, a, lookahead_nb=0, lookahead_sb=0, rhs_row=False, tail_inverses=True):
        """In-place lower Cholesky of the leading ``n x n`` of ``a`` (rows, n), rows > n, carrying the rows under it through the
        factorisation (``gpk_potrf_rows``): they come out as ``a[n:] L^{-T}``.  ``n`` a multiple of 128.  Returns ``(dinv, info, dinv_sb or None)``.
        ``rhs_row``: the last 64 rows are a strip whose first row is ONE right-hand side, the rest zero padding (``gpk_potrf_rows_rhs``,
        ``GPK_ROWS_RHS``: same result, ``L^{-1} b`` in that row, computed by matrix-vector products beside the look-ahead's trailing updates).  ``tail_inverses=False``: the merged inverses of the
        look-ahead's plain tail are not computed -- ``dinv_sb`` comes back as ``None`` (incomplete), callers merge on demand."""
        if a.dim() != 2 or a.stride(-1) != 1 or a.shape[0] <= a.shape[1] or a.shape[1] % 128 != 0:
            raise ValueError("potrf_rows_ takes one (rows, n) matrix with rows > n, n a multiple of 128 and unit inner stride")
        self._check(a)
        rows, n = a.shape
        dinv = torch.empty((1, n // 128 + 1, 128, 128), dtype=a.dtype, device=a.device)      # (one slot more: control words of the last panel)
        info = torch.zeros((1,), dtype=torch.int32, device=a.device)
        nb = int(lookahead_nb)
        sb = (int(lookahead_sb) or nb) if nb else 0
        dnb = ws = None
        if nb:
            dnb = torch.empty((1, (n + sb - 1) // sb, sb, sb), dtype=a.dtype, device=a.device)
            ws = torch.empty((int(self.lib.gpk_potrf_la_ws_elems(rows, nb)),), dtype=a.dtype, device=a.device)
        flags = (1 if rhs_row else 0) | (0 if tail_inverses else 2)
        if flags:
            code = self.lib.gpk_potrf_rows_rhs(_dtype_id(a), self._ptr(a), n, rows, a.stride(0), self._ptr(dinv), self._ptr(dnb) if nb else None,
                                               nb, sb if sb != nb else 0, self._ptr(ws) if nb else None, self._ptr(info), flags, self._stream())
            self._st(code, "gpk_potrf_rows_rhs")
            return dinv[:, : n // 128], info, (dnb if tail_inverses else None)
        code = self.lib.gpk_potrf_rows(_dtype_id(a), self._ptr(a), n, rows, a.stride(0), self._ptr(dinv), self._ptr(dnb) if nb else None, nb,
                                       sb if sb != nb else 0, self._ptr(ws) if nb else None, self._ptr(info), self._stream())
        self._st(code, "gpk_potrf_rows")
        return dinv[:, : n // 128], info, dnb

    @_on_operand_device
    def rowreduce(self, z, w=None, *, want_dot=True, want_ss=False):
        """Row reductions of ``z`` (rows, n): ``dot[i] = sum_k z[i, k] w[k]`` and / or ``ss[i] = sum_k z[i, k]^2`` in one pass
        (``gpk_rowreduce``) -- the posterior mean and marginal variance from the TRANSPOSED whitened cross-covariance."""
        if z.dim() != 2 or z.stride(-1) != 1:
            raise ValueError("rowreduce takes one matrix with unit inner stride")
        self._check(z, w)
        rows, n = z.shape
        dot = torch.empty((rows,), dtype=z.dtype, device=z.device) if (want_dot and w is not None) else None
        ss = torch.empty((rows,), dtype=z.dtype, device=z.device) if want_ss else None
        if w is not None:
            w = w.reshape(-1).contiguous()
            if w.shape[0] != n:
                raise ValueError("rowreduce: the vector has %d entries, the rows %d" % (w.shape[0], n))
        code = self.lib.gpk_rowreduce(_dtype_id(z), self._ptr(z), rows, n, z.stride(0), self._ptr(w) if dot is not None else None,
                                      self._ptr(dot) if dot is not None else None, self._ptr(ss) if ss is not None else None, self._stream())
        self._st(code, "gpk_rowreduce")
        return dot, ss

    @_on_operand_device
    def trtri_merge(self, l, dinv, sb):
        l3, _ = _as3(l)
        self._check(l3, dinv)
        B, n, _ = l3.shape
        nsb = (n + sb - 1) // sb
        dsb = torch.empty((B, nsb, sb, sb), dtype=l.dtype, device=l.device)
        tmp = torch.empty((nsb * sb * sb // 4 + 16,), dtype=l.dtype, device=l.device)
        code = self.lib.gpk_trtri_merge(_dtype_id(l3), self._ptr(l3), n, _ld(l3), _bs(l3), B, self._ptr(dinv), sb,
                                        self._ptr(dsb), self._ptr(tmp), self._stream())
        self._st(code, "gpk_trtri_merge")
        return dsb

    @_on_operand_device
    def tri_solve_(self, l, dinv_sb, sb, b):
        """``L^{-1} b``; ``b`` is (..., n, nrhs) with unit inner stride and is OVERWRITTEN (with the solution for up to 8 columns,
        with intermediate values otherwise): use the return value."""
        l3, _ = _as3(l)
        b3, _ = _as3(b)
        if b3.data_ptr() != b.data_ptr():
            raise ValueError("tri_solve_ needs a right-hand side with unit inner stride")
        self._check(l3, b3, dinv_sb)
        B, n, _ = l3.shape
        nrhs = b3.shape[2]
        if b3.shape[0] != B or b3.shape[1] != n:
            raise ValueError("right-hand side does not match the factor")
        if n == 0 or nrhs == 0:
            return b
        if nrhs <= 8:
            tmp = torch.empty((B * sb * nrhs + 16,), dtype=b.dtype, device=b.device)     # (+ GPK_TRSV_CTRL_ELEMS: the single-launch sweep's control words)
            code = self.lib.gpk_trsv_lower(_dtype_id(l3), self._ptr(l3), n, _ld(l3), _bs(l3), self._ptr(dinv_sb), sb,
                                           self._ptr(b3), nrhs, _ld(b3), _bs(b3), self._ptr(tmp), B, self._stream())
            self._st(code, "gpk_trsv_lower")
            return b
        # many right-hand sides: the recursive blocked solve, out of place -- solved blocks go to a second buffer (no copy per
        # block), `b` is used up as workspace; the caller takes the return value (Chol.solve_)
        x = torch.empty(b.shape, dtype=b.dtype, device=b.device)
        x3, _ = _as3(x)
        code = self.lib.gpk_trsm_lower_to(_dtype_id(l3), self._ptr(l3), n, _ld(l3), _bs(l3), self._ptr(dinv_sb), sb,
                                          self._ptr(b3), nrhs, _ld(b3), _bs(b3), self._ptr(x3), _ld(x3), _bs(x3), B, self._stream())
        self._st(code, "gpk_trsm_lower_to")
        return x

    # -- products ------------------------------------------------------------
    @_on_operand_device
    def trtri(self, l, dinv_sb, sb):
        """``L^{-1}`` of an unbatched factor as a full (n, n) lower-triangular matrix."""
        self._check(l, dinv_sb)
        n = l.shape[-1]
        w = _alloc((), n, n, l.dtype, l.device)
        tmp = torch.empty((sb * n,), dtype=l.dtype, device=l.device)
        code = self.lib.gpk_trtri_lower(_dtype_id(l), self._ptr(l), n, l.stride(0), self._ptr(dinv_sb), sb,
                                        self._ptr(w), w.stride(0), self._ptr(tmp), self._stream())
        self._st(code, "gpk_trtri_lower")
        return w

    @_on_operand_device
    def gemm(self, a, b, *, a_kmajor=True, b_kmajor=True, alpha=1.0, beta=0.0, out=None, lower_only=False,
             tri_k=False, tri_k_lower=False):
        """``out[m, n] = alpha * sum_k a(m, k) b(n, k) + beta * out``.

        ``a_kmajor``: ``a`` is stored (M, K); otherwise (K, M).  ``b_kmajor``: ``b`` is stored
        (N, K); otherwise (K, N)."""
        a3, bshape = _as3(a)
        b3, _ = _as3(b)
        self._check(a3, b3, out)
        B = max(a3.shape[0], b3.shape[0])
        M, K = (a3.shape[1], a3.shape[2]) if a_kmajor else (a3.shape[2], a3.shape[1])
        N, K2 = (b3.shape[1], b3.shape[2]) if b_kmajor else (b3.shape[2], b3.shape[1])
        if K != K2:
            raise ValueError("inner dimensions do not match")
        if out is None:
            if beta != 0.0:
                raise ValueError("beta != 0 requires `out`")
            out = _alloc(bshape if a3.shape[0] >= b3.shape[0] else tuple(b.shape[:-2]), M, N, a.dtype, a.device)
        o3, _ = _as3_out(out)
        code = self.lib.gpk_gemm(_dtype_id(a3), int(a_kmajor), int(b_kmajor), M, N, K, float(alpha), self._ptr(a3),
                                 _ld(a3), _bs(a3), self._ptr(b3), _ld(b3), _bs(b3), float(beta), self._ptr(o3),
                                 _ld(o3), _bs(o3), B, int(lower_only) | (2 if tri_k else 0) | (4 if tri_k_lower else 0),
                                 self._stream())
        self._st(code, "gpk_gemm")
        return out

    @_on_operand_device
    def gemm_colscale(self, a, b, colscale=None, *, want_colss=False, a_kmajor=True, b_kmajor=True, tri_k_lower=False):
        """``out[m, n] = sum_k a(m, k) b(n, k) * colscale[n]`` (unbatched) and, with ``want_colss``, the column sums of squares of the
        UNSCALED product -- both in the epilogue of the one GEMM (``gpk_gemm_colscale``).  Returns ``(out, colss or None)``."""
        M, K = (a.shape[0], a.shape[1]) if a_kmajor else (a.shape[1], a.shape[0])
        N, K2 = (b.shape[0], b.shape[1]) if b_kmajor else (b.shape[1], b.shape[0])
        if a.dim() != 2 or b.dim() != 2 or K != K2:
            raise ValueError("gemm_colscale takes two matrices with matching inner dimensions")
        a3, _ = _as3(a)
        b3, _ = _as3(b)
        self._check(a3, b3, None)
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
        part = None
        if want_colss:
            part = torch.empty((int(self.lib.gpk_gemm_colss_rows(M)), N), dtype=a.dtype, device=a.device)
        if colscale is not None:
            colscale = colscale.to(dtype=a.dtype).contiguous()
            if colscale.shape != (N,):
                raise ValueError("colscale must have one entry per column")
        code = self.lib.gpk_gemm_colscale(_dtype_id(a3), int(a_kmajor), int(b_kmajor), M, N, K, 1.0, self._ptr(a3), _ld(a3), self._ptr(b3),
                                          _ld(b3), self._ptr(out), N, 4 if tri_k_lower else 0,
                                          self._ptr(colscale) if colscale is not None else None,
                                          self._ptr(part) if part is not None else None, N, self._stream())
        self._st(code, "gpk_gemm_colscale")
        return out, (part.sum(0) if part is not None else None)       # (64 x N partial sums: a statistics buffer, added up by torch)

    @_on_operand_device
    def gemv(self, a, x, *, alpha=1.0, beta=0.0, out=None):
        """``out = alpha * a @ x + beta * out`` for (..., M, K) @ (..., K, nrhs <= 8)."""
        a3, bshape = _as3(a)
        x3, _ = _as3(x)
        self._check(a3, x3, out)
        B, M, K = a3.shape
        nrhs = x3.shape[2]
        if out is None:
            out = torch.empty(bshape + (M, nrhs), dtype=a.dtype, device=a.device)
        o3, _ = _as3_out(out)
        code = self.lib.gpk_gemv(_dtype_id(a3), 0, M, K, nrhs, float(alpha), self._ptr(a3), _ld(a3), _bs(a3),
                                 self._ptr(x3), _ld(x3), _bs(x3), float(beta), self._ptr(o3), _ld(o3), _bs(o3), B,
                                 self._stream())
        self._st(code, "gpk_gemv")
        return out

    # -- reductions ----------------------------------------------------------
    @_on_operand_device
    def logdet_chol(self, l):
        l3, bshape = _as3(l)
        self._check(l3)
        B, n, _ = l3.shape
        out = torch.empty((B,), dtype=l.dtype, device=l.device)
        code = self.lib.gpk_logdet_chol(_dtype_id(l3), self._ptr(l3), n, _ld(l3), _bs(l3), B, self._ptr(out),
                                        self._stream())
        self._st(code, "gpk_logdet_chol")
        return out.reshape(bshape)

    @_on_operand_device
    def colreduce(self, v, w=None, *, want_dot=False, want_ss=True):
        """Column reductions of ``v`` (..., R, C): ``(v^T w, colsumsq(v))`` (``None`` for the
        one not requested).  ``w``: (..., R) or (..., R, 1)."""
        v3, bshape = _as3(v)
        self._check(v3, w)
        B, R, C = v3.shape
        dot = ss = None
        w2 = None
        if want_dot:
            if w is None:
                raise ValueError("want_dot needs w")
            w2 = w.reshape(B, R).contiguous()
            dot = torch.empty((B, C), dtype=v.dtype, device=v.device)
        if want_ss:
            ss = torch.empty((B, C), dtype=v.dtype, device=v.device)
        nchunk = int(self.lib.gpk_colreduce_chunks(R))
        ws = torch.empty((2 * B * nchunk * max(C, 1),), dtype=v.dtype, device=v.device)
        code = self.lib.gpk_colreduce(_dtype_id(v3), self._ptr(v3), R, C, _ld(v3), _bs(v3), self._ptr(w2), R,
                                      self._ptr(dot), self._ptr(ss), self._ptr(ws), B, self._stream())
        self._st(code, "gpk_colreduce")
        if dot is not None:
            dot = dot.reshape(bshape + (C,))
        if ss is not None:
            ss = ss.reshape(bshape + (C,))
        return dot, ss

    @_on_operand_device
    def kmat_vjp(self, terms, x, kinv, alpha, g):
        """Sums for the hyper-parameter gradient of the log-density (see gpk_kmat_vjp):
        returns ``(S, trace_G, diag_G)`` with ``S[t] = (sum G kappa_t, sum G kappa_t' q)``.
        ``x`` (n, d), ``kinv`` (n, n; lower triangle read), ``alpha`` (n, C <= 8), ``g``: C floats."""
        self._check(x, kinv, alpha)
        n, d = x.shape
        C = alpha.shape[1]
        kinds, _, ils, nt = terms.c_arrays()
        nb = int(self.lib.gpk_kmat_vjp_blocks(n))
        width = 2 * _native.MAX_TERMS + 1
        partial = torch.zeros((nb, width), dtype=x.dtype, device=x.device)
        diag_g = torch.empty((n,), dtype=x.dtype, device=x.device)
        gs = (ctypes.c_double * max(C, 1))(*[float(v) for v in g])
        alpha = alpha.contiguous()
        code = self.lib.gpk_kmat_vjp(_dtype_id(x), kinds, ils, nt, self._ptr(x), n, x.stride(0), d, self._ptr(kinv),
                                     kinv.stride(0), self._ptr(alpha), C, alpha.stride(0), gs, self._ptr(partial),
                                     self._ptr(diag_g), self._stream())
        self._st(code, "gpk_kmat_vjp")
        tot = partial.sum(0)
        return tot[: 2 * nt].reshape(nt, 2), tot[2 * _native.MAX_TERMS], diag_g

    @_on_operand_device
    def kmat_vjp_dense(self, terms, x, y, g, colscale=None, w=None, b=None, want_colsum=False, want_gradx=False):
        """Sums over an explicit cotangent ``Geff = g * colscale[None, :] + w[:, None] b[None, :]`` of
        ``K = k(x, y)`` (see gpk_kmat_vjp_dense): returns ``(S, colsum, gradx)`` with
        ``S[t] = (sum Geff kappa_t, sum Geff kappa_t' q)``, ``colsum[j] = sum_i Geff_ij K_ij`` and
        ``gradx[i] = sum_j Geff_ij dK_ij/dx_i`` (``None`` unless asked for)."""
        self._check(x, y, g, colscale, w, b)
        n, d = x.shape
        m = y.shape[0]
        if g.shape != (n, m) or g.stride(1) != 1:
            raise ValueError("cotangent must be an (n, m) matrix with unit inner stride")
        rt, nc = ctypes.c_int64(), ctypes.c_int64()
        self._st(self.lib.gpk_kmat_vjp_dense_grid(n, m, ctypes.byref(rt), ctypes.byref(nc)), "gpk_kmat_vjp_dense_grid")
        rt, nc = rt.value, nc.value
        width = 2 * _native.MAX_TERMS + 1
        partial = torch.zeros((rt * nc, width), dtype=x.dtype, device=x.device)
        colsum = torch.zeros((rt, m), dtype=x.dtype, device=x.device) if want_colsum else None
        gradx = torch.zeros((nc, n, d), dtype=x.dtype, device=x.device) if want_gradx else None
        kinds, var, ils, nt = terms.c_arrays()
        x, y = x.contiguous(), y.contiguous()
        cs = colscale.contiguous() if colscale is not None else None
        w = w.contiguous() if w is not None else None
        b = b.contiguous() if b is not None else None
        code = self.lib.gpk_kmat_vjp_dense(_dtype_id(x), kinds, var, ils, nt, self._ptr(x), n, x.stride(0), self._ptr(y),
                                           m, y.stride(0), d, self._ptr(g), g.stride(0), self._ptr(cs), self._ptr(w),
                                           self._ptr(b), self._ptr(partial), self._ptr(colsum), self._ptr(gradx),
                                           self._stream())
        self._st(code, "gpk_kmat_vjp_dense")
        tot = partial.sum(0)
        return (tot[: 2 * nt].reshape(nt, 2), colsum.sum(0) if want_colsum else None,
                gradx.sum(0) if want_gradx else None)

    # -- in-place odds and ends ------------------------------------------------
    @_on_operand_device
    def tril_(self, a):
        a3, _ = _as3_out(a)
        self._check(a3)
        B, n, _ = a3.shape
        self._st(self.lib.gpk_tril(_dtype_id(a3), self._ptr(a3), n, _ld(a3), _bs(a3), B, self._stream()), "gpk_tril")
        return a

    @_on_operand_device
    def sum_lower(self, parts, out):
        """``out[i, j] = sum_s parts[s, i, j]`` for ``j <= i`` (``gpk_sum_lower``): the partial products of a split-K symmetric update,
        of which only the lower tiles were written, added up in a fixed order.  Entries above the diagonal of ``out`` are unspecified."""
        if parts.dim() != 3 or out.dim() != 2 or parts.shape[1] != parts.shape[2] or tuple(out.shape) != tuple(parts.shape[1:]):
            raise ValueError("sum_lower takes (S, n, n) partial products and an (n, n) output")
        if parts.stride(-1) != 1 or out.stride(-1) != 1:
            raise ValueError("sum_lower: unit inner strides")
        self._check(parts, out)
        self._st(self.lib.gpk_sum_lower(_dtype_id(out), self._ptr(parts), parts.shape[0], out.shape[0], parts.stride(1), parts.stride(0),
                                        self._ptr(out), out.stride(0), self._stream()), "gpk_sum_lower")
        return out

    @_on_operand_device
    def symmetrize_(self, a):
        a3, _ = _as3_out(a)
        self._check(a3)
        B, n, _ = a3.shape
        self._st(self.lib.gpk_symmetrize(_dtype_id(a3), self._ptr(a3), n, _ld(a3), _bs(a3), B, self._stream()),
                 "gpk_symmetrize")
        return a

    @_on_operand_device
    def add_diag_(self, a, s=0.0, v=None):
        a3, _ = _as3_out(a)
        self._check(a3, v)
        B, n, _ = a3.shape
        v2 = v.reshape(B, n).contiguous() if v is not None else None
        self._st(self.lib.gpk_add_diag(_dtype_id(a3), self._ptr(a3), n, _ld(a3), _bs(a3), float(s), self._ptr(v2),
                                       n if v2 is not None else 0, B, self._stream()), "gpk_add_diag")
        return a

    @_on_operand_device
    def scale_cols_(self, v, s):
        v3, _ = _as3_out(v)
        self._check(v3, s)
        B, R, C = v3.shape
        s2 = s.reshape(B, C).contiguous()
        self._st(self.lib.gpk_scale_cols(_dtype_id(v3), self._ptr(v3), R, C, _ld(v3), _bs(v3), self._ptr(s2), C, B,
                                         self._stream()), "gpk_scale_cols")
        return v

    @_on_operand_device
    def copy(self, src):
        """Fresh copy of a (..., R, C) tensor (strided 2-D copy kernel; rows 16-byte aligned)."""
        s3, bshape = _as3(src)
        self._check(s3)
        B, R, C = s3.shape
        out = _alloc(bshape, R, C, src.dtype, src.device)
        o3, _ = _as3_out(out)
        self._st(self.lib.gpk_copy2d(_dtype_id(s3), self._ptr(s3), _ld(s3), _bs(s3), self._ptr(o3), _ld(o3), _bs(o3),
                                     R, C, B, self._stream()), "gpk_copy2d")
        return out


_backend = None


def get_backend():
    """The op backend; instantiates :class:`HipBackend` on first use (raises if
    ``libgpk.so`` is not built)."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def set_backend(backend):
    """Install an op backend.  TEST HOOK ONLY (host-logic unit tests without a GPU);
    returns the previous backend."""
    global _backend
    prev, _backend = _backend, backend
    return prev
