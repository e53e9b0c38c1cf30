"""pytest configuration.

* ``-m "not gpu"`` (CPU box): the oracle is checked against the reference's known
  answers, the golden fixtures and SciPy; the host-side model logic runs on top of
  ``OracleBackend`` -- a TEST-ONLY op backend implemented with ``oracle/gp_oracle.py``
  and installed through ``stheno_amd.ops.set_backend``; the C ABI library is loaded and
  its exported symbols compared with ``include/gpk.h``.
* ``-m gpu`` (MI355X box): everything goes through ``libgpk.so`` (``HipBackend``) and is
  compared with the oracle / golden fixtures.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import gp_oracle as O  # noqa: E402
from stheno_amd import ops  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu")


def _np(t):
    return t.detach().cpu().numpy()


class OracleBackend:
    """NumPy/SciPy stand-in for ``stheno_amd.ops.HipBackend`` (same method contracts),
    so that the Python host logic can be unit-tested without a GPU.  Never used by the
    package itself."""

    name = "oracle-test-backend"

    @staticmethod
    def _t(a, like):
        return torch.as_tensor(np.ascontiguousarray(a), dtype=like.dtype, device=like.device)

    def kmat(self, terms, x, y=None, *, lower=False, diag_add=0.0, diag_vec=None, out=None, accumulate=False):
        xs = _np(x)
        k = O.kernel_matrix(terms.terms, xs, None if y is None else _np(y)) if len(terms) else \
            np.zeros(xs.shape[:-1] + ((xs if y is None else _np(y)).shape[-2],), dtype=xs.dtype)
        k = np.array(k, dtype=xs.dtype)
        if y is None:
            n = k.shape[-1]
            idx = np.arange(n)
            k[..., idx, idx] += diag_add
            if diag_vec is not None:
                k[..., idx, idx] += _np(diag_vec).reshape(k.shape[:-2] + (n,))
        res = self._t(k, x)
        if out is not None:
            if accumulate:
                out += res
            else:
                out.copy_(res)
            return out
        return res

    def kdiag(self, terms, x):
        return self._t(O.kernel_diag(terms.terms, _np(x)), x)

    supports_potrf_rhs = True

    def potrf_(self, a, nbo=0, rhs=None):
        arr = _np(a)
        batch = int(np.prod(arr.shape[:-2])) if arr.ndim > 2 else 1
        info = torch.zeros((batch,), dtype=torch.int32)
        flat = arr.reshape((-1,) + arr.shape[-2:]).copy()
        sol = _np(rhs).copy() if rhs is not None else None      # (batch, n): one right-hand side per matrix, solved along (gpk_potrf_rhs)
        for b in range(flat.shape[0]):
            sym = np.tril(flat[b]) + np.tril(flat[b], -1).T
            try:
                flat[b] = np.linalg.cholesky(sym)
                if sol is not None:
                    sol[b] = O.solve_lower(flat[b], sol[b][:, None])[:, 0]
            except np.linalg.LinAlgError:
                info[b] = 1
                flat[b] = np.nan
        a.copy_(self._t(flat.reshape(arr.shape), a))
        if sol is not None:
            rhs.copy_(self._t(sol, rhs))
        return None, info

    def potrf_rows_(self, a, lookahead_nb=0, lookahead_sb=0, rhs_row=False, tail_inverses=True):
        """Stand-in for ``HipBackend.potrf_rows_`` (``gpk_potrf_rows``): the leading n x n of ``a`` (rows, n) is factorised in place,
        the rows under it become ``a[n:] L^{-T}`` -- so that the host logic of the posterior-first path runs on the GPU-less box."""
        arr = _np(a).copy()
        rows, n = arr.shape
        info = torch.zeros((1,), dtype=torch.int32)
        sym = np.tril(arr[:n]) + np.tril(arr[:n], -1).T
        try:
            L = np.linalg.cholesky(sym)
            arr[:n] = L
            arr[n:] = O.solve_lower(L, arr[n:].T).T
        except np.linalg.LinAlgError:
            info[0] = 1
            arr[:] = np.nan
        a.copy_(self._t(arr, a))
        return None, info, None

    def rowreduce(self, z, w=None, *, want_dot=True, want_ss=False):
        dot = (z * w.reshape(1, -1)).sum(-1) if (want_dot and w is not None) else None
        ss = (z * z).sum(-1) if want_ss else None
        return dot, ss

    def trtri_merge(self, l, dinv, sb):
        return None

    def tri_solve_(self, l, dinv_sb, sb, b):
        L, Bm = _np(l), _np(b)
        Lf = L.reshape((-1,) + L.shape[-2:])
        Bf = Bm.reshape((-1,) + Bm.shape[-2:]).copy()
        for i in range(Bf.shape[0]):
            if Bf[i].size:
                Bf[i] = O.solve_lower(np.tril(Lf[i if Lf.shape[0] > 1 else 0]), Bf[i])
        b.copy_(self._t(Bf.reshape(Bm.shape), b))
        return b

    def trtri(self, l, dinv_sb, sb):
        return self._t(np.linalg.inv(np.tril(_np(l))), l)

    def gemm(self, a, b, *, a_kmajor=True, b_kmajor=True, alpha=1.0, beta=0.0, out=None, lower_only=False,
             tri_k=False, tri_k_lower=False):
        A = a if a_kmajor else a.transpose(-1, -2)
        Bt = b.transpose(-1, -2) if b_kmajor else b
        res = alpha * (A @ Bt)
        if out is None:
            return res.contiguous()
        out.copy_(res + beta * out if beta != 0.0 else res)
        return out

    def gemm_colscale(self, a, b, colscale=None, *, want_colss=False, a_kmajor=True, b_kmajor=True, tri_k_lower=False):
        A = a if a_kmajor else a.transpose(-1, -2)
        Bt = b.transpose(-1, -2) if b_kmajor else b
        res = A @ Bt
        ss = (res * res).sum(-2) if want_colss else None
        if colscale is not None:
            res = res * colscale.reshape(1, -1)
        return res.contiguous(), ss

    def gemv(self, a, x, *, alpha=1.0, beta=0.0, out=None):
        res = alpha * (a @ x)
        if out is None:
            return res
        out.copy_(res + beta * out if beta != 0.0 else res)
        return out

    def logdet_chol(self, l):
        return 2 * torch.log(torch.diagonal(l, dim1=-2, dim2=-1)).sum(-1)

    def colreduce(self, v, w=None, *, want_dot=False, want_ss=True):
        dot = ss = None
        if want_dot:
            dot = (v * w.reshape(v.shape[:-1] + (1,))).sum(-2)
        if want_ss:
            ss = (v * v).sum(-2)
        return dot, ss

    def kmat_vjp_dense(self, terms, x, y, g, colscale=None, w=None, b=None, want_colsum=False, want_gradx=False):
        xs, ys, Ge = _np(x), _np(y), _np(g).copy()
        if colscale is not None:
            Ge = Ge * _np(colscale)[None, :]
        if w is not None:
            Ge = Ge + _np(w)[:, None] * _np(b)[None, :]
        S, kfull = [], np.zeros_like(Ge)
        gx = np.zeros_like(xs)
        for kind, var, scale in terms.terms:
            diff = xs[:, None, :] - ys[None, :, :]
            q = (xs @ ys.T if kind == "linear" else (diff ** 2).sum(-1)) / scale ** 2
            if kind == "eq":
                k = np.exp(-0.5 * q); dk = -0.5 * k
            elif kind == "matern12":
                r = np.sqrt(q); k = np.exp(-r)
                with np.errstate(divide="ignore", invalid="ignore"):
                    dk = np.where(r > 0, -0.5 * k / r, 0.0)
            elif kind == "matern32":
                s_ = np.sqrt(3 * q); k = (1 + s_) * np.exp(-s_); dk = -1.5 * np.exp(-s_)
            elif kind == "matern52":
                s_ = np.sqrt(5 * q); k = (1 + s_ + s_ * s_ / 3) * np.exp(-s_); dk = -(5.0 / 6.0) * (1 + s_) * np.exp(-s_)
            elif kind == "linear":
                k = q; dk = np.ones_like(q)
            else:
                k = np.ones_like(q); dk = np.zeros_like(q)
            dkq = -0.5 * np.sqrt(q) * k if kind == "matern12" else dk * q
            S.append([np.sum(Ge * k), np.sum(Ge * dkq)])
            kfull += var * k
            if kind == "linear":
                gx += (Ge * var / scale ** 2) @ ys
            elif kind != "const":
                coef = Ge * var * dk * 2 / scale ** 2
                gx += coef.sum(1)[:, None] * xs - coef @ ys
        return (self._t(np.array(S).reshape(len(terms), 2), x),
                self._t((Ge * kfull).sum(0), x) if want_colsum else None,
                self._t(gx, x) if want_gradx else None)

    def kmat_vjp(self, terms, x, kinv, alpha, g):
        xs = _np(x)
        ki = np.tril(_np(kinv)) + np.tril(_np(kinv), -1).T
        A = _np(alpha)
        gv = np.asarray(g, dtype=np.float64)
        G = 0.5 * ((A * gv) @ A.T - gv.sum() * ki)
        S = []
        for kind, _, scale in terms.terms:
            if kind == "linear":
                q = (xs / scale) @ (xs / scale).T
            else:
                q = O.pw_dists2(xs / scale, xs / scale)
            eps = 1e-300
            if kind == "eq":
                k, dkq = np.exp(-0.5 * q), -0.5 * q * np.exp(-0.5 * q)
            elif kind == "matern12":
                r = np.sqrt(q); k, dkq = np.exp(-r), -0.5 * r * np.exp(-r)
            elif kind == "matern32":
                s = np.sqrt(3 * q); k, dkq = (1 + s) * np.exp(-s), -0.5 * s * s * np.exp(-s)
            elif kind == "matern52":
                s = np.sqrt(5 * q); k, dkq = (1 + s + s * s / 3) * np.exp(-s), -(s * s / 6) * (1 + s) * np.exp(-s)
            elif kind == "linear":
                k, dkq = q, q
            else:
                k, dkq = np.ones_like(q), np.zeros_like(q)
            S.append([np.sum(G * k), np.sum(G * dkq)])
        return (self._t(np.array(S).reshape(len(terms), 2), x), self._t(np.trace(G), x),
                self._t(np.diag(G).copy(), x))

    def tril_(self, a):
        a.copy_(torch.tril(a))
        return a

    def sum_lower(self, parts, out):
        out.copy_(torch.tril(parts.sum(0)))
        return out

    def symmetrize_(self, a):
        a.copy_(torch.tril(a) + torch.tril(a, -1).transpose(-1, -2))
        return a

    def add_diag_(self, a, s=0.0, v=None):
        d = torch.diagonal(a, dim1=-2, dim2=-1)
        d += s
        if v is not None:
            d += v.reshape(d.shape)
        return a

    def scale_cols_(self, v, s):
        v *= s.reshape(v.shape[:-2] + (1, v.shape[-1]))
        return v

    def copy(self, src):
        return src.clone(memory_format=torch.contiguous_format)


@pytest.fixture()
def oracle_backend():
    """Install the test-only CPU backend for one test."""
    prev = ops.set_backend(OracleBackend())
    yield
    ops.set_backend(prev)


@pytest.fixture()
def hip_backend():
    """Make sure the real HIP backend is active (GPU tests)."""
    prev = ops.set_backend(None)
    be = ops.get_backend()
    assert be.name == "hip"
    yield be
    ops.set_backend(prev)


#: device new test tensors go to under ``any_backend`` ("cpu" with the oracle backend, "cuda" with the HIP one)
DEVICE = ["cpu"]


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def any_backend(request):
    """Run a test twice: over the test-only oracle backend on the CPU box, over ``libgpk.so`` on the MI355X."""
    if request.param == "oracle":
        prev = ops.set_backend(OracleBackend())
        DEVICE[0] = "cpu"
        yield request.param
        ops.set_backend(prev)
    else:
        prev = ops.set_backend(None)
        assert ops.get_backend().name == "hip"
        DEVICE[0] = "cuda"
        try:
            yield request.param
        finally:
            DEVICE[0] = "cpu"
            ops.set_backend(prev)


def T(a, dtype=torch.float64):
    """Tensor on the device of the active ``any_backend``."""
    return torch.as_tensor(np.asarray(a), dtype=dtype, device=DEVICE[0])


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=False)
