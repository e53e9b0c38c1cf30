"""Round-3 host-side behaviour, on the CPU box over the oracle backend and on the MI355X over ``libgpk.so``:

 * the NaN scan of the observations is remembered per tensor VERSION (``matrix.any_missing``; reference semantics of the scan:
   ``stheno/random.py:262-264``, ``stheno/model/observations.py:73-74``);
 * ``deferred_checks()``: a failed factorisation is reported when the block ends, never later, and nothing is reported early;
 * the dense posterior lets the blocked solve use the cross matrix ``k(x, x*)`` as its workspace (``own_cross``): same numbers as
   the copying path, and the pseudo-point posterior (two matrices solved against the same cross matrix) keeps the copy;
 * ``Normal.computed`` answers without computing.
"""
import numpy as np
import pytest
import torch

import stheno_amd as st
from oracle import gp_oracle as O
from stheno_amd import kernels, matrix

from .conftest import T

pytestmark = pytest.mark.usefixtures("any_backend")
f64 = torch.float64


def test_nan_scan_is_remembered_until_the_tensor_changes(monkeypatch):
    calls = []
    real = torch.isnan

    def counting(t, *a, **k):
        calls.append(tuple(t.shape))
        return real(t, *a, **k)

    monkeypatch.setattr(torch, "isnan", counting)
    rng = np.random.default_rng(0)
    x, y = T(rng.standard_normal((40, 2)), f64), T(rng.standard_normal((40, 1)), f64)
    f = st.GP(st.EQ())
    fdd = f(x, 0.1)
    lp = fdd.logpdf(y)
    post = f | (fdd, y)
    n_first = len(calls)
    assert n_first == 1                                  # the log-density scanned, the conditioning on the same y did not
    f2 = st.GP(st.EQ())
    f2(x, 0.1).logpdf(y)
    assert len(calls) == n_first                         # another evaluation on the same data: no new scan
    y[3, 0] = float("nan")                               # an in-place write bumps the version: scanned again, and found
    lp_nan = f2(x, 0.1).logpdf(y)
    assert len(calls) > n_first
    keep = np.arange(40) != 3
    ref = O.gp_logpdf([("eq", 1.0, 1.0)], x.cpu().numpy()[keep], 0.1, np.nan_to_num(y.cpu().numpy())[keep])
    assert abs(float(lp_nan) - ref) < 1e-8 * abs(ref) and np.isfinite(float(lp)) and post is not None


def test_deferred_checks_report_at_the_end_of_the_block():
    bad = T(np.array([[1.0, 2.0], [2.0, 1.0]]), f64)      # not positive definite
    good = T(np.array([[2.0, 0.5], [0.5, 1.0]]), f64)
    with pytest.raises(torch.linalg.LinAlgError):
        matrix.Dense(bad.clone()).chol()                  # outside a block: at once
    reached = []
    with pytest.raises(torch.linalg.LinAlgError):
        with matrix.deferred_checks():
            c = matrix.Dense(bad.clone()).chol()          # no exception here ...
            reached.append(c)
            with matrix.deferred_checks():                # ... nor at the end of a nested block
                matrix.Dense(good.clone()).chol()
            reached.append("inner block left")
    assert len(reached) == 2                              # ... but when the outermost block ends
    with matrix.deferred_checks():
        ok = matrix.Dense(good.clone()).chol()
    assert float(ok.logdet()) == pytest.approx(np.log(np.linalg.det(good.cpu().numpy())), rel=1e-9)      # (B.epsilon on the diagonal)
    assert matrix._deferred is None


def test_dense_posterior_consumes_the_cross_matrix_and_the_pseudo_point_posterior_does_not(monkeypatch):
    rng = np.random.default_rng(1)
    n, ns = 60, 24
    x, xs = rng.standard_normal((n, 3)), rng.standard_normal((ns, 3))
    y = rng.standard_normal((n, 1))
    seen = []
    real = kernels._whiten

    def spy(cache, K_z, k, z, xx, own_cross=False):
        seen.append(own_cross)
        return real(cache, K_z, k, z, xx, own_cross)

    monkeypatch.setattr(kernels, "_whiten", spy)
    f = st.GP(st.EQ())
    post = f | (f(T(x, f64), 0.1), T(y, f64))
    mean, var = post(T(xs, f64)).marginals()
    assert seen and all(seen)                             # dense conditioning: the solve owns the cross matrix
    ref_mean, _, ref_vd = O.gp_posterior([("eq", 1.0, 1.0)], x, 0.1, y, xs, full_cov=False)
    assert np.max(np.abs(mean.cpu().numpy() - ref_mean)) < 1e-9 and np.max(np.abs(var.cpu().numpy() - ref_vd)) < 1e-9
    # the full covariance (pairwise: two whitenings of the same key) agrees with the marginal variances
    cov = st.B.dense(post(T(xs, f64)).var)
    assert np.max(np.abs(np.diag(cov.cpu().numpy()) - ref_vd)) < 1e-9
    seen.clear()
    z = T(x[::6], f64)
    fz = st.GP(st.EQ())
    sparse = fz | st.PseudoObs(fz(z), fz(T(x, f64), 0.1), T(y, f64))
    sparse(T(xs, f64)).marginals()
    assert seen and not any(seen)                         # K_z and A are both solved against k(z, x*): it is kept


def test_normal_computed_does_not_compute():
    made = []

    def mean():
        made.append("mean")
        return torch.ones(3, 1, dtype=f64)

    def var():
        made.append("var")
        return torch.eye(3, dtype=f64)

    d = st.Normal(mean, var)
    assert not d.computed("mean") and not d.computed("var") and not d.computed("var_diag") and made == []
    d.var_diag
    assert made == ["var"] and d.computed("var") and d.computed("var_diag") and not d.computed("mean")
