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
    f2 = st.GP(st.EQ())
    f2(x, 0.1).logpdf(y)
    if y.is_cuda:
        assert n_first == 1                              # the log-density scanned, the conditioning on the same y did not
        assert len(calls) == n_first                     # another evaluation on the same data: no new scan
        matrix.forget_scan(y)                            # ... until the caller says the data changed behind torch's back
        f2(x, 0.1).logpdf(y)
        assert len(calls) == n_first + 1
        n_first = len(calls)
    else:
        assert n_first == 2 and len(calls) == 3          # host memory can be written through NumPy: nothing is remembered
        n_first = len(calls)
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
    assert matrix._deferred_state.pending is None


def test_a_failed_factor_stays_failed_in_its_cache():
    """inside ``deferred_checks()`` (``Normal.logpdf``) the factor is cached before it is checked -- the second use of the
    same distribution must raise again, not return NaN; a block left through an exception leaves its factors to their next user."""
    bad = T(np.array([[1.0, 2.0], [2.0, 1.0]]), f64)
    y = T(np.array([[0.3], [-0.2]]), f64)
    d = st.Normal(matrix.Dense(bad.clone()))
    for _ in range(3):
        with pytest.raises(torch.linalg.LinAlgError):
            d.logpdf(y)
    with pytest.raises(torch.linalg.LinAlgError):
        d.var.chol()
    m = matrix.Dense(bad.clone())
    with pytest.raises(KeyError):
        with matrix.deferred_checks():
            m.chol()                                      # queued for the end of the block ...
            raise KeyError("something else went wrong")   # ... which never checks it
    assert matrix._deferred_state.pending is None
    with pytest.raises(torch.linalg.LinAlgError):
        m.chol()                                          # the next user does
    # several failures in one block: every factor is looked at, the first failure is reported, the others stay failed too
    m1, m2 = matrix.Dense(bad.clone()), matrix.Dense(bad.clone())
    with pytest.raises(torch.linalg.LinAlgError):
        with matrix.deferred_checks():
            m1.chol(), m2.chol()
    assert m1._chol._error is not None and m2._chol._error is not None


def test_nan_scan_with_inference_tensors_and_numpy_aliases():
    """inference tensors have no version counter (the scan must not ask for one), and host memory shared with NumPy can
    change without torch noticing (nothing may be remembered about it)."""
    rng = np.random.default_rng(3)
    xh, yh = rng.standard_normal((30, 2)), rng.standard_normal((30, 1))
    with torch.inference_mode():
        x, y = T(xh, f64), T(yh, f64)
        f = st.GP(st.EQ())
        lp = f(x, 0.1).logpdf(y)
        mean = (f | (f(x, 0.1), y))(x).mean
    ref = O.gp_logpdf([("eq", 1.0, 1.0)], xh, 0.1, yh)
    assert abs(float(lp) - ref) < 1e-8 * abs(ref) and torch.isfinite(mean).all()
    shared = yh.copy()
    y_alias = torch.from_numpy(shared)                    # host tensor aliasing a NumPy array
    assert matrix.any_missing(y_alias) is False
    shared[4, 0] = np.nan                                 # written through NumPy: no version bump
    assert matrix.any_missing(y_alias) is True


def test_dense_posterior_consumes_the_cross_matrix_and_the_pseudo_point_posterior_does_not(monkeypatch):
    rng = np.random.default_rng(1)
    n, ns = 60, 24
    x, xs = rng.standard_normal((n, 3)), rng.standard_normal((ns, 3))
    y = rng.standard_normal((n, 1))
    seen = []
    real = kernels._whiten

    def spy(cache, K_z, k, z, xx, own_cross=False, *more):
        seen.append(own_cross)
        return real(cache, K_z, k, z, xx, own_cross, *more)

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


def test_common_length_scale_view_is_not_remembered_past_an_in_place_write():
    """``Sum.input_scaled_view`` caches the comparison of the summands' length-scale vectors -- an in-place write to one of
    them afterwards must be seen (mlkernels' ``Stretched`` has no such cache: ``k(x / l)`` is evaluated with the current ``l``)."""
    rng = np.random.default_rng(5)
    x = T(rng.standard_normal((30, 3)), f64)
    la, lb = T(np.array([0.7, 1.1, 1.9]), f64), T(np.array([0.7, 1.1, 1.9]), f64)
    k = st.EQ().stretch(la) + st.Matern32().stretch(lb)
    assert k.a.scales is la and k.b.scales is lb                # the caller's tensors, not copies (also under a default-device mode)
    assert k.input_scaled_view() is not None                    # equal vectors: one division, one fused launch
    ref1 = O.kernel_matrix([("eq", 1.0, 1.0)], x.cpu().numpy() / la.cpu().numpy()) + \
        O.kernel_matrix([("matern32", 1.0, 1.0)], x.cpu().numpy() / lb.cpu().numpy())
    assert np.max(np.abs(k.pairwise(x).cpu().numpy() - ref1)) < 1e-8
    lb[1] = 2.5                                                  # now they differ
    assert k.input_scaled_view() is None
    ref2 = O.kernel_matrix([("eq", 1.0, 1.0)], x.cpu().numpy() / la.cpu().numpy()) + \
        O.kernel_matrix([("matern32", 1.0, 1.0)], x.cpu().numpy() / lb.cpu().numpy())
    assert np.max(np.abs(k.pairwise(x).cpu().numpy() - ref2)) < 1e-8
    # a default-device mode must not make the kernel copy the caller's vector
    torch.set_default_device(str(la.device))
    try:
        k2 = st.EQ().stretch(lb)
        assert k2.scales is lb
    finally:
        torch.set_default_device(None)
