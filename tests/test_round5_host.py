"""Round 5, host logic on the GPU-less box: the posterior evaluated BEFORE anything factorised the observations' kernel matrix takes the
factorisation with rows under the matrix (``KernelDense.chol_with_rows`` -> ``_whiten`` -> ``WhitenedT`` -> row reductions / the
k-contiguous product), here over the test-only oracle backend's stand-in for ``gpk_potrf_rows``.  Against the oracle's direct formulas
(``stheno/model/observations.py:148-168``) and against the log-density-first order; the MI355X versions are in ``test_round5_rows.py``."""
import numpy as np
import pytest
import torch

import stheno_amd as st
from oracle import gp_oracle as O
from stheno_amd import B, matrix, ops

from .conftest import DEVICE, OracleBackend


@pytest.fixture()
def oracle_backend():
    prev = ops.set_backend(OracleBackend())
    old = (matrix.config.posterior_rows_from, matrix.config.posterior_rows_min_points)
    matrix.config.posterior_rows_from, matrix.config.posterior_rows_min_points = 128, 8
    yield
    matrix.config.posterior_rows_from, matrix.config.posterior_rows_min_points = old
    ops.set_backend(prev)


def _rel(a, ref):
    a, ref = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(ref, dtype=np.float64).reshape(-1)
    return float(np.max(np.abs(a - ref)) / np.max(np.abs(ref)))


@pytest.mark.parametrize("kind", ["eq", "eq+linear", "scaled"])
def test_posterior_first_takes_the_rows_path_and_matches_the_oracle(oracle_backend, kind):
    rng = np.random.default_rng(1)
    n, ns, d = 256, 24, 2
    x, xs = rng.standard_normal((n, d)), rng.standard_normal((ns, d))
    y = np.sin(x.sum(-1, keepdims=True)) + 0.1 * rng.standard_normal((n, 1))
    kernel = {"eq": st.EQ(), "eq+linear": st.EQ() + st.Linear(), "scaled": 2.0 * st.EQ().stretch(0.7)}[kind]
    terms = {"eq": [("eq", 1.0, 1.0)], "eq+linear": [("eq", 1.0, 1.0), ("linear", 1.0, 1.0)], "scaled": [("eq", 2.0, 0.7)]}[kind]
    tx, ty, txs = (torch.as_tensor(a) for a in (x, y, xs))
    f = st.GP(kernel)
    fdd = f(tx, 0.1)
    post = f | (fdd, ty)
    mean, var = post(txs).marginals()
    chol = fdd.var.chol()
    assert chol.rows_under == ns
    lp = float(fdd.logpdf(ty))
    assert fdd.var.chol() is chol
    ref_mean, ref_cov, ref_var = O.gp_posterior(terms, x, 0.1, y, xs, full_cov=True)
    assert abs(lp - O.gp_logpdf(terms, x, 0.1, y)) <= 1e-9 * abs(lp)
    assert _rel(mean.numpy(), ref_mean) <= 1e-9 and _rel(var.numpy(), ref_var) <= 1e-9
    # the full covariance through Z Z^T, and a covariance between two different input sets (one whitened transposed, one not)
    f2 = st.GP(kernel)
    fdd2 = f2(tx, 0.1)
    post2 = f2 | (fdd2, ty)
    assert _rel(B.dense(post2(txs).var).numpy(), ref_cov) <= 1e-9
    assert fdd2.var.chol().rows_under == ns
    kxy = post2.kernel.pairwise(txs, txs[:7])
    assert _rel(kxy.numpy(), ref_cov[:, :7]) <= 1e-9
    # the other order: the factor exists before the posterior is asked for
    f3 = st.GP(kernel)
    fdd3 = f3(tx, 0.1)
    fdd3.logpdf(ty)
    mean3, var3 = (f3 | (fdd3, ty))(txs).marginals()
    assert fdd3.var.chol().rows_under == 0
    assert _rel(mean.numpy(), mean3.numpy()) <= 1e-10 and _rel(var.numpy(), var3.numpy()) <= 1e-10


def test_shapes_outside_the_rows_path_fall_back(oracle_backend):
    rng = np.random.default_rng(2)
    for n, ns, noise in ((200, 24, 0.1), (256, 4, 0.1), (256, 24, None)):       # not a multiple of 128 (round 6: padded, taken); too few points; fine (no noise: jitter only)
        x, xs = rng.standard_normal((n, 1)), rng.standard_normal((ns, 1))
        y = rng.standard_normal((n, 1))
        tx, ty, txs = (torch.as_tensor(a) for a in (x, y, xs))
        f = st.GP(st.EQ())
        eps0 = B.epsilon
        try:
            B.epsilon = 1e-8
            fdd = f(tx) if noise is None else f(tx, noise)
            mean, var = (f | (fdd, ty))(txs).marginals()
            ref_mean, _, ref_var = O.gp_posterior([("eq", 1.0, 1.0)], x, 0.0 if noise is None else noise, y, xs, eps=1e-8, full_cov=False)
        finally:
            B.epsilon = eps0
        assert (fdd.var.chol().rows_under == ns) == (ns >= 8)
        assert _rel(mean.numpy(), ref_mean) <= 1e-6 and np.max(np.abs(var.numpy().reshape(-1) - ref_var.reshape(-1))) <= 1e-6


def test_not_positive_definite_raises_from_the_rows_path(oracle_backend):
    rng = np.random.default_rng(3)
    x, xs, y = rng.standard_normal((128, 1)), rng.standard_normal((16, 1)), rng.standard_normal((128, 1))
    f = st.GP(st.EQ())
    with pytest.raises(torch.linalg.LinAlgError):
        (f | (f(torch.as_tensor(x), -5.0), torch.as_tensor(y)))(torch.as_tensor(xs)).marginals()


def test_mean_as_the_reference_returns_it(any_backend):
    """``config.mean_as_matrix``: ``.mean`` is a ``Dense`` that ``B.dense`` strips (reference README.md:58-68); everything computed
    FROM the mean (log-density, marginals, sampling, arithmetic) is unchanged."""
    from stheno_amd import EQ, GP

    dev = DEVICE[0]
    x = torch.linspace(0, 2, 40, dtype=torch.float64, device=dev)
    f = GP(lambda t: t ** 2, EQ())
    y = torch.sin(x)
    plain = f(x, 0.1)
    want_mean, want_lp = plain.mean, plain.logpdf(y)
    want_m, want_v = plain.marginals()
    matrix.config.mean_as_matrix = True
    try:
        d = f(x, 0.1)
        assert isinstance(d.mean, matrix.Dense)
        assert torch.equal(B.dense(d.mean), want_mean)
        assert torch.equal(d.logpdf(y), want_lp)
        m, v = d.marginals()
        assert torch.equal(m, want_m) and torch.equal(v, want_v)
        mv_m, mv_v = d.mean_var
        assert isinstance(mv_m, matrix.Dense) and isinstance(mv_v, matrix.AbstractMatrix)
        assert torch.equal(B.dense((d + d).mean), 2 * want_mean)
        post = f | (d, y)
        assert isinstance(post(x).mean, matrix.Dense)
    finally:
        matrix.config.mean_as_matrix = False
    assert torch.is_tensor(f(x, 0.1).mean)


def test_an_independent_process_is_not_sent_through_the_rows_path(oracle_backend):
    """the cross-kernel of a process independent of the observed one is the ZERO kernel, whose ``terms()``
    is the empty list -- the rows-under-the-matrix path took it, ``ZeroKernel.pairwise`` ignored ``out=`` and the uninitialised
    rows of the buffer came back as a whitened cross-covariance.  The posterior of an independent process is its prior
    (``stheno/model/measure.py:172-173``: independent processes get ``ZeroKernel`` cross-kernels)."""
    rng = np.random.default_rng(3)
    n, ns, d = 256, 24, 2
    x, xs = rng.standard_normal((n, d)), rng.standard_normal((ns, d))
    y = rng.standard_normal((n, 1))
    tx, ty, txs = (torch.as_tensor(a) for a in (x, y, xs))
    m = st.Measure()
    f = st.GP(st.EQ(), measure=m)
    g = st.GP(2.0 * st.EQ(), measure=m)
    fdd = f(tx, 0.1)
    post = m | (fdd, ty)
    mean, var = post(g)(txs).marginals()
    assert np.allclose(mean.numpy(), 0.0, atol=1e-12) and np.allclose(var.numpy(), 2.0, rtol=1e-12)
    # the observed process afterwards (the factor exists by now: the posterior mean of `g` asked for L^{-1} y) matches the oracle
    mean_f, var_f = post(f)(txs).marginals()
    ref_mean, _, ref_var = O.gp_posterior([("eq", 1.0, 1.0)], x, 0.1, y, xs, full_cov=False)
    assert _rel(mean_f.numpy(), ref_mean) <= 1e-9 and _rel(var_f.numpy(), ref_var) <= 1e-9


def test_zero_kernel_writes_into_the_view_it_is_given(oracle_backend):
    from stheno_amd.kernels import ZeroKernel

    buf = torch.full((6, 5), 7.0, dtype=torch.float64)
    x, y = torch.zeros(3, 2, dtype=torch.float64), torch.zeros(5, 2, dtype=torch.float64)
    out = ZeroKernel().pairwise(x, y, out=buf[2:5])
    assert out.data_ptr() == buf[2:5].data_ptr() and float(buf[2:5].abs().max()) == 0.0 and float(buf[:2].min()) == 7.0
    k = ZeroKernel().pairwise(x, None, diag_add=0.5, diag_vec=torch.ones(3, dtype=torch.float64))
    assert torch.equal(k, 1.5 * torch.eye(3, dtype=torch.float64))
    with pytest.raises(ValueError):
        ZeroKernel().pairwise(x, y, out=buf[:2])


def test_rows_path_is_refused_where_the_native_panels_have_no_room():
    """``can_factor_with_rows`` mirrors the native control-word budget (``gpk_potrf_pipe.hpp``:
    96 + 2 ceil(m / 64) npb words in one 128 x 128 slot of ``dinv``) and caps the rows at the order of the matrix."""
    fit = matrix.rows_panels_fit
    assert fit(4096, 4096, 4, False) and fit(4096, 4096, 8, False)
    assert not fit(4096, 16384, 4, False)           # the advisor's example: n = 4096 fp32, ns = 16384 -> 96 + 2 * 320 * 32 = 20576 > 16384
    assert fit(4096, 16384, 8, False)
    assert fit(32768, 32768, 4, True) and fit(11136, 11136, 4, False)
    assert not fit(32768, 60000, 4, True) and fit(32768, 60000, 8, True)
    k = matrix.KernelDense(st.EQ(), torch.zeros(4096, 2), None)

    class _Be:
        name = "hip"

        def potrf_rows_(self, *a, **kw):
            raise AssertionError

    prev = ops.set_backend(_Be())
    try:
        assert k.can_factor_with_rows(4096) and not k.can_factor_with_rows(4097) and not k.can_factor_with_rows(16384)
    finally:
        ops.set_backend(prev)


def test_a_ragged_order_is_padded_into_the_rows_path(oracle_backend):
    """Round 6: orders that are no multiple of 128 take the factorisation with rows under the matrix too -- ``chol_with_rows`` pads
    with the identity and hands on views (``tests/test_round5_rows.py`` has the MI355X version)."""
    rng = np.random.default_rng(5)
    n, ns, d = 300, 40, 2
    x, xs = rng.standard_normal((n, d)), rng.standard_normal((ns, d))
    y = np.cos(x.sum(-1, keepdims=True)) + 0.1 * rng.standard_normal((n, 1))
    tx, ty, txs = (torch.as_tensor(a) for a in (x, y, xs))
    f = st.GP(st.EQ() + st.Linear())
    fdd = f(tx, 0.1)
    post = f | (fdd, ty)
    mean, var = post(txs).marginals()
    chol = fdd.var.chol()
    assert chol.rows_under == ns and chol.n == n and chol.l.stride(0) == 384
    terms = [("eq", 1.0, 1.0), ("linear", 1.0, 1.0)]
    ref_mean, ref_cov, ref_var = O.gp_posterior(terms, x, 0.1, y, xs, full_cov=True)
    assert _rel(mean.numpy(), ref_mean) <= 1e-9 and _rel(var.numpy(), ref_var) <= 1e-9
    assert abs(float(fdd.logpdf(ty)) - O.gp_logpdf(terms, x, 0.1, y)) <= 1e-9 * abs(O.gp_logpdf(terms, x, 0.1, y))
    f2 = st.GP(st.EQ() + st.Linear())
    assert _rel(B.dense((f2 | (f2(tx, 0.1), ty))(txs).var).numpy(), ref_cov) <= 1e-9
