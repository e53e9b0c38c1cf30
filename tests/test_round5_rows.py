"""Round 5: the posterior's cross-covariance carried THROUGH the factorisation (``gpk_potrf_rows``, ``KernelDense.chol_with_rows``).

When a process is conditioned and queried before anything has factorised the observations' kernel matrix, ``K(x*, x)`` is written under
the kernel matrix in one buffer and the factorisation carries those rows through its panel solves and trailing updates: they come out
as ``K(x*, x) L^{-T}`` and the separate many-column triangular solve is gone.  The reference computes the same quantities as
``cholesky`` + ``solve(L, K_zx)`` inside mlkernels' ``PosteriorKernel`` / ``PosteriorMean`` (``stheno/model/observations.py:148-168``).

Checked here, on the MI355X through ``libgpk.so``: that the path RUNS (``Chol.rows_under``), against the oracle (fp64 1e-6, fp32 1e-3),
against the unfused path on the same device (call order swapped), plain and look-ahead orders, full covariance, a log-density after
the posterior (shares the factor), the full-size cfg2 golden in this call order."""
import json
import os

import numpy as np
import pytest
import torch

import stheno_amd as st
from bench import NOISE, make_inputs
from oracle import gp_oracle as O
from stheno_amd import B, matrix

from .conftest import ROOT

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("hip_backend")]
DEV = torch.device("cuda")


def _rel(a, ref):
    a, ref = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(ref, dtype=np.float64).reshape(-1)
    return float(np.max(np.abs(a - ref)) / np.max(np.abs(ref)))


def _data(n, d, ns, dtype, seed=0):
    rng = np.random.default_rng(seed)
    x, xs = rng.standard_normal((n, d)), rng.standard_normal((ns, d))
    y = np.sin(x.sum(-1, keepdims=True)) + 0.1 * rng.standard_normal((n, 1))
    return tuple(a.astype(dtype) for a in (x, y, xs))


@pytest.mark.parametrize("n,ns,kind", [(2048, 200, "eq"), (2560, 64, "eq+linear"), (4096, 333, "matern52"), (12288, 512, "eq")])
def test_posterior_first_rides_in_the_factorisation_fp64(n, ns, kind):
    x, y, xs = _data(n, 3, ns, np.float64)
    kernel = {"eq": st.EQ(), "eq+linear": st.EQ() + st.Linear(), "matern52": st.Matern52()}[kind]
    terms = {"eq": [("eq", 1.0, 1.0)], "eq+linear": [("eq", 1.0, 1.0), ("linear", 1.0, 1.0)], "matern52": [("matern52", 1.0, 1.0)]}[kind]
    tx, ty, txs = (torch.as_tensor(a, device=DEV) for a in (x, y, xs))
    f = st.GP(kernel)
    fdd = f(tx, NOISE)
    post = f | (fdd, ty)
    mean, var = post(txs).marginals()                    # conditioning + prediction FIRST: nothing has factorised K yet
    chol = fdd.var.chol()
    assert chol.rows_under == ns, "the factorisation with rows under the matrix did not run"
    assert (chol.lookahead_nb == 1024) == (n >= matrix.config.potrf_lookahead_from)
    lp = float(fdd.logpdf(ty))                           # ... and the log-density afterwards shares the factor
    assert fdd.var.chol() is chol
    ref_lp = O.gp_logpdf(terms, x, NOISE, y)
    ref_mean, ref_cov, ref_var = O.gp_posterior(terms, x, NOISE, y, xs, full_cov=True)
    tol = 1e-6 if kind != "matern52" else 2e-6          # (Matern: direct-difference distances vs the oracle's |a|^2 + |b|^2 - 2ab)
    assert abs(lp - ref_lp) <= 1e-6 * abs(ref_lp)
    assert _rel(mean.cpu().numpy(), ref_mean) <= tol and _rel(var.cpu().numpy(), ref_var) <= tol
    # the full posterior covariance from the transposed form (Z Z^T on the k-contiguous GEMM)
    f2 = st.GP(kernel)
    fdd2 = f2(tx, NOISE)
    cov = B.dense((f2 | (fdd2, ty))(txs).var)
    assert fdd2.var.chol().rows_under == ns
    assert _rel(cov.cpu().numpy(), ref_cov) <= tol
    # the unfused path on the same device (log-density first: the factor exists before the posterior is asked for)
    f3 = st.GP(kernel)
    fdd3 = f3(tx, NOISE)
    lp3 = float(fdd3.logpdf(ty))
    mean3, var3 = (f3 | (fdd3, ty))(txs).marginals()
    assert fdd3.var.chol().rows_under == 0
    assert abs(lp - lp3) <= 1e-10 * abs(lp3)
    assert _rel(mean.cpu().numpy(), mean3.cpu().numpy()) <= 1e-9 and _rel(var.cpu().numpy(), var3.cpu().numpy()) <= 1e-9


def test_posterior_first_fp32_against_the_oracle():
    n, ns = 4096, 256
    x, y, xs = _data(n, 4, ns, np.float32, seed=3)
    eps0 = B.epsilon
    try:
        B.epsilon = 1e-6
        tx, ty, txs = (torch.as_tensor(a, device=DEV) for a in (x, y, xs))
        f = st.GP(st.EQ() + st.Linear())
        fdd = f(tx, NOISE)
        mean, var = (f | (fdd, ty))(txs).marginals()
        assert fdd.var.chol().rows_under == ns
        lp = float(fdd.logpdf(ty))
        terms = [("eq", 1.0, 1.0), ("linear", 1.0, 1.0)]
        x64, y64, xs64 = (a.astype(np.float64) for a in (x, y, xs))
        ref_lp = O.gp_logpdf(terms, x64, NOISE, y64, eps=1e-6)
        ref_mean, _, ref_var = O.gp_posterior(terms, x64, NOISE, y64, xs64, eps=1e-6, full_cov=False)
        assert abs(lp - ref_lp) <= 1e-3 * abs(ref_lp)
        assert _rel(mean.cpu().numpy(), ref_mean) <= 1e-3 and _rel(var.cpu().numpy(), ref_var) <= 1e-3
    finally:
        B.epsilon = eps0


def test_shapes_the_native_path_does_not_take_fall_back():
    # too few test points, more test points than observations: the separate solve as before
    for n, ns in ((2048, 8), (2048, 2304)):
        x, y, xs = _data(n, 2, ns, np.float64, seed=5)
        tx, ty, txs = (torch.as_tensor(a, device=DEV) for a in (x, y, xs))
        f = st.GP(st.EQ())
        fdd = f(tx, NOISE)
        mean, var = (f | (fdd, ty))(txs).marginals()
        assert fdd.var.chol().rows_under == 0
        ref_mean, _, ref_var = O.gp_posterior([("eq", 1.0, 1.0)], x, NOISE, y, xs, full_cov=False)
        assert _rel(mean.cpu().numpy(), ref_mean) <= 1e-6 and _rel(var.cpu().numpy(), ref_var) <= 1e-6


@pytest.mark.parametrize("n,ns,dtype", [(2100, 200, np.float64), (4001, 77, np.float64), (2177, 2177, np.float32), (11300, 333, np.float64)])
def test_orders_that_are_no_multiple_of_128_are_padded_into_the_rows_path(n, ns, dtype):
    """Round 6: ``KernelDense.chol_with_rows`` pads the order to whole 128-blocks with the identity (zero columns
    under it), the factor and the whitened rows handed on are views of the padded buffer.  Plain panels and the look-ahead; the
    posterior, the full covariance (the k-contiguous product on strided views) and the log-density that shares the padded factor."""
    fp64 = dtype == np.float64
    x, y, xs = _data(n, 3, ns, dtype, seed=11)
    tx, ty, txs = (torch.as_tensor(a, device=DEV) for a in (x, y, xs))
    eps0 = B.epsilon
    try:
        B.epsilon = 1e-12 if fp64 else 1e-6
        f = st.GP(st.EQ())
        fdd = f(tx, NOISE)
        post = f | (fdd, ty)
        mean, var = post(txs).marginals()
        chol = fdd.var.chol()
        assert chol.rows_under == ns and chol.n == n
        if n >= 11264:
            assert chol.lookahead_nb == 1024
        lp = float(fdd.logpdf(ty))
        x64, y64, xs64 = (a.astype(np.float64) for a in (x, y, xs))
        ref_lp = O.gp_logpdf([("eq", 1.0, 1.0)], x64, NOISE, y64, eps=B.epsilon)
        ref_mean, ref_cov, ref_var = O.gp_posterior([("eq", 1.0, 1.0)], x64, NOISE, y64, xs64, eps=B.epsilon, full_cov=ns <= 400)
        tol = 1e-6 if fp64 else 1e-3
        assert abs(lp - ref_lp) <= tol * abs(ref_lp)
        assert _rel(mean.cpu().numpy(), ref_mean) <= tol and _rel(var.cpu().numpy(), ref_var) <= tol
        if ns <= 400:
            f2 = st.GP(st.EQ())
            cov = B.dense((f2 | (f2(tx, NOISE), ty))(txs).var)
            assert _rel(cov.cpu().numpy(), ref_cov) <= tol
    finally:
        B.epsilon = eps0


def test_not_positive_definite_is_reported_through_the_rows_path():
    n, ns = 2048, 128
    x, y, xs = _data(n, 1, ns, np.float64, seed=7)
    tx, ty, txs = (torch.as_tensor(a, device=DEV) for a in (x, y, xs))
    f = st.GP(st.EQ())
    with pytest.raises(torch.linalg.LinAlgError):
        eps0 = B.epsilon
        try:
            B.epsilon = 0.0
            (f | (f(tx, -5.0), ty))(txs).marginals()        # a negative "noise": not positive-definite
        finally:
            B.epsilon = eps0


def test_config2_full_size_golden_posterior_first():
    with open(os.path.join(ROOT, "tests", "golden", "cfg2_n16384.json")) as fh:
        g = json.load(fh)
    w, t = make_inputs("dense_f64", DEV)
    f = st.GP(st.EQ())
    fdd = f(t["x"], NOISE)
    mean, var = (f | (fdd, t["y"]))(t["xs"]).marginals()
    assert fdd.var.chol().rows_under == 2048 and fdd.var.chol().lookahead_nb == 1024
    lp = float(fdd.logpdf(t["y"]))
    assert abs(lp - g["logpdf"]) <= 1e-6 * abs(g["logpdf"])
    for got, ref in ((mean, g["posterior_mean_all"]), (var, g["posterior_var_all"])):
        got = got.cpu().numpy().reshape(-1)
        assert abs(got.sum() - ref["sum"]) <= 1e-6 * ref["sum_abs"]
        assert np.max(np.abs(got[::64] - np.array(ref["every_64th"]))) <= 1e-6 * ref["max_abs"]
        assert abs(np.abs(got).max() - ref["max_abs"]) <= 1e-6 * ref["max_abs"]


@pytest.mark.parametrize("method", ["vfe", "dtc"])
def test_pseudo_point_bound_with_the_padded_transposed_cross_covariance(method):
    """cfg5's shape class (many more observations than inducing points, M a multiple of 128): ``k(x_pad, z)`` transposed and padded to
    whole tiles, the padding columns switched off by a zero column scale (``observations.py:_compute``) -- the bound and ``mu`` against
    the oracle, and against the same call with the ordinary cross-covariance."""
    rng = np.random.default_rng(21)
    n, m, d = 3000, 256, 3            # 3000 is not a multiple of 128: 72 padding columns
    x, z = rng.standard_normal((n, d)), rng.standard_normal((m, d))
    y = np.sin(x.sum(-1, keepdims=True)) + 0.1 * rng.standard_normal((n, 1))
    ref = O.pseudo_obs([("eq", 1.0, 1.0)], x, NOISE, y, z, method=method, eps=1e-10)
    eps0 = B.epsilon
    try:
        B.epsilon = 1e-10
        got = {}
        for flag in (True, False):
            matrix.config.pseudo_padded_transposed = flag
            prior = st.Measure()
            f = st.GP(st.EQ(), measure=prior)
            tx, ty, tz = (torch.as_tensor(a, device=DEV) for a in (x, y, z))
            obs = {"vfe": st.PseudoObs, "dtc": st.PseudoObsDTC}[method](f(tz), f(tx, NOISE), ty)
            mu = obs.mu(prior)
            got[flag] = (float(obs.elbo(prior)), (mu.mat if hasattr(mu, "mat") else mu).reshape(-1).cpu().numpy())
    finally:
        matrix.config.pseudo_padded_transposed = True
        B.epsilon = eps0
    for flag in (True, False):
        assert abs(got[flag][0] - ref["elbo"]) <= 1e-6 * abs(ref["elbo"]), (flag, got[flag][0], ref["elbo"])
        assert _rel(got[flag][1], ref["mu"]) <= 1e-6
    assert abs(got[True][0] - got[False][0]) <= 1e-10 * abs(got[False][0])
