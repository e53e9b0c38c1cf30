"""Pin the CPU oracle (oracle/gp_oracle.py) before anything is compared against it:
the reference's own known answers and checks, restated.

* README known answers (kernel matrix, logpdf, posterior) -- README.md:43-86,470-497
* ``Normal.logpdf`` == SciPy's multivariate normal -- tests/test_random.py:185-192
* inducing points == inputs => ELBO == logpdf, approximate == exact posterior for
  VFE/FITC/DTC -- tests/model/test_model.py:283-308
* README ELBO gap order of magnitude -- README.md:703-720
* marginals == diagonal of the full posterior -- tests/model/test_fdd.py:111-134
* the committed golden fixtures are what the oracle produces (generator is deterministic)
* every kernel formula against an independent definition: the Matern family through the general
  Bessel-function form (scipy.special.kv / gamma), Linear through explicit dot products, EQ through
  explicit squared differences (the test kernel of tests/model/test_model.py:342-350)
* the VFE / FITC / DTC bounds against DENSE textbook expressions that share no algebra with the
  factorised restatement (log N(y | 0, Q + ...) by slogdet / solve of the N x N matrix, trace term explicit)
"""
import json
import os

import numpy as np
import pytest
from scipy.special import gamma, kv
from scipy.stats import multivariate_normal

from oracle import gp_oracle as O

from .conftest import ROOT, golden

EQ = [("eq", 1.0, 1.0)]


@pytest.fixture(scope="module")
def kats():
    with open(os.path.join(ROOT, "tests", "golden", "readme_kats.json")) as f:
        return json.load(f)


def test_readme_kernel_matrix(kats):
    k = kats["eq_matrix_x012"]
    np.testing.assert_allclose(O.kernel_matrix(EQ, np.array(k["x"])), np.array(k["k"]), atol=5e-4)


def test_readme_logpdf(kats):
    k1, k2 = kats["logpdf_y1"], kats["logpdf_y2"]
    # printed y has 8 digits: agreement to ~1e-8 relative
    assert abs(O.gp_logpdf(EQ, np.array(k1["x"]), None, np.array(k1["y"])) - k1["logpdf"]) < 5e-8
    np.testing.assert_allclose(O.gp_logpdf(EQ, np.array(k2["x"]), None, np.array(k2["y"])), k2["logpdf"], rtol=2e-8)


def test_readme_posterior_needs_default_epsilon(kats):
    k = kats["posterior_20s"]
    x = np.linspace(*k["x_linspace"][:2], int(k["x_linspace"][2]))
    mean, var, var_diag = O.gp_posterior(EQ, x, None, x**2, np.array(k["x_new"]), eps=k["epsilon"])
    np.testing.assert_allclose(mean, k["mean"], rtol=2e-7)
    # kappa(K) ~ 1/eps: the well-determined entry agrees to a few 1e-6
    assert abs(var[2, 2] - k["var"][2][2]) / k["var"][2][2] < 2e-5
    assert abs(var_diag[2] - var[2, 2]) < 1e-12
    # without the jitter the README value is not reproduced
    mean0, _, _ = O.gp_posterior(EQ, x, None, x**2, np.array(k["x_new"]), eps=0.0)
    assert abs(mean0[2] - k["mean"][2]) > 1e-3


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_logpdf_matches_scipy(seed):
    rng = np.random.default_rng(seed)
    mean = rng.standard_normal((3, 1))
    chol = rng.standard_normal((3, 3))
    var = chol @ chol.T
    x = rng.standard_normal((3, 10))
    ref = multivariate_normal(mean[:, 0], var).logpdf(x.T)
    np.testing.assert_allclose(O.normal_logpdf(mean, var, x, eps=0.0), ref, rtol=1e-6)
    assert np.shape(O.normal_logpdf(mean, var, np.ones((3, 1)))) == ()
    assert np.shape(O.normal_logpdf(mean, var, np.ones((3, 2)))) == (2,)


@pytest.mark.parametrize("method", ["vfe", "fitc", "dtc"])
def test_pseudo_points_equal_inputs_is_exact(method):
    rng = np.random.default_rng(3)
    x = np.linspace(0, 5, 12)
    noise = rng.uniform(0.1, 0.5, 12)
    terms = [("eq", 1.0, 1.0), ("matern12", 2.0, 1.0)]
    k = O.kernel_matrix(terms, x) + np.diag(noise)
    y = np.linalg.cholesky(k) @ rng.standard_normal((12, 1))
    xs = np.linspace(0, 5, 7)
    exact = O.gp_logpdf(terms, x, noise, y, eps=1e-12)
    r = O.pseudo_obs(terms, x, noise, y, x, method=method, eps=1e-12)
    np.testing.assert_allclose(r["elbo"], exact, atol=1e-8, rtol=1e-8)
    m_e, v_e, vd_e = O.gp_posterior(terms, x, noise, y, xs)
    m_a, v_a, vd_a = O.pseudo_posterior(terms, x, noise, y, x, xs, method=method)
    np.testing.assert_allclose(m_a, m_e, atol=1e-7)
    np.testing.assert_allclose(v_a, v_e, atol=1e-7)
    np.testing.assert_allclose(vd_a, vd_e, atol=1e-7)


def test_readme_elbo_gap_magnitude(kats):
    k = kats["elbo_gap"]
    rng = np.random.default_rng(0)
    x = np.linspace(0, 10, k["n"])
    z = np.linspace(0, 10, k["m"])
    y = np.linalg.cholesky(O.kernel_matrix(EQ, x) + k["noise"] * np.eye(k["n"])) @ rng.standard_normal((k["n"], 1))
    gap = O.pseudo_obs(EQ, x, k["noise"], y, z)["elbo"] - O.gp_logpdf(EQ, x, k["noise"], y)
    assert -5e-9 < gap <= 1e-10      # README: -3.5e-10 (different random y)


def test_marginals_are_posterior_diagonal():
    g = golden("dense_eq_n256_d8.npz")
    np.testing.assert_allclose(np.diag(g["post_var"]), g["post_var_diag"], atol=1e-10)
    mean, lo, hi = O.credible_bounds(g["post_mean"], g["post_var_diag"])
    assert np.all(lo <= mean) and np.all(mean <= hi)


def test_batched_is_a_loop():
    g = golden("batched_eq_b16_n100_d3.npz")
    terms = list(zip(g["kinds"], g["variances"], g["scales"]))
    out = O.gp_logpdf_batched(terms, g["x"], float(g["noise"]), g["y"])
    assert out.shape == (16,)
    np.testing.assert_allclose(out, g["logpdf"], rtol=1e-12)


@pytest.mark.parametrize("name", ["dense_eq_n256_d8", "dense_matern32_n200_d3", "dense_eq_linear_n512_d4"])
def test_golden_fixtures_reproducible(name):
    g = golden(name + ".npz")
    terms = list(zip(g["kinds"], g["variances"], g["scales"]))
    np.testing.assert_allclose(np.atleast_1d(O.gp_logpdf(terms, g["x"], float(g["noise"]), g["y"])), g["logpdf"], rtol=1e-10)
    mean, _, vd = O.gp_posterior(terms, g["x"], float(g["noise"]), g["y"][:, :1], g["xs"], full_cov=False)
    np.testing.assert_allclose(mean, g["post_mean"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(vd, g["post_var_diag"], rtol=1e-9, atol=1e-12)


# ------------------------------------------------------------------ independent definitions of every kernel
def _matern_general(r, nu):
    """2^{1-nu} / Gamma(nu) (sqrt(2 nu) r)^nu K_nu(sqrt(2 nu) r): the textbook Matern (Rasmussen & Williams eq. 4.14)."""
    r = np.asarray(r, dtype=np.float64)
    out = np.ones_like(r)
    nz = r > 0
    a = np.sqrt(2 * nu) * r[nz]
    out[nz] = 2 ** (1 - nu) / gamma(nu) * a**nu * kv(nu, a)
    return out


@pytest.mark.parametrize("kind,nu", [("matern12", 0.5), ("matern32", 1.5), ("matern52", 2.5)])
def test_matern_kernels_are_the_bessel_family(kind, nu):
    rng = np.random.default_rng(11)
    x, y = rng.standard_normal((40, 3)), rng.standard_normal((25, 3))
    var, scale = 1.7, 0.6
    r = np.sqrt(((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)) / scale
    np.testing.assert_allclose(O.kernel_matrix([(kind, var, scale)], x, y), var * _matern_general(r, nu), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(O.kernel_diag([(kind, var, scale)], x), var * np.ones(40), rtol=1e-12)


def test_eq_and_linear_against_explicit_loops():
    rng = np.random.default_rng(12)
    x, y = rng.standard_normal((17, 4)), rng.standard_normal((9, 4))
    var, scale = 0.9, 1.4
    eq = np.array([[var * np.exp(-0.5 * sum((a - b) ** 2 for a, b in zip(xi, yj)) / scale**2) for yj in y] for xi in x])
    np.testing.assert_allclose(O.kernel_matrix([("eq", var, scale)], x, y), eq, rtol=1e-11)
    lin = np.array([[var * sum(a * b for a, b in zip(xi, yj)) / scale**2 for yj in y] for xi in x])
    np.testing.assert_allclose(O.kernel_matrix([("linear", var, scale)], x, y), lin, rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(O.kernel_diag([("linear", var, scale)], x), var * (x * x).sum(1) / scale**2, rtol=1e-12)
    # sums and the constant kernel
    both = O.kernel_matrix([("eq", var, scale), ("linear", 2.0, 1.0), ("const", 0.3, 1.0)], x, y)
    np.testing.assert_allclose(both, eq + 2.0 * x @ y.T + 0.3, rtol=1e-11)


@pytest.mark.parametrize("terms", [[("eq", 1.0, 1.0)], [("matern32", 1.5, 0.8), ("linear", 0.5, 1.0)], [("matern52", 0.7, 1.3)]])
@pytest.mark.parametrize("method", ["vfe", "fitc", "dtc"])
def test_sparse_bounds_against_dense_textbook_expressions(terms, method):
    """Titsias (2009) eq. 9 / Snelson & Ghahramani (2006) / Seeger et al. (2003), evaluated with the N x N matrices:
    VFE  log N(y | 0, Q + s2 I) - tr(K - Q) / (2 s2);  FITC  log N(y | 0, Q + diag(K - Q) + s2 I);  DTC  log N(y | 0, Q + s2 I),
    Q = K_xz K_z^{-1} K_zx -- no Cholesky of K_z, no whitening, no Woodbury: np.linalg.solve / slogdet on dense matrices."""
    rng = np.random.default_rng(13)
    n, m, noise = 60, 9, 0.3
    x, z = rng.uniform(0, 3, (n, 2)), rng.uniform(0, 3, (m, 2))
    y = rng.standard_normal((n, 1))
    kxx, kzz, kzx = O.kernel_matrix(terms, x), O.kernel_matrix(terms, z), O.kernel_matrix(terms, z, x)
    q = kzx.T @ np.linalg.solve(kzz, kzx)
    cov = q + noise * np.eye(n)
    trace = 0.0
    if method == "fitc":
        cov = cov + np.diag(np.diag(kxx - q))
    if method == "vfe":
        trace = np.trace(kxx - q) / (2 * noise)
    sign, logdet = np.linalg.slogdet(cov)
    want = -0.5 * (logdet + n * np.log(2 * np.pi) + (y.T @ np.linalg.solve(cov, y)).item()) - trace
    got = O.pseudo_obs(terms, x, noise, y, z, method=method, eps=0.0)["elbo"]
    assert sign > 0 and abs(got - want) <= 1e-9 * abs(want)
    # the approximate posterior against the dense predictive equations of the same papers
    xs = rng.uniform(0, 3, (7, 2))
    lam = noise * np.ones(n) + (np.diag(kxx - q) if method == "fitc" else 0.0)
    sigma = np.linalg.inv(kzz + (kzx / lam) @ kzx.T)
    kzs = O.kernel_matrix(terms, z, xs)
    mean_want = (kzs.T @ sigma @ (kzx / lam) @ y)[:, 0]
    var_want = O.kernel_matrix(terms, xs) - kzs.T @ np.linalg.solve(kzz, kzs) + kzs.T @ sigma @ kzs
    mean, var, vd = O.pseudo_posterior(terms, x, noise, y, z, xs, method=method, eps=0.0)
    np.testing.assert_allclose(mean, mean_want, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(var, var_want, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(vd, np.diag(var_want), rtol=1e-7, atol=1e-9)
