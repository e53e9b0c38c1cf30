"""Parity of the HIP path (libgpk.so through the C ABI, via stheno_amd) with the CPU
oracle and the committed golden fixtures, on a real MI355X.

Tolerances (BASELINE.json north_star): logpdf / ELBO 1e-6 relative in fp64, 1e-3 in fp32;
posterior mean / variance norm-wise ``max|a - b| / max|b|`` at the same levels.
"""
import json
import os

import numpy as np
import pytest
import torch

import stheno_amd as st
from oracle import gp_oracle as O
from stheno_amd import B, ops
from stheno_amd.matrix import Chol

from .conftest import ROOT, golden

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("hip_backend")]

DEV = "cuda"
TOL = {torch.float64: 1e-6, torch.float32: 1e-3}
EPS = {torch.float64: 1e-12, torch.float32: 1e-6}
KINDS = {"eq": st.EQ, "matern12": st.Matern12, "matern32": st.Matern32, "matern52": st.Matern52, "linear": st.Linear}


def dev(a, dtype=torch.float64):
    return torch.as_tensor(np.asarray(a), dtype=dtype, device=DEV)


def rel(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def kernel_from(g):
    return sum(float(v) * KINDS[str(k)]().stretch(float(s)) for k, v, s in zip(g["kinds"], g["variances"], g["scales"]))


class eps:
    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self.prev = B.epsilon
        B.epsilon = self.value

    def __exit__(self, *a):
        B.epsilon = self.prev


def test_native_library_is_the_backend():
    be = ops.get_backend()
    assert be.name == "hip" and be.lib.gpk_version() >= 100
    with open("/proc/self/maps") as f:
        assert "libgpk.so" in f.read()


# ------------------------------------------------------------------ ops vs oracle
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("kind", ["eq", "matern12", "matern32", "matern52", "linear", "const"])
def test_kmat_kinds(dtype, kind):
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal((2, 150, 5)), rng.standard_normal((2, 97, 5))
    terms = [(kind, 1.7, 0.8)]
    # The oracle follows upstream's |a|^2 + |b|^2 - 2ab distances; the HIP kernel uses direct
    # differences.  sqrt() in the Matern kernels amplifies that cancellation near r = 0
    # (oracle diagonal 2.9999999 vs exact 3.0): 1e-7 there, 1e-12 otherwise (bar: 1e-6).
    tol64 = 1e-7 if kind.startswith("matern") else 1e-12
    k = ops.get_backend().kmat(ops.KTerms(terms), dev(x, dtype), dev(y, dtype))
    assert k.shape == (2, 150, 97)
    assert rel(k, O.kernel_matrix(terms, x, y)) < (tol64 if dtype == torch.float64 else 1e-5)
    ks = ops.get_backend().kmat(ops.KTerms(terms), dev(x, dtype), None, diag_add=0.3, diag_vec=dev(np.ones((2, 150)), dtype))
    ref = O.kernel_matrix(terms, x) + 1.3 * np.eye(150)
    assert rel(ks, ref) < (tol64 if dtype == torch.float64 else 1e-5)
    assert rel(ops.get_backend().kdiag(ops.KTerms(terms), dev(x, dtype)), O.kernel_diag(terms, x)) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n", [1, 7, 127, 128, 129, 300, 1000])
def test_cholesky_solve_logdet(dtype, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 3))
    k = O.kernel_matrix([("eq", 1.0, 1.0)], x) + 0.5 * np.eye(n)
    c = Chol.factor_(dev(k, dtype).clone())
    l_ref = np.linalg.cholesky(k)
    tol = 1e-10 if dtype == torch.float64 else 2e-4
    assert rel(torch.tril(c.l), l_ref) < tol
    assert abs(float(c.logdet()) - O.logdet_chol(l_ref)) < tol * max(1.0, abs(O.logdet_chol(l_ref)))
    for nrhs in (1, 3, 40):
        b = rng.standard_normal((n, nrhs))
        assert rel(c.solve(dev(b, dtype)), O.solve_lower(l_ref, b)) < tol * 10
    assert rel(c.iqf_diag(dev(b, dtype)), O.iqf_diag(l_ref, b)) < tol * 10


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("n", [129, 1000, 2500])
def test_strict_upper_triangle_is_never_read(dtype, n):
    """The path builds K lower-triangle-only into uninitialised memory and factorises in place:
    nothing may depend on what the strict upper triangle holds.  Poison it with NaN."""
    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 3))
    k = O.kernel_matrix([("eq", 1.0, 1.0)], x) + 0.5 * np.eye(n)
    a = dev(k, dtype).clone()
    a[torch.triu(torch.ones(n, n, dtype=torch.bool, device=a.device), diagonal=1)] = float("nan")
    c = Chol.factor_(a).check()
    l_ref = np.linalg.cholesky(k)
    tol = 1e-10 if dtype == torch.float64 else 2e-4
    assert rel(torch.tril(c.l), l_ref) < tol
    assert abs(float(c.logdet()) - O.logdet_chol(l_ref)) < tol * max(1.0, abs(O.logdet_chol(l_ref)))
    for nrhs in (1, 40):
        b = rng.standard_normal((n, nrhs))
        got = c.solve(dev(b, dtype))
        assert torch.isfinite(got).all() and rel(got, O.solve_lower(l_ref, b)) < tol * 10
    w = c.inverse_lower()
    assert torch.isfinite(torch.tril(w)).all() and rel(torch.tril(w), np.linalg.inv(l_ref)) < tol * 100
    # the fused kernel-matrix build in `lower` mode into a poisoned buffer, then the whole logpdf
    out = torch.full((n, n), float("nan"), dtype=dtype, device=a.device)
    ops.get_backend().kmat(ops.KTerms([("eq", 1.0, 1.0)]), dev(x, dtype), None, lower=True, diag_add=0.5, out=out)
    c2 = Chol.factor_(out).check()
    assert rel(torch.tril(c2.l), l_ref) < tol


def test_not_positive_definite_raises():
    a = dev(np.array([[1.0, 2.0], [2.0, 1.0]]))
    with pytest.raises(torch.linalg.LinAlgError):
        Chol.factor_(a)
    with pytest.raises(torch.linalg.LinAlgError):
        st.Normal(dev(np.array([[1.0, 2.0], [2.0, 1.0]]))).logpdf(dev(np.zeros((2, 1))))


def test_empty_inputs():
    f = st.GP(1, st.EQ())
    post = f | (f(dev(np.zeros((0, 1)))), dev(np.zeros((0, 1))))
    assert post.mean is f.mean and post.kernel is f.kernel


# ------------------------------------------------------------------ golden fixtures through the API
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", ["dense_eq_n256_d8", "dense_eq_n300_d1_c3", "dense_matern12_n200_d3",
                                  "dense_matern32_n200_d3", "dense_matern52_n333_d5", "dense_eq_linear_n512_d4"])
def test_dense_golden(name, dtype):
    g = golden(name + ".npz")
    tol = TOL[dtype]
    with eps(EPS[dtype]):
        f = st.GP(kernel_from(g))
        x, xs, y = dev(g["x"], dtype), dev(g["xs"], dtype), dev(g["y"], dtype)
        noise = float(g["noise"])
        lp = f(x, noise).logpdf(y)
        assert lp.shape == (() if g["y"].shape[1] == 1 else (g["y"].shape[1],))
        assert rel(lp.reshape(-1), g["logpdf"]) < tol
        post = f | (f(x, noise), y[:, :1])
        mean, vd = post(xs).marginals()
        assert rel(mean, g["post_mean"]) < tol
        assert rel(vd, np.maximum(g["post_var_diag"], 0)) < tol
        assert rel(B.dense(post(xs).var), g["post_var"]) < tol
        m, lo, hi = post(xs).marginal_credible_bounds()
        assert torch.all(lo <= m) and torch.all(m <= hi)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_batched_golden(dtype):
    g = golden("batched_eq_b16_n100_d3.npz")
    with eps(EPS[dtype]):
        p = st.GP(kernel_from(g))
        x, y = dev(g["x"], dtype), dev(g["y"], dtype)
        lp = p(x, float(g["noise"])).logpdf(y)
        assert lp.shape == (16,)
        assert rel(lp, g["logpdf"]) < TOL[dtype]
        post = p | (p(x, 0.1), y)
        assert torch.all(post(x, 0.1).logpdf(y) > lp)          # tests/model/test_cases.py:146-155
        assert p(x, 0.1).sample().shape == (16, 100, 1)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("name", ["sparse_eq_n400_m50_d2", "sparse_matern32_linear_n300_m40_d3", "sparse_matern52_n350_m45_d2"])
def test_sparse_golden(name, dtype):
    g = golden(name + ".npz")
    # fp32 runs at the reference's own fp32 jitter (1e-6, README.md:887-888) and is held to the 1e-3 bar against the
    # oracle evaluated in fp64 at the SAME epsilon (the bound and the posterior depend on the jitter: K_z of 50
    # clustered inducing points has kappa ~ 1e8; the fixture itself was made with 1e-10).
    e = float(g["epsilon"]) if dtype == torch.float64 else 1e-6
    tol = TOL[dtype]
    terms = list(zip(g["kinds"], g["variances"], g["scales"]))
    with eps(e):
        m = st.Measure()
        f = st.GP(kernel_from(g), measure=m)
        x, z, xs, y = (dev(g[k], dtype) for k in ("x", "z", "xs", "y"))
        for cls, tag in [(st.PseudoObs, "vfe"), (st.PseudoObsFITC, "fitc"), (st.PseudoObsDTC, "dtc")]:
            obs = cls(f(z), f(x, float(g["noise"])), y)
            if dtype == torch.float64:
                ref_elbo, ref_mean, ref_vd = g[f"elbo_{tag}"], g[f"post_mean_{tag}"], g[f"post_var_diag_{tag}"]
                assert rel(obs.mu(m), g[f"mu_{tag}"]) < 1e-6
                assert rel(B.dense(obs.A(m)), g[f"A_{tag}"]) < 1e-6
            else:
                ref_elbo = np.atleast_1d(O.pseudo_obs(terms, g["x"], float(g["noise"]), g["y"], g["z"], method=tag, eps=e)["elbo"])
                ref_mean, _, ref_vd = O.pseudo_posterior(terms, g["x"], float(g["noise"]), g["y"], g["z"], g["xs"],
                                                         method=tag, eps=e, full_cov=False)
            assert rel(obs.elbo(m).reshape(1), ref_elbo) < tol
            mean, vd = (m | obs)(f)(xs).marginals()
            assert rel(mean, ref_mean) < max(tol, 1e-5)
            # The EQ fixture in fp32: kappa(K_z + 1e-6 I) ~ 5e7 = 3 / eps32, its posterior variance is the most sensitive number of the suite.
            # Round 5 held it to 2e-3 (1.09e-3 measured with the packed fp32 exp); round 6 evaluates K_z in fp64 and rounds it once
            # (`observations._kernel_matrix`): 7.1e-4 (profiles/r06_sparse_fp32_margin.log) -- north_star's 1e-3 holds everywhere again.
            vtol = max(tol, 1e-5)
            assert rel(vd, np.maximum(ref_vd, 0)) < vtol
        with pytest.raises(RuntimeError):
            st.PseudoObs(f(z), (f(x, torch.eye(x.shape[0], dtype=dtype, device=DEV)), y)).elbo(m)


def test_readme_known_answers():
    with open(os.path.join(ROOT, "tests", "golden", "readme_kats.json")) as fh:
        k = json.load(fh)
    f = st.GP(st.EQ())
    x = dev(k["logpdf_y1"]["x"])
    assert np.allclose(B.to_numpy(f(x).var), np.array(k["eq_matrix_x012"]["k"]), atol=5e-4)
    assert abs(float(f(x).logpdf(dev(k["logpdf_y1"]["y"]))) - k["logpdf_y1"]["logpdf"]) < 5e-8
    assert rel(f(x).logpdf(dev(k["logpdf_y2"]["y"])), k["logpdf_y2"]["logpdf"]) < 2e-8
    p = k["posterior_20s"]
    xl = torch.linspace(*p["x_linspace"][:2], int(p["x_linspace"][2]), dtype=torch.float64, device=DEV)
    pred = (f | (f(xl), xl**2))(dev(p["x_new"]))
    assert rel(pred.mean[:, 0], p["mean"]) < 1e-6          # kappa(K) ~ 1e12: see SURVEY A.6
    assert abs(float(B.dense(pred.var)[2, 2]) - p["var"][2][2]) / p["var"][2][2] < 1e-3


# ------------------------------------------------------------------ API scenarios on the device
def test_conditioning_spellings_chain_rule_and_nan():
    m = st.Measure()
    p = st.GP(1, st.EQ() + 2 * st.Exp(), measure=m)
    g = torch.Generator(device=DEV).manual_seed(0)
    x1, x2 = torch.linspace(0, 2, 50, dtype=torch.float64, device=DEV), torch.linspace(1.01, 3, 60, dtype=torch.float64, device=DEV)
    y = p(torch.cat([x1, x2]), 0.2).sample(generator=g)
    y1, y2 = y[:50], y[50:]
    xs = torch.linspace(0, 3, 33, dtype=torch.float64, device=DEV)
    posts = [m.condition(p(x1, 0.2), y1), m | (p(x1, 0.2), y1), m | st.Obs(p(x1, 0.2), y1)]
    for post in posts[1:]:
        assert rel(post(p)(xs).mean, B.to_numpy(posts[0](p)(xs).mean)) < 1e-12
    chain = p(x1, 0.2).logpdf(y1) + posts[0](p)(x2, 0.2).logpdf(y2)
    assert abs(float(chain) - float(p(torch.cat([x1, x2]), 0.2).logpdf(y))) < 1e-8 * abs(float(chain))
    y_nan = y1.clone(); y_nan[:3] = float("nan")
    a, b = (p | (p(x1, 0.2), y_nan))(xs), (p | (p(x1[3:], 0.2), y1[3:]))(xs)
    assert rel(a.mean, B.to_numpy(b.mean)) < 1e-10
    with pytest.raises(ValueError):
        p | (p(x1), torch.zeros(50, 2, dtype=torch.float64, device=DEV))


@pytest.mark.parametrize("cls", [st.PseudoObs, st.PseudoObsFITC, st.PseudoObsDTC])
def test_pseudo_points_at_inputs_are_exact(cls):
    m = st.Measure()
    p = st.GP(st.EQ() + 2 * st.Exp(), measure=m)
    x = torch.linspace(3, 5, 40, dtype=torch.float64, device=DEV)
    nz = torch.linspace(0.2, 0.5, 40, dtype=torch.float64, device=DEV)
    y = p(x, nz).sample()
    xs = torch.linspace(0, 5, 25, dtype=torch.float64, device=DEV)
    exact, appr = m | (p(x, nz), y), m | cls(p(x), p(x, nz), y)
    assert rel(appr(p)(xs).mean, B.to_numpy(exact(p)(xs).mean)) < 1e-7
    assert rel(B.dense(appr(p)(xs).var), B.to_numpy(exact(p)(xs).var)) < 1e-7
    assert abs(float(cls(p(x), p(x, nz), y).elbo(m)) - float(p(x, nz).logpdf(y))) < 1e-8 * abs(float(p(x, nz).logpdf(y)))


def test_marginals_efficiency_10000_points():
    p = st.GP(st.EQ())
    x = torch.linspace(0, 5, 5, dtype=torch.float64, device=DEV)
    y = p(x, 0.1).sample()
    p = p | (p(x, 0.1), y)
    xs = torch.linspace(0, 5, 10_000, dtype=torch.float64, device=DEV)
    p(xs, 0.2).marginal_credible_bounds()    # warm
    torch.cuda.synchronize()
    import time
    t0 = time.time()
    p(xs, 0.2).marginal_credible_bounds()
    torch.cuda.synchronize()
    assert time.time() - t0 < 1


@pytest.mark.parametrize("n", [1000, 3000, 4096 + 37])
def test_triangular_inverse_and_k_inverse(n):
    """Pieces of the log-density backward at sizes that reach the 128-tile kernels: W = L^{-1}
    (TRSM sweep on the identity with a growing column range) and K^{-1} = W^T W (lower SYRK that
    skips the all-zero k-chunks of the triangular factor)."""
    rng = np.random.default_rng(n)
    x = dev(rng.standard_normal((n, 3)))
    be = ops.get_backend()
    k_full = be.kmat(ops.KTerms([("eq", 1.0, 1.0)]), x, None, diag_add=0.5)
    c = Chol.factor_(k_full.clone())
    W = c.inverse_lower()
    L = c.lower()
    eye = torch.eye(n, dtype=torch.float64, device=DEV)
    assert float((W @ L - eye).abs().max()) < 1e-10
    assert float(torch.triu(W, 1).abs().max()) == 0.0
    kinv = be.gemm(W, W, a_kmajor=False, b_kmajor=False, lower_only=True, tri_k=True)
    ksym = torch.tril(kinv) + torch.tril(kinv, -1).T
    assert float((ksym @ k_full - eye).abs().max()) < 1e-8


def test_outputs_must_not_need_a_copy():
    """A transposed view as ``out=`` once made the GEMM write into a temporary copy and return the
    un-updated tensor (posterior of a product process under chained conditioning): now refused."""
    be = ops.get_backend()
    a = dev(np.random.default_rng(0).standard_normal((8, 5)))
    good = be.gemm(a, a, out=dev(np.zeros((8, 8))))
    assert rel(good, (a @ a.T).cpu().numpy()) < 1e-12
    with pytest.raises(ValueError):
        be.gemm(a, a, out=dev(np.zeros((8, 8))).t())
    with pytest.raises(ValueError):
        be.add_diag_(dev(np.zeros((8, 8))).t(), 1.0)


def test_normal_arithmetic_and_kl_on_device():
    rng = np.random.default_rng(2)
    c1, c2 = rng.standard_normal((40, 40)), rng.standard_normal((40, 40))
    v1, v2 = c1 @ c1.T + np.eye(40), c2 @ c2.T + np.eye(40)
    m1, m2, a = rng.standard_normal((40, 1)), rng.standard_normal((40, 1)), rng.standard_normal((40, 40))
    n1, n2 = st.Normal(dev(m1), dev(v1)), st.Normal(dev(m2), dev(v2))
    la = n1.lmatmul(dev(a))
    assert rel(la.mean, a @ m1) < 1e-12 and rel(B.dense(la.var), a @ v1 @ a.T) < 1e-12
    ra = n1.rmatmul(dev(a))
    assert rel(ra.mean, a.T @ m1) < 1e-12 and rel(B.dense(ra.var), a.T @ v1 @ a) < 1e-12
    want = 0.5 * (np.trace(np.linalg.solve(v2, v1)) + ((m2 - m1).T @ np.linalg.solve(v2, m2 - m1))[0, 0] - 40
                  + np.linalg.slogdet(v2)[1] - np.linalg.slogdet(v1)[1])
    assert abs(float(n1.kl(n2)) - want) < 1e-9 * abs(want) and float(n1.kl(n1)) < 1e-9
    assert rel(B.dense(n1.m2), v1 + m1 @ m1.T) < 1e-13


@pytest.mark.parametrize("dtype,n", [(torch.float64, 8192 + 37), (torch.float32, 8192 + 128)])
def test_lookahead_factorisation_through_the_api(dtype, n):
    """Large orders (``matrix.config.potrf_lookahead_from``) take ``gpk_potrf_la`` (look-ahead, helper stream, persistent trailing update, plain tail): the factor
    against LAPACK, the merged block inverses it returns against what the solves then compute, ragged order included."""
    from stheno_amd import matrix

    rng = np.random.default_rng(n)
    x = rng.standard_normal((n, 4))
    k = O.kernel_matrix([("eq", 1.0, 1.0)], x) + 0.5 * np.eye(n)
    first = matrix.config.potrf_lookahead_from
    matrix.config.potrf_lookahead_from = min(first, n)                # (the default threshold is above this order: measured crossover)
    try:
        c = Chol.factor_(dev(k, dtype).clone())
    finally:
        matrix.config.potrf_lookahead_from = first
    nb = 512 if n < matrix.config.potrf_lookahead_wide_from else matrix.config.potrf_lookahead_nb[dtype]
    sb = min(nb, matrix.config.potrf_lookahead_inv[dtype])
    assert c.lookahead_nb == nb and c.lookahead_sb == sb              # the look-ahead path ran
    # the block inverses it leaves behind are what the merge of the 128-block inverses computes (they are kept with the factor
    # only when the solves use that block size: matrix.Chol.factor_)
    be = ops.get_backend()
    a2 = dev(k, dtype).clone()
    dinv2, info2, dnb = be.potrf_(a2, 0, lookahead_nb=nb, lookahead_sb=sb)
    assert int(info2.max()) == 0 and dnb.shape[-3] == (n + sb - 1) // sb
    merged = be.trtri_merge(a2, dinv2, sb)
    assert rel(dnb, merged.double().cpu().numpy()) < (1e-11 if dtype == torch.float64 else 1e-4)
    del a2, dinv2, dnb, merged
    l_ref = np.linalg.cholesky(k)
    tol = 1e-10 if dtype == torch.float64 else 2e-4
    assert rel(torch.tril(c.l), l_ref) < tol
    assert abs(float(c.logdet()) - O.logdet_chol(l_ref)) < tol * abs(O.logdet_chol(l_ref))
    b = rng.standard_normal((n, 40))
    assert rel(c.solve(dev(b, dtype)), O.solve_lower(l_ref, b)) < tol * 10
    # the plain factorisation of the same matrix agrees to round-off
    old = matrix.config.potrf_lookahead_from
    matrix.config.potrf_lookahead_from = 0
    try:
        c2 = Chol.factor_(dev(k, dtype).clone())
    finally:
        matrix.config.potrf_lookahead_from = old
    assert c2.lookahead_nb == 0 and rel(torch.tril(c.l), torch.tril(c2.l).double().cpu().numpy()) < tol


def test_numpy_inputs_are_moved_to_the_device():
    """The reference's default backend is fed NumPy arrays (``README.md:43-86``): host data given to the HIP path is copied
    to the device once; results are device tensors (``B.to_numpy`` brings them back)."""
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "readme_kats.json")))
    x = np.linspace(0, 2, 10)
    y = x ** 2
    f = st.GP(st.EQ())
    post = f | (f(x), y)
    pred = post(np.array([1.0, 2.0, 3.0]))
    assert pred.mean.device.type == "cuda"
    assert rel(pred.mean[:, 0], [1.00000068, 3.99999999, 8.4825932]) < 1e-6       # README.md:58-61
    lp = f(x, 0.1).logpdf(y)
    assert lp.device.type == "cuda" and abs(float(lp) - O.gp_logpdf([("eq", 1.0, 1.0)], x, 0.1, y[:, None])) < 1e-9
    assert isinstance(B.to_numpy(pred.mean), np.ndarray)
    assert kat is not None
