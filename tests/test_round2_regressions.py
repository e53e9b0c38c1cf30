"""Round-2 additions: parity of sampling with the oracle (row f2 of SURVEY section 8), BASELINE config 1 at
its stated size, the advisor's findings (pseudo-points with fewer than three inducing points, gradients that
must not be cut silently, the sharded bound under autograd), and the epsilon-regularised factor of the
pseudo-point posterior.  Every test runs on the CPU box over the oracle backend and on the MI355X over libgpk.so."""
import numpy as np
import pytest
import torch

import stheno_amd as st
from oracle import gp_oracle as O
from stheno_amd import B, dist

from .conftest import DEVICE, T

pytestmark = pytest.mark.usefixtures("any_backend")
TOL = {torch.float64: 1e-6, torch.float32: 1e-3}


def rel(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


class eps:
    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self.prev = B.epsilon
        B.epsilon = self.value

    def __exit__(self, *a):
        B.epsilon = self.prev


# ------------------------------------------------------------------ sampling vs the oracle (stheno/random.py:331-363)
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_normal_sample_matches_oracle_given_the_same_draws(dtype):
    rng = np.random.default_rng(0)
    n, num = 300, 5
    x = rng.standard_normal((n, 3))
    terms = [("eq", 1.3, 0.9), ("matern32", 0.5, 2.0)]
    k = 1.3 * st.EQ().stretch(0.9) + 0.5 * st.Matern32().stretch(2.0)
    xi = rng.standard_normal((n, num))
    e = 1e-12 if dtype == torch.float64 else 1e-6
    with eps(e):
        f = st.GP(k)
        got = f(T(x, dtype), 0.2).sample(xi=T(xi, dtype))
        ref = O.sample(O.kernel_matrix(terms, x) + 0.2 * np.eye(n), xi, eps=e)
        assert got.shape == (n, num) and rel(got, ref) < TOL[dtype]
        # mean and the extra `noise` argument of Normal.sample (random.py:343-349)
        mean = rng.standard_normal((n, 1))
        d = st.Normal(T(mean, dtype), T(O.kernel_matrix(terms, x) + 0.2 * np.eye(n), dtype))
        ref2 = O.sample(O.kernel_matrix(terms, x) + 0.5 * np.eye(n), xi, eps=e, mean=mean)
        assert rel(d.sample(noise=0.3, xi=T(xi, dtype)), ref2) < TOL[dtype]
    with pytest.raises(ValueError):
        f(T(x, dtype), 0.2).sample(xi=T(xi[:7], dtype))


def test_batched_sample_matches_oracle():
    rng = np.random.default_rng(1)
    b, n = 4, 120
    x = rng.standard_normal((b, n, 2))
    xi = rng.standard_normal((b, n, 2))
    got = st.GP(st.EQ())(T(x), 0.1).sample(xi=T(xi))
    assert got.shape == (b, n, 2)
    for i in range(b):
        ref = O.sample(O.kernel_matrix([("eq", 1.0, 1.0)], x[i]) + 0.1 * np.eye(n), xi[i])
        assert rel(got[i], ref) < 1e-6


def test_joint_sample_of_two_processes_matches_oracle():
    """``Measure.sample(f1(x1), f2(x2))`` draws from the JOINT (measure.py:425-461): with f2 = f1 + g the block kernel
    matrix is [[k1, k1], [k1, k1 + k2]]."""
    rng = np.random.default_rng(2)
    n1, n2 = 70, 50
    x1, x2 = rng.standard_normal((n1, 1)), rng.standard_normal((n2, 1))
    xi = rng.standard_normal((n1 + n2, 3))
    m = st.Measure()
    f1 = st.GP(st.EQ(), measure=m)
    g = st.GP(0.5 * st.Matern52(), measure=m)
    f2 = f1 + g
    s1, s2 = m.sample(f1(T(x1), 0.05), f2(T(x2), 0.1), xi=T(xi))
    k1, k2 = [("eq", 1.0, 1.0)], [("matern52", 0.5, 1.0)]
    joint = np.block([
        [O.kernel_matrix(k1, x1) + 0.05 * np.eye(n1), O.kernel_matrix(k1, x1, x2)],
        [O.kernel_matrix(k1, x2, x1), O.kernel_matrix(k1, x2) + O.kernel_matrix(k2, x2) + 0.1 * np.eye(n2)],
    ])
    ref = O.sample(joint, xi)
    assert s1.shape == (n1, 3) and s2.shape == (n2, 3)
    assert rel(torch.cat([s1, s2]), ref) < 1e-6


# ------------------------------------------------------------------ BASELINE.json configs[0] at its stated size
def test_config1_eq_n512_d1_fp64():
    """EQ kernel, N = 512, D = 1, fp64: logpdf + posterior mean / variance against the oracle."""
    rng = np.random.default_rng(3)
    n, ns, noise = 512, 100, 0.1
    x = np.sort(rng.uniform(0, 10, n))
    xs = np.linspace(0, 10, ns)
    terms = [("eq", 1.0, 1.0)]
    y = O.sample(O.kernel_matrix(terms, x[:, None]) + noise * np.eye(n), rng.standard_normal((n, 1)))
    f = st.GP(st.EQ())
    fd = f(T(x), noise)
    ref_lp = O.gp_logpdf(terms, x[:, None], noise, y)
    assert abs(float(fd.logpdf(T(y))) - ref_lp) <= 1e-6 * abs(ref_lp)
    mean, var = (f | (fd, T(y)))(T(xs)).marginals()
    ref_m, _, ref_v = O.gp_posterior(terms, x[:, None], noise, y, xs[:, None], full_cov=False)
    assert rel(mean, ref_m) < 1e-6 and rel(var, np.maximum(ref_v, 0)) < 1e-6
    m2, lo, hi = (f | (fd, T(y)))(T(xs)).marginal_credible_bounds()
    assert torch.all(lo <= m2) and torch.all(m2 <= hi)


# ------------------------------------------------------------------ advisor: pseudo-points with M < 3
@pytest.mark.parametrize("cls,tag", [(st.PseudoObs, "vfe"), (st.PseudoObsFITC, "fitc"), (st.PseudoObsDTC, "dtc")])
@pytest.mark.parametrize("m_pts", [1, 2, 3])
def test_pseudo_points_with_one_or_two_inducing_points(cls, tag, m_pts):
    rng = np.random.default_rng(10 + m_pts)
    n = 40
    x, z, xs = rng.standard_normal((n, 2)), rng.standard_normal((m_pts, 2)), rng.standard_normal((9, 2))
    y = rng.standard_normal((n, 1))
    terms = [("eq", 1.0, 1.0)]
    meas = st.Measure()
    f = st.GP(st.EQ(), measure=meas)
    obs = cls(f(T(z)), f(T(x), 0.3), T(y))
    ref = O.pseudo_obs(terms, x, 0.3, y, z, method=tag)
    assert abs(float(obs.elbo(meas)) - ref["elbo"]) <= 1e-8 * abs(ref["elbo"])
    assert rel(obs.mu(meas), ref["mu"]) < 1e-7
    mean, vd = (meas | obs)(f)(T(xs)).marginals()
    rm, _, rv = O.pseudo_posterior(terms, x, 0.3, y, z, xs, method=tag, full_cov=False)
    assert rel(mean, rm) < 1e-6 and rel(vd, np.maximum(rv, 0)) < 1e-6


# ------------------------------------------------------------------ the epsilon-regularised subspace factor
@pytest.mark.parametrize("e", [1e-10, 1e-6, 1e-4])
def test_pseudo_point_posterior_matches_the_reference_at_every_epsilon(e):
    """The reference takes chol(L_z A L_z^T + eps I) (mlkernels.SubspaceKernel -> B.iqf -> B.cholesky(B.reg(.))); the
    product-form factor used here must give the same posterior, not the one of chol(L_z A L_z^T)."""
    g = np.load(__file__.replace("test_round2_regressions.py", "golden/sparse_eq_n400_m50_d2.npz"))
    terms = list(zip(g["kinds"], g["variances"], g["scales"]))
    with eps(e):
        m = st.Measure()
        f = st.GP(st.EQ(), measure=m)
        obs = st.PseudoObs(f(T(g["z"])), f(T(g["x"]), float(g["noise"])), T(g["y"]))
        mean, vd = (m | obs)(f)(T(g["xs"])).marginals()
        rm, _, rv = O.pseudo_posterior(terms, g["x"], float(g["noise"]), g["y"], g["z"], g["xs"], method="vfe", eps=e, full_cov=False)
        assert rel(mean, rm) < 1e-6
        assert rel(vd, np.maximum(rv, 0)) < 2e-6     # kappa(K_z) ~ 1e8 amplifies round-off of the two factorisations


# ------------------------------------------------------------------ advisor: gradients are never cut silently
def test_logpdf_with_missing_data_stays_differentiable():
    rng = np.random.default_rng(5)
    x, y = rng.standard_normal((30, 2)), rng.standard_normal((30, 1))
    y[[3, 17]] = np.nan
    ls = torch.tensor(1.3, dtype=torch.float64, device=DEVICE[0], requires_grad=True)
    f = st.GP(st.EQ().stretch(ls))
    lp = f(T(x), 0.2).logpdf(T(y))
    assert lp.requires_grad
    lp.backward()
    keep = ~np.isnan(y[:, 0])
    h = 1e-6
    fd = (O.gp_logpdf([("eq", 1.0, 1.3 + h)], x[keep], 0.2, y[keep]) - O.gp_logpdf([("eq", 1.0, 1.3 - h)], x[keep], 0.2, y[keep])) / (2 * h)
    assert abs(float(ls.grad) - fd) <= 1e-5 * max(1.0, abs(fd))
    assert abs(float(lp) - O.gp_logpdf([("eq", 1.0, 1.3)], x[keep], 0.2, y[keep])) <= 1e-9 * abs(float(lp))


def test_gradients_outside_the_differentiable_path_are_refused():
    x = torch.randn(10, 2, dtype=torch.float64, device=DEVICE[0], requires_grad=True)
    y = torch.randn(10, 1, dtype=torch.float64, device=DEVICE[0])
    f = st.GP(st.EQ())
    assert f(x, 0.1).logpdf(y).requires_grad                 # d/dx: implemented since (tests/test_autograd_inputs.py)
    post = f | (f(x[:5].detach(), 0.1), y[:5])
    with pytest.raises(NotImplementedError):
        post(x[5:], 0.1).logpdf(y[5:])                       # d/dx of a POSTERIOR log-density is not: loud, not silent
    with torch.no_grad():
        assert torch.isfinite(post(x[5:], 0.1).logpdf(y[5:]))
    c = torch.tensor(2.0, dtype=torch.float64, device=DEVICE[0], requires_grad=True)
    with pytest.raises(NotImplementedError):
        c * f                                                # a learnable scale belongs on the kernel
    with pytest.raises(NotImplementedError):
        f + c
    assert (2.0 * f).kernel is not None and (f + 1.0).mean is not None


def test_mu_and_A_before_K_z_with_gradients_enabled():
    rng = np.random.default_rng(6)
    x, z, y = rng.standard_normal((50, 1)), rng.standard_normal((7, 1)), rng.standard_normal((50, 1))
    ls = torch.tensor(0.8, dtype=torch.float64, device=DEVICE[0], requires_grad=True)
    m = st.Measure()
    f = st.GP(st.EQ().stretch(ls), measure=m)
    obs = st.PseudoObs(f(T(z)), f(T(x), 0.2), T(y))
    assert obs.elbo(m).requires_grad                         # differentiable path: caches nothing ...
    ref = O.pseudo_obs([("eq", 1.0, 0.8)], x, 0.2, y, z)
    assert rel(obs.mu(m), ref["mu"]) < 1e-7                  # ... and mu() / A() still work, in any order
    assert rel(B.dense(obs.A(m)), ref["A"]) < 1e-7


def test_sharded_bound_refuses_gradients_and_returns_the_reduced_value():
    rng = np.random.default_rng(7)
    x, z, y = rng.standard_normal((50, 1)), rng.standard_normal((7, 1)), rng.standard_normal((50, 1))
    ls = torch.tensor(0.8, dtype=torch.float64, device=DEVICE[0], requires_grad=True)
    m = st.Measure()
    f = st.GP(st.EQ().stretch(ls), measure=m)
    obs = st.PseudoObs(f(T(z)), f(T(x), 0.2), T(y))
    with pytest.raises(NotImplementedError):
        dist.sharded_elbo(obs, m)
    with torch.no_grad():
        val = dist.sharded_elbo(obs, m)
    ref = O.pseudo_obs([("eq", 1.0, 0.8)], x, 0.2, y, z)["elbo"]
    assert not val.requires_grad and abs(float(val) - ref) <= 1e-8 * abs(ref)


# ------------------------------------------------------------------ gradients through the BATCHED logpdf (README.md:744-766)
@pytest.mark.parametrize("b,n,d,kinds", [(3, 40, 2, ("eq",)), (8, 300, 3, ("eq", "matern32"))])
def test_batched_logpdf_gradients(b, n, d, kinds):
    """Hyper-parameters shared by B independent data sets (the configuration that shards over GPUs): gradients w.r.t.
    variances, length scales, the noise and y against central finite differences of the oracle, entry by entry."""
    from .test_autograd import KINDS, fd_grad, logpdf_direct

    rng = np.random.default_rng(b * n)
    x, y = rng.standard_normal((b, n, d)), rng.standard_normal((b, n, 1))
    var0, sc0, noise0 = rng.uniform(0.5, 1.5, len(kinds)), rng.uniform(0.7, 1.6, len(kinds)), 0.3
    wts = rng.uniform(0.5, 1.5, b)

    def oracle(params):
        v, s, nz = params[: len(kinds)], params[len(kinds): 2 * len(kinds)], params[-1]
        terms = [(k, v[i], s[i]) for i, k in enumerate(kinds)]
        return float(sum(wts[i] * logpdf_direct(terms, x[i], nz, y[i]) for i in range(b)))

    p0 = np.concatenate([var0, sc0, [noise0]])
    ref = fd_grad(oracle, p0)
    dev = DEVICE[0]
    vs = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in var0]
    ss = [torch.tensor(s, dtype=torch.float64, requires_grad=True) for s in sc0]
    nz = torch.tensor(noise0, dtype=torch.float64, requires_grad=True)
    ty = torch.tensor(y, dtype=torch.float64, device=dev, requires_grad=True)
    f = st.GP(sum(v * KINDS[k]().stretch(s) for v, k, s in zip(vs, kinds, ss)))
    lp = f(T(x), nz.to(dev)).logpdf(ty)
    assert lp.shape == (b,) and lp.requires_grad
    assert abs(float((lp.detach().cpu() * torch.tensor(wts)).sum()) - oracle(p0)) <= 1e-7 * abs(oracle(p0))
    (lp * torch.tensor(wts, device=dev)).sum().backward()
    got = np.array([float(v.grad) for v in vs] + [float(s.grad) for s in ss] + [float(nz.grad)])
    assert np.max(np.abs(got - ref)) <= 5e-6 * max(np.max(np.abs(ref)), 1.0), (got, ref)
    # d/dy of entry 0 in closed form: -w_0 K_0^{-1} y_0 (entries are independent)
    t0 = [(k, var0[i], sc0[i]) for i, k in enumerate(kinds)]
    alpha = np.linalg.solve(sum(v * O._kappa(k, ((x[0][:, None, :] - x[0][None, :, :]) ** 2).sum(-1) / s**2, x[0] @ x[0].T / s**2)
                                for k, v, s in t0) + noise0 * np.eye(n), y[0])
    assert rel(ty.grad[0], -wts[0] * alpha) < 1e-6


# ------------------------------------------------------------------ per-dimension length scales (k.stretch(vector))
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_per_dimension_length_scales(dtype):
    """``EQ().stretch([l_1, ..., l_D])`` = the EQ kernel on ``x / l`` (mlkernels' Stretched with a vector): log-density and
    posterior against the oracle evaluated on explicitly rescaled inputs; a sum with a differently scaled term."""
    rng = np.random.default_rng(21)
    n, ns, d, noise = 200, 30, 3, 0.2
    x, xs, y = rng.standard_normal((n, d)), rng.standard_normal((ns, d)), rng.standard_normal((n, 1))
    ell = np.array([0.5, 1.5, 3.0])
    e = 1e-12 if dtype == torch.float64 else 1e-6
    with eps(e):
        f = st.GP(1.3 * st.EQ().stretch(T(ell, dtype)))
        ref_lp = O.gp_logpdf([("eq", 1.3, 1.0)], x / ell, noise, y, eps=e)
        fd = f(T(x, dtype), noise)
        assert abs(float(fd.logpdf(T(y, dtype))) - ref_lp) <= TOL[dtype] * abs(ref_lp)
        mean, var = (f | (fd, T(y, dtype)))(T(xs, dtype)).marginals()
        rm, _, rv = O.gp_posterior([("eq", 1.3, 1.0)], x / ell, noise, y, xs / ell, full_cov=False, eps=e)
        assert rel(mean, rm) < TOL[dtype] and rel(var, np.maximum(rv, 0)) < TOL[dtype]
        # kernel values, diagonal, scalar re-stretch on top, sum with a scalar-stretched Matern term
        k2 = st.EQ().stretch(T(ell, dtype)).stretch(2.0) + 0.5 * st.Matern32().stretch(0.7)
        want = O.kernel_matrix([("eq", 1.0, 1.0)], x / (2 * ell), xs / (2 * ell)) + O.kernel_matrix([("matern32", 0.5, 0.7)], x, xs)
        assert rel(k2.pairwise(T(x, dtype), T(xs, dtype)), want) < (1e-6 if dtype == torch.float64 else 1e-5)
        assert rel(k2.elwise(T(x, dtype))[:, 0], 1.5 * np.ones(n)) < 1e-6
        g = st.GP(k2)
        want_k = O.kernel_matrix([("eq", 1.0, 1.0)], x / (2 * ell)) + O.kernel_matrix([("matern32", 0.5, 0.7)], x) + noise * np.eye(n)
        ref2 = O.normal_logpdf(None, want_k, y, eps=e)
        assert abs(float(g(T(x, dtype), noise).logpdf(T(y, dtype))) - ref2) <= TOL[dtype] * abs(ref2)
    with pytest.raises(ValueError):
        st.EQ().stretch(T([1.0, 2.0], dtype)).pairwise(T(x, dtype))           # 2 scales, 3 input dimensions
    ls = torch.tensor([1.0, 2.0, 3.0], dtype=dtype, device=DEVICE[0], requires_grad=True)
    lp = st.GP(st.EQ().stretch(ls))(T(x, dtype), noise).logpdf(T(y, dtype))     # learnable: tests/test_autograd_inputs.py
    assert lp.requires_grad
    lp.backward()
    assert ls.grad.shape == (3,) and bool(torch.isfinite(ls.grad).all())


def test_nan_scan_can_be_switched_off():
    """``config.check_nan = False``: no device->host read of the NaN flag; values for complete data are unchanged."""
    from stheno_amd.matrix import config

    rng = np.random.default_rng(31)
    x, y = rng.standard_normal((40, 2)), rng.standard_normal((40, 1))
    f = st.GP(st.EQ())
    want = float(f(T(x), 0.2).logpdf(T(y)))
    post_want = (f | (f(T(x), 0.2), T(y)))(T(x[:5])).mean
    config.check_nan = False
    try:
        assert abs(float(f(T(x), 0.2).logpdf(T(y))) - want) <= 1e-12 * abs(want)
        assert rel((f | (f(T(x), 0.2), T(y)))(T(x[:5])).mean, post_want.cpu().numpy()) < 1e-12
    finally:
        config.check_nan = True


def test_whitened_observations_are_solved_for_once(any_backend, monkeypatch):
    """Round 3: the log-density (``random.py:276``) and the posterior mean (``observations.py:161-168``) of the same observations both
    need ``L^{-1} y``; with a zero prior mean the factor hands the second asker the first one's result -- and notices when ``y``
    has changed in place in between."""
    from stheno_amd import matrix

    dev = torch.device(DEVICE[0])
    g = torch.Generator().manual_seed(5)
    x = torch.randn(300, 2, generator=g, dtype=torch.float64).to(dev)
    y = torch.randn(300, 1, generator=g, dtype=torch.float64).to(dev)
    xs = torch.randn(7, 2, generator=g, dtype=torch.float64).to(dev)
    calls = []
    orig = matrix.Chol.solve
    monkeypatch.setattr(matrix.Chol, "solve", lambda self, b: (calls.append(tuple(b.shape)), orig(self, b))[1])

    f = st.GP(st.EQ())
    fdd = f(x, 0.1)
    lp = fdd.logpdf(y)
    post = f | (fdd, y)
    mean, var = post(xs).marginals()
    # one single-column solve for both on the device; host memory can change behind torch's back (NumPy aliases): solved twice there
    assert calls.count((300, 1)) == (1 if y.is_cuda else 2)
    # the same numbers as without the shortcut
    f2 = st.GP(st.EQ())
    mean2, var2 = (f2 | (f2(x, 0.1), y))(xs).marginals()
    assert torch.equal(mean, mean2) and torch.equal(var, var2)
    ref = O.gp_logpdf([("eq", 1.0, 1.0)], x.cpu().numpy(), 0.1, y.cpu().numpy())
    assert abs(float(lp) - ref) < 1e-9 * abs(ref)
    # y modified in place after the log-density: the posterior must see the new values
    f3 = st.GP(st.EQ())
    fdd3 = f3(x, 0.1)
    fdd3.logpdf(y)
    y.mul_(2.0)
    mean3, _ = (f3 | (fdd3, y))(xs).marginals()
    assert torch.allclose(mean3, 2.0 * mean2, rtol=1e-10, atol=0)
