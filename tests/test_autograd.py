"""Gradients of ``f(x, noise).logpdf(y)`` w.r.t. kernel hyper-parameters, the noise and
``y`` (reference usage ``readme_example13_optimisation_torch.py:20-52``),
checked against central finite differences of the CPU oracle's log-density.

The CPU variant runs the autograd.Function's host logic on the test-only oracle backend; the
GPU variant runs it through libgpk.so (TRSM on the identity, lower SYRK, gpk_kmat_vjp).
"""
import numpy as np
import pytest
import torch

import stheno_amd as st
from oracle import gp_oracle as O

KINDS = {"eq": st.EQ, "matern12": st.Matern12, "matern32": st.Matern32, "matern52": st.Matern52, "linear": st.Linear}


def logpdf_direct(terms, x, noise, y):
    """The oracle's log-density with DIRECT-difference distances.  Upstream's (and the
    oracle's) ``|a|^2 + |b|^2 - 2ab`` leaves ~1e-15 rounding noise in the zero distances on the
    diagonal; under the square root of the Matern kernels that noise is 3e-8 and makes finite
    differences in the length scale meaningless (error ~1e-2).  Values agree to 1e-7."""
    n = x.shape[0]
    d2 = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    dot = x @ x.T
    k = sum(v * O._kappa(kind, d2 / s**2, dot / s**2) for kind, v, s in terms)
    return O.normal_logpdf(None, k + noise * np.eye(n), y)


def fd_grad(fun, p, h=1e-6):
    g = np.zeros_like(p)
    for i in range(p.size):
        e = np.zeros_like(p); e.flat[i] = h
        g.flat[i] = (fun(p + e) - fun(p - e)) / (2 * h)
    return g


def run_case(dev, dtype, kinds, n, d, c, seed, tol):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d))
    y = rng.standard_normal((n, c))
    var0 = rng.uniform(0.5, 1.5, len(kinds))
    sc0 = rng.uniform(0.7, 1.6, len(kinds))
    noise0 = 0.3
    wts = rng.uniform(0.5, 1.5, c)           # upstream weights of the C log-densities

    def oracle(params):
        v, s, nz = params[: len(kinds)], params[len(kinds): 2 * len(kinds)], params[-1]
        terms = [(k, v[i], s[i]) for i, k in enumerate(kinds)]
        return float(np.sum(wts * np.atleast_1d(logpdf_direct(terms, x, nz, y))))

    p0 = np.concatenate([var0, sc0, [noise0]])
    t0 = [(k, var0[i], sc0[i]) for i, k in enumerate(kinds)]
    assert abs(oracle(p0) - float(np.sum(wts * np.atleast_1d(O.gp_logpdf(t0, x, noise0, y))))) <= 1e-7 * abs(oracle(p0))
    ref = fd_grad(oracle, p0)
    ref_y = fd_grad(lambda yy: float(np.sum(wts * np.atleast_1d(logpdf_direct(t0, x, noise0, yy.reshape(n, c))))),
                    y.copy().ravel(), h=1e-5)

    vs = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in var0]
    ss = [torch.tensor(s, dtype=torch.float64, requires_grad=True) for s in sc0]
    nz = torch.tensor(noise0, dtype=torch.float64, requires_grad=True)
    ty = torch.tensor(y, dtype=dtype, device=dev, requires_grad=True)
    kernel = sum(v * KINDS[k]().stretch(s) for v, k, s in zip(vs, kinds, ss))
    f = st.GP(kernel)
    lp = f(torch.tensor(x, dtype=dtype, device=dev), nz.to(dtype=dtype, device=dev)).logpdf(ty)
    assert lp.shape == (() if c == 1 else (c,))
    assert abs(float((lp.detach().reshape(-1).double().cpu() * torch.tensor(wts)).sum()) - oracle(p0)) <= tol * abs(oracle(p0))
    (lp.reshape(-1) * torch.tensor(wts, dtype=dtype, device=dev)).sum().backward()
    got = np.array([float(v.grad) for v in vs] + [float(s.grad) for s in ss] + [float(nz.grad)])
    assert np.max(np.abs(got - ref)) <= tol * max(np.max(np.abs(ref)), 1.0), (got, ref)
    gy = ty.grad.double().cpu().numpy().ravel()
    assert np.max(np.abs(gy - ref_y)) <= tol * max(np.max(np.abs(ref_y)), 1.0)


CASES = [(("eq",), 40, 3, 1), (("eq", "linear"), 37, 2, 1), (("matern32",), 50, 1, 3), (("matern52", "matern12"), 45, 4, 2)]


@pytest.mark.parametrize("kinds,n,d,c", CASES)
def test_logpdf_gradients_host_logic(oracle_backend, kinds, n, d, c):
    run_case("cpu", torch.float64, kinds, n, d, c, seed=len(kinds) + n, tol=2e-6)


def test_no_grad_path_is_untouched(oracle_backend):
    x = torch.linspace(0, 3, 20, dtype=torch.float64)
    y = torch.randn(20, 1, dtype=torch.float64)
    v = torch.tensor(1.3, dtype=torch.float64, requires_grad=True)
    f = st.GP(v * st.EQ())
    with torch.no_grad():
        lp = f(x, 0.1).logpdf(y)
    assert not lp.requires_grad
    lp2 = f(x, 0.1).logpdf(y)
    assert lp2.requires_grad and abs(float(lp2) - float(lp)) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("kinds,n,d,c", CASES + [(("eq",), 700, 8, 1), (("eq", "linear"), 300, 4, 2)])
def test_logpdf_gradients_gpu(hip_backend, kinds, n, d, c):
    run_case("cuda", torch.float64, kinds, n, d, c, seed=len(kinds) + n, tol=5e-6)


@pytest.mark.gpu
def test_logpdf_gradients_gpu_fp32(hip_backend):
    st.B.epsilon = 1e-6
    try:
        run_case("cuda", torch.float32, ("eq",), 200, 3, 1, seed=7, tol=5e-3)
    finally:
        st.B.epsilon = 1e-12


@pytest.mark.gpu
def test_gradient_descent_step_improves_the_fit(hip_backend):
    """One Adam-free gradient step on (variance, scale, noise) raises the log-density."""
    g = torch.Generator().manual_seed(0)
    x = torch.linspace(0, 10, 400, dtype=torch.float64)
    ytrue = torch.sin(x)[:, None] + 0.1 * torch.randn(400, 1, generator=g, dtype=torch.float64)
    params = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in (0.5, 0.3, 0.5)]

    def logpdf():
        f = st.GP(params[0] * st.EQ().stretch(params[1]))
        return f(x.cuda(), params[2].cuda()).logpdf(ytrue.cuda())

    lp0 = logpdf()
    lp0.backward()
    with torch.no_grad():
        for p in params:
            p += 1e-4 * p.grad / p.grad.abs().max()
    assert float(logpdf()) > float(lp0)


# ---------------------------------------------------------------------------------------------
# Gradients of the pseudo-point bound (VFE / DTC): kernel hyper-parameters, per-point noise,
# inducing inputs z and y, against central finite differences of a direct-distance restatement of
# the oracle's ELBO (``oracle.pseudo_obs``; same reason for direct distances as above).
# ---------------------------------------------------------------------------------------------
def elbo_direct(terms, x, noise_vec, y, z, method, eps):
    import scipy.linalg as sl

    def km(a, b):
        d2 = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
        return sum(v * O._kappa(kind, d2 / s**2, (a @ b.T) / s**2) for kind, v, s in terms)

    m = z.shape[0]
    l_z = np.linalg.cholesky(km(z, z) + eps * np.eye(m))
    v = sl.solve_triangular(l_z, km(z, x), lower=True)
    a = np.eye(m) + (v / noise_vec) @ v.T
    l_a = np.linalg.cholesky(a + eps * np.eye(m))
    u = sl.solve_triangular(l_a, (v / noise_vec) @ y, lower=True)
    trace = 0.0
    if method in ("vfe", "fitc"):
        kd = np.array([float(km(x[i:i + 1], x[i:i + 1])[0, 0]) for i in range(x.shape[0])])
        if method == "vfe":
            trace = np.sum((kd - (v * v).sum(0)) / noise_vec)
        else:
            noise_vec = noise_vec + kd - (v * v).sum(0)
            a = np.eye(m) + (v / noise_vec) @ v.T
            l_a = np.linalg.cholesky(a + eps * np.eye(m))
            u = sl.solve_triangular(l_a, (v / noise_vec) @ y, lower=True)
    return -0.5 * (np.sum(np.log(2 * np.pi * noise_vec)) + 2 * np.sum(np.log(np.diag(l_a)))
                   + np.sum(y[:, 0] ** 2 / noise_vec) - np.sum(u**2) + trace)


def run_elbo_case(dev, dtype, kinds, n, m, d, method, seed, tol):
    rng = np.random.default_rng(seed)
    x, z = rng.standard_normal((n, d)), rng.standard_normal((m, d))
    y = rng.standard_normal((n, 1))
    var0, sc0 = rng.uniform(0.5, 1.5, len(kinds)), rng.uniform(0.8, 1.7, len(kinds))
    nz0 = rng.uniform(0.1, 0.4, n)
    eps = 1e-10
    nk = len(kinds)
    nz_idx, z_idx, y_idx = [0, n // 2, n - 1], [(0, 0), (m // 2, d - 1), (m - 1, 0)], [1, n // 3]

    def unpack(p):
        terms = [(k, p[i], p[nk + i]) for i, k in enumerate(kinds)]
        nzv, zz, yy = nz0.copy(), z.copy(), y.copy()
        o = 2 * nk
        for i in nz_idx:
            nzv[i] = p[o]; o += 1
        for (i, c) in z_idx:
            zz[i, c] = p[o]; o += 1
        for i in y_idx:
            yy[i, 0] = p[o]; o += 1
        return terms, nzv, zz, yy

    def oracle(p):
        terms, nzv, zz, yy = unpack(p)
        return float(elbo_direct(terms, x, nzv, yy, zz, method, eps))

    p0 = np.concatenate([var0, sc0, nz0[nz_idx], [z[i, c] for i, c in z_idx], y[y_idx, 0]])
    t0 = [(k, var0[i], sc0[i]) for i, k in enumerate(kinds)]
    want = O.pseudo_obs(t0, x, nz0, y, z, method=method, eps=eps)["elbo"]
    assert abs(oracle(p0) - want) <= 1e-6 * abs(want)
    ref = fd_grad(oracle, p0, h=1e-6)

    old = st.B.epsilon
    st.B.epsilon = eps
    try:
        vs = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in var0]
        ss = [torch.tensor(s, dtype=torch.float64, requires_grad=True) for s in sc0]
        nz = torch.tensor(nz0, dtype=dtype, device=dev, requires_grad=True)
        tz = torch.tensor(z, dtype=dtype, device=dev, requires_grad=True)
        ty = torch.tensor(y, dtype=dtype, device=dev, requires_grad=True)
        kernel = sum(v * KINDS[k]().stretch(s) for v, k, s in zip(vs, kinds, ss))
        f = st.GP(kernel)
        cls = {"vfe": st.PseudoObs, "dtc": st.PseudoObsDTC, "fitc": st.PseudoObsFITC}[method]
        obs = cls(f(tz), f(torch.tensor(x, dtype=dtype, device=dev), nz), ty)
        elbo = obs.elbo(f.measure)
        assert elbo.requires_grad and abs(float(elbo) - want) <= tol * abs(want)
        elbo.backward()
    finally:
        st.B.epsilon = old
    got = np.array([float(v.grad) for v in vs] + [float(s.grad) for s in ss]
                   + [float(nz.grad[i]) for i in nz_idx] + [float(tz.grad[i, c]) for i, c in z_idx]
                   + [float(ty.grad[i, 0]) for i in y_idx])
    assert np.max(np.abs(got - ref)) <= tol * max(np.max(np.abs(ref)), 1.0), (got, ref)
    # without gradients the cached, non-differentiable path is used and agrees
    with torch.no_grad():
        plain = cls(f(tz.detach()), f(torch.tensor(x, dtype=dtype, device=dev), nz.detach()), ty.detach())
        st.B.epsilon = eps
        try:
            val = plain.elbo(f.measure)
        finally:
            st.B.epsilon = old
    assert not val.requires_grad and abs(float(val) - float(elbo)) <= 1e-9 * abs(want) + tol * 1e-3


ELBO_CASES = [(("eq",), 70, 9, 2, "vfe"), (("eq",), 70, 9, 2, "dtc"), (("eq",), 70, 9, 2, "fitc"),
              (("matern32", "linear"), 80, 10, 3, "fitc"), (("matern32", "linear"), 90, 12, 3, "vfe"),
              (("matern52", "matern12"), 64, 7, 1, "vfe"), (("eq", "const"), 50, 5, 8, "dtc")]
KINDS["const"] = st.OneKernel


@pytest.mark.parametrize("kinds,n,m,d,method", ELBO_CASES)
def test_elbo_gradients_host_logic(oracle_backend, kinds, n, m, d, method):
    run_elbo_case("cpu", torch.float64, kinds, n, m, d, method, seed=n + m, tol=5e-6)


def test_elbo_gradients_unsupported_cases_are_loud(oracle_backend):
    x = torch.randn(30, 2, dtype=torch.float64)
    y = torch.randn(30, 1, dtype=torch.float64)
    v = torch.tensor(1.0, dtype=torch.float64, requires_grad=True)
    f = st.GP(v * st.EQ())
    # logpdf cases outside the differentiable path refuse instead of returning a detached value
    lp_b = f(torch.randn(3, 20, 2, dtype=torch.float64), 0.1).logpdf(torch.randn(3, 20, 1, dtype=torch.float64))
    assert lp_b.shape == (3,) and lp_b.requires_grad        # batched: differentiable since round 2 (tests/test_round2_regressions.py)
    with pytest.raises(NotImplementedError):        # batched with several columns of y is not
        f(torch.randn(3, 20, 2, dtype=torch.float64), 0.1).logpdf(torch.randn(3, 20, 2, dtype=torch.float64))
    g = st.GP(st.Matern32(), measure=f.measure)
    # several processes observed jointly: differentiable since round 2 (tests/test_autograd_inputs.py) ...
    assert f.measure.logpdf((f(x[:7], 0.1), y[:7]), (g(x[7:12], 0.1), y[7:12])).requires_grad
    dense_noise = torch.eye(7, dtype=torch.float64) * 0.1
    assert f(x[:7], dense_noise).logpdf(y[:7]).requires_grad       # ... and so is a dense noise covariance
    post = f | (f(x[:7], 0.1), y[:7])
    assert post(x[7:12], 0.1).logpdf(y[7:12]).requires_grad          # ... and a posterior log-density (chain rule over two prior ones)
    sparse_post = f | st.PseudoObs(f(x[:4]), f(x[:9], 0.1), y[:9])
    with pytest.raises(NotImplementedError):        # a log-density under a PSEUDO-POINT posterior is not
        sparse_post(x[9:14], 0.1).logpdf(y[9:14])
    with torch.no_grad():                           # ... unless gradients are off
        assert torch.isfinite(sparse_post(x[9:14], 0.1).logpdf(y[9:14]))
    with pytest.raises(NotImplementedError):        # noisy inducing points
        st.PseudoObs(f(x[:5], 0.01), f(x, 0.1), y).elbo(f.measure)


@pytest.mark.gpu
@pytest.mark.parametrize("kinds,n,m,d,method", ELBO_CASES + [(("eq",), 1500, 200, 8, "vfe"), (("eq", "linear"), 900, 130, 4, "dtc")])
def test_elbo_gradients_gpu(hip_backend, kinds, n, m, d, method):
    run_elbo_case("cuda", torch.float64, kinds, n, m, d, method, seed=n + m, tol=2e-5)


def test_readme_example13_flow_host_logic(oracle_backend):
    """The reference's torch learning example, line for line on the API level
    (``readme_example13_optimisation_torch.py:9-60``): nn.Parameters, predictions before and after,
    ``logpdf`` in an Adam loop."""
    from stheno_amd.torch import EQ, GP, B as B_

    old = B_.epsilon
    B_.epsilon = 1e-6
    try:
        torch.manual_seed(0)
        x = torch.linspace(0, 2, 40, dtype=torch.float64)
        x_obs = torch.linspace(0, 2, 25, dtype=torch.float64)
        y_obs = torch.sin(5 * x_obs) + 0.05 ** 0.5 * torch.randn(25, dtype=torch.float64)

        class Model(torch.nn.Module):
            def __init__(self, init_var=0.3, init_scale=1.0, init_noise=0.2):
                super().__init__()
                self.log_var = torch.nn.Parameter(torch.log(torch.tensor(init_var, dtype=torch.float64)))
                self.log_scale = torch.nn.Parameter(torch.log(torch.tensor(init_scale, dtype=torch.float64)))
                self.log_noise = torch.nn.Parameter(torch.log(torch.tensor(init_noise, dtype=torch.float64)))

            def construct(self):
                kernel = torch.exp(self.log_var) * EQ().stretch(torch.exp(self.log_scale))
                return GP(kernel), torch.exp(self.log_noise)

        model = Model()
        f, noise = model.construct()
        f_post = f | (f(x_obs, noise), y_obs)
        mean0, lo0, hi0 = f_post(x, noise).marginal_credible_bounds()
        assert mean0.shape == lo0.shape == hi0.shape == (40,)
        opt = torch.optim.Adam(model.parameters(), lr=5e-2)
        losses = []
        for _ in range(60):
            opt.zero_grad()
            f, noise = model.construct()
            loss = -f(x_obs, noise).logpdf(y_obs)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        assert losses[-1] < losses[0] - 1.0
        f, noise = model.construct()
        f_post = f | (f(x_obs, noise), y_obs)
        mean1, lo1, hi1 = f_post(x, noise).marginal_credible_bounds()
        err0 = float((mean0.detach() - torch.sin(5 * x)).abs().mean())
        err1 = float((mean1.detach() - torch.sin(5 * x)).abs().mean())
        assert err1 < err0
    finally:
        B_.epsilon = old
