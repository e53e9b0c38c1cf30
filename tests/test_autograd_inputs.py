"""Gradients with respect to the INPUTS of a GP and to per-dimension length scales (``k.stretch(vector)``: the
reference differentiates everything through lab/torch, ``readme_example13_optimisation_torch.py:47-53``): dense
log-density (unbatched, batched) and the pseudo-point bounds, against central finite differences of the CPU oracle.

The CPU variants run the autograd.Function host logic on the test-only oracle backend, the GPU variants go through
``libgpk.so`` (``gpk_kmat_vjp_dense`` with an explicit cotangent).
"""
import numpy as np
import pytest
import torch

import stheno_amd as st
from oracle import gp_oracle as O

from .test_autograd import KINDS, elbo_direct, fd_grad, logpdf_direct


def _kernel(vs, kinds, ss, ls=None):
    k = sum(v * KINDS[kd]().stretch(s) for v, kd, s in zip(vs, kinds, ss))
    return k if ls is None else k.stretch(ls)


def run_logpdf_inputs(dev, dtype, kinds, n, d, c, seed, tol, ard):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d))
    y = rng.standard_normal((n, c))
    var0, sc0 = rng.uniform(0.5, 1.5, len(kinds)), rng.uniform(0.7, 1.6, len(kinds))
    ls0 = rng.uniform(0.6, 1.8, d) if ard else np.ones(d)
    noise0 = 0.3
    wts = rng.uniform(0.5, 1.5, c)
    t0 = [(k, var0[i], sc0[i]) for i, k in enumerate(kinds)]
    picks = [(0, 0), (n // 2, d - 1), (n - 1, d // 2)]

    def value(xx, ls):
        return float(np.sum(wts * np.atleast_1d(logpdf_direct(t0, xx / ls, noise0, y))))

    want = value(x, ls0)
    assert abs(want - float(np.sum(wts * np.atleast_1d(O.gp_logpdf(t0, x / ls0, noise0, y))))) <= 1e-7 * abs(want)

    def at(i, cdim, h):
        xx = x.copy(); xx[i, cdim] += h
        return value(xx, ls0)

    ref_x = np.array([(at(i, cd, 1e-6) - at(i, cd, -1e-6)) / 2e-6 for i, cd in picks])
    ref_ls = fd_grad(lambda ls: value(x, ls), ls0.copy()) if ard else None

    vs = [torch.tensor(v, dtype=torch.float64) for v in var0]
    ss = [torch.tensor(s, dtype=torch.float64) for s in sc0]
    tx = torch.tensor(x, dtype=dtype, device=dev, requires_grad=True)
    tls = torch.tensor(ls0, dtype=torch.float64, requires_grad=True) if ard else None
    f = st.GP(_kernel(vs, kinds, ss, tls))
    lp = f(tx, noise0).logpdf(torch.tensor(y, dtype=dtype, device=dev))
    assert abs(float((lp.detach().reshape(-1).double().cpu() * torch.tensor(wts)).sum()) - want) <= tol * abs(want)
    (lp.reshape(-1) * torch.tensor(wts, dtype=dtype, device=dev)).sum().backward()
    got_x = np.array([float(tx.grad[i, cd]) for i, cd in picks])
    assert np.max(np.abs(got_x - ref_x)) <= tol * max(np.max(np.abs(ref_x)), 1.0), (got_x, ref_x)
    # every row at once: the sum over rows of d/dx equals the derivative along a common shift -- zero for stationary kernels
    if all(k != "linear" for k in kinds):
        assert float(tx.grad.sum(0).abs().max()) <= 100 * tol * float(tx.grad.abs().max())
    if ard:
        got_ls = tls.grad.numpy()
        assert np.max(np.abs(got_ls - ref_ls)) <= tol * max(np.max(np.abs(ref_ls)), 1.0), (got_ls, ref_ls)


INPUT_CASES = [(("eq",), 40, 3, 1, False), (("eq", "linear"), 37, 2, 2, False), (("matern52",), 45, 1, 1, False),
               (("eq",), 40, 3, 1, True), (("matern32", "linear"), 50, 4, 2, True)]


@pytest.mark.parametrize("kinds,n,d,c,ard", INPUT_CASES)
def test_logpdf_input_gradients_host_logic(oracle_backend, kinds, n, d, c, ard):
    run_logpdf_inputs("cpu", torch.float64, kinds, n, d, c, seed=3 * n + d, tol=2e-6, ard=ard)


@pytest.mark.gpu
@pytest.mark.parametrize("kinds,n,d,c,ard", INPUT_CASES + [(("eq",), 700, 8, 1, True), (("eq", "linear"), 300, 4, 2, False)])
def test_logpdf_input_gradients_gpu(hip_backend, kinds, n, d, c, ard):
    run_logpdf_inputs("cuda", torch.float64, kinds, n, d, c, seed=3 * n + d, tol=5e-6, ard=ard)


@pytest.mark.gpu
def test_logpdf_input_gradients_gpu_fp32(hip_backend):
    st.B.epsilon = 1e-6
    try:
        run_logpdf_inputs("cuda", torch.float32, ("eq",), 200, 3, 1, seed=11, tol=5e-3, ard=True)
    finally:
        st.B.epsilon = 1e-12


def run_batched_inputs(dev, tol):
    """(B, N, D) inputs with shared per-dimension length scales: d/dx per data set, d/d scales summed over the batch."""
    rng = np.random.default_rng(5)
    B, n, d = 3, 30, 2
    x, y = rng.standard_normal((B, n, d)), rng.standard_normal((B, n, 1))
    ls0 = np.array([0.8, 1.4])
    t0 = [("eq", 1.2, 1.0)]
    wts = np.array([0.7, 1.0, 1.3])

    def value(xx, ls):
        return float(sum(wts[b] * logpdf_direct(t0, xx[b] / ls, 0.2, y[b]) for b in range(B)))

    ref_ls = fd_grad(lambda ls: value(x, ls), ls0.copy())
    picks = [(0, 0, 0), (1, n // 2, 1), (2, n - 1, 0)]
    ref_x = []
    for b, i, cd in picks:
        xp, xm = x.copy(), x.copy()
        xp[b, i, cd] += 1e-6; xm[b, i, cd] -= 1e-6
        ref_x.append((value(xp, ls0) - value(xm, ls0)) / 2e-6)
    tx = torch.tensor(x, dtype=torch.float64, device=dev, requires_grad=True)
    tls = torch.tensor(ls0, dtype=torch.float64, requires_grad=True)
    f = st.GP(1.2 * st.EQ().stretch(tls))
    lp = f(tx, 0.2).logpdf(torch.tensor(y, dtype=torch.float64, device=dev))
    assert lp.shape == (B,)
    (lp * torch.tensor(wts, dtype=torch.float64, device=dev)).sum().backward()
    got_x = np.array([float(tx.grad[b, i, cd]) for b, i, cd in picks])
    assert np.max(np.abs(got_x - np.array(ref_x))) <= tol * max(np.max(np.abs(ref_x)), 1.0)
    assert np.max(np.abs(tls.grad.numpy() - ref_ls)) <= tol * max(np.max(np.abs(ref_ls)), 1.0)


def test_batched_input_gradients_host_logic(oracle_backend):
    run_batched_inputs("cpu", 2e-6)


@pytest.mark.gpu
def test_batched_input_gradients_gpu(hip_backend):
    run_batched_inputs("cuda", 5e-6)


def run_elbo_inputs(dev, dtype, kinds, n, m, d, method, seed, tol, ard):
    rng = np.random.default_rng(seed)
    x, z = rng.standard_normal((n, d)), rng.standard_normal((m, d))
    y = rng.standard_normal((n, 1))
    var0, sc0 = rng.uniform(0.5, 1.5, len(kinds)), rng.uniform(0.8, 1.7, len(kinds))
    ls0 = rng.uniform(0.7, 1.6, d) if ard else np.ones(d)
    nz0 = rng.uniform(0.1, 0.4, n)
    eps = 1e-10
    t0 = [(k, var0[i], sc0[i]) for i, k in enumerate(kinds)]
    picks_x = [(0, 0), (n // 2, d - 1), (n - 1, 0)]
    picks_z = [(0, d - 1), (m - 1, 0)]

    def value(xx, zz, ls):
        return float(elbo_direct(t0, xx / ls, nz0, y, zz / ls, method, eps))

    want = value(x, z, ls0)
    assert abs(want - O.pseudo_obs(t0, x / ls0, nz0, y, z / ls0, method=method, eps=eps)["elbo"]) <= 1e-6 * abs(want)

    def fd_entry(arr_name, i, cd):
        vals = []
        for h in (1e-6, -1e-6):
            xx, zz = x.copy(), z.copy()
            (xx if arr_name == "x" else zz)[i, cd] += h
            vals.append(value(xx, zz, ls0))
        return (vals[0] - vals[1]) / 2e-6

    ref_x = np.array([fd_entry("x", i, cd) for i, cd in picks_x])
    ref_z = np.array([fd_entry("z", i, cd) for i, cd in picks_z])
    ref_ls = fd_grad(lambda ls: value(x, z, ls), ls0.copy()) if ard else None

    old = st.B.epsilon
    st.B.epsilon = eps
    try:
        vs = [torch.tensor(v, dtype=torch.float64) for v in var0]
        ss = [torch.tensor(s, dtype=torch.float64) for s in sc0]
        tx = torch.tensor(x, dtype=dtype, device=dev, requires_grad=True)
        tz = torch.tensor(z, dtype=dtype, device=dev, requires_grad=True)
        tls = torch.tensor(ls0, dtype=torch.float64, requires_grad=True) if ard else None
        f = st.GP(_kernel(vs, kinds, ss, tls))
        cls = {"vfe": st.PseudoObs, "dtc": st.PseudoObsDTC, "fitc": st.PseudoObsFITC}[method]
        obs = cls(f(tz), f(tx, torch.tensor(nz0, dtype=dtype, device=dev)), torch.tensor(y, dtype=dtype, device=dev))
        elbo = obs.elbo(f.measure)
        assert elbo.requires_grad and abs(float(elbo) - want) <= tol * abs(want)
        elbo.backward()
    finally:
        st.B.epsilon = old
    got_x = np.array([float(tx.grad[i, cd]) for i, cd in picks_x])
    got_z = np.array([float(tz.grad[i, cd]) for i, cd in picks_z])
    assert np.max(np.abs(got_x - ref_x)) <= tol * max(np.max(np.abs(ref_x)), 1.0), (got_x, ref_x)
    assert np.max(np.abs(got_z - ref_z)) <= tol * max(np.max(np.abs(ref_z)), 1.0), (got_z, ref_z)
    if ard:
        got_ls = tls.grad.numpy()
        assert np.max(np.abs(got_ls - ref_ls)) <= tol * max(np.max(np.abs(ref_ls)), 1.0), (got_ls, ref_ls)


ELBO_INPUT_CASES = [(("eq",), 60, 8, 2, "vfe", False), (("eq",), 60, 8, 2, "dtc", True), (("eq",), 60, 8, 2, "fitc", True),
                    (("matern32", "linear"), 70, 9, 3, "fitc", False), (("matern32", "linear"), 70, 9, 3, "vfe", True)]


@pytest.mark.parametrize("kinds,n,m,d,method,ard", ELBO_INPUT_CASES)
def test_elbo_input_gradients_host_logic(oracle_backend, kinds, n, m, d, method, ard):
    run_elbo_inputs("cpu", torch.float64, kinds, n, m, d, method, seed=n + m + d, tol=5e-6, ard=ard)


@pytest.mark.gpu
@pytest.mark.parametrize("kinds,n,m,d,method,ard", ELBO_INPUT_CASES + [(("eq",), 1200, 150, 8, "vfe", True)])
def test_elbo_input_gradients_gpu(hip_backend, kinds, n, m, d, method, ard):
    run_elbo_inputs("cuda", torch.float64, kinds, n, m, d, method, seed=n + m + d, tol=2e-5, ard=ard)


def test_learnable_length_scale_vector_outside_the_differentiable_paths_is_loud(oracle_backend):
    """A log-density under a pseudo-point posterior with learnable per-dimension length scales is not covered: it refuses."""
    x = torch.randn(20, 2, dtype=torch.float64)
    y = torch.randn(20, 1, dtype=torch.float64)
    ls = torch.tensor([0.9, 1.3], dtype=torch.float64, requires_grad=True)
    f = st.GP(st.EQ().stretch(ls))
    post = f | st.PseudoObs(f(x[:4]), f(x[:10], 0.1), y[:10])
    with pytest.raises(NotImplementedError):
        post(x[10:], 0.1).logpdf(y[10:])
    with torch.no_grad():
        assert torch.isfinite(post(x[10:], 0.1).logpdf(y[10:]))
    # more than 8 input dimensions: refused up front
    x9 = torch.randn(12, 9, dtype=torch.float64, requires_grad=True)
    with pytest.raises(NotImplementedError):
        st.GP(st.EQ())(x9, 0.1).logpdf(torch.randn(12, 1, dtype=torch.float64))


# ---------------------------------------------------------------------------------------------
# Several processes of one measure observed jointly: gradients of the joint log-density w.r.t. the hyper-parameters of the
# component kernels, the noises and y, against finite differences of the joint Gaussian built block by block in NumPy.
# ---------------------------------------------------------------------------------------------
def run_joint_logpdf(dev, dtype, tol):
    rng = np.random.default_rng(17)
    n1, n2, d = 25, 35, 2
    x1, x2 = rng.standard_normal((n1, d)), rng.standard_normal((n2, d))
    y1, y2 = rng.standard_normal((n1, 1)), rng.standard_normal((n2, 1))
    p0 = np.array([1.3, 0.9, 0.7, 1.4, 0.2, 0.3])          # v1, s1, v2, s2, noise1, noise2

    def kmat(kind, v, s_, a, b):
        d2 = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
        return v * O._kappa(kind, d2 / s_**2, (a @ b.T) / s_**2)

    def value(p):
        v1, s1, v2, s2, nz1, nz2 = p
        # f1 ~ GP(v1 EQ / s1), f2 ~ GP(v2 Matern32 / s2) independent, f = f1 + f2; observed: f1 at x1, f at x2
        k11 = kmat("eq", v1, s1, x1, x1) + nz1 * np.eye(n1)
        k22 = kmat("eq", v1, s1, x2, x2) + kmat("matern32", v2, s2, x2, x2) + nz2 * np.eye(n2)
        k21 = kmat("eq", v1, s1, x2, x1)
        joint = np.block([[k11, k21.T], [k21, k22]])
        return float(O.normal_logpdf(None, joint, np.concatenate([y1, y2])))

    ref = fd_grad(value, p0.copy())
    ts = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in p0]
    m = st.Measure()
    f1 = st.GP(ts[0] * st.EQ().stretch(ts[1]), measure=m)
    f2 = st.GP(ts[2] * st.Matern32().stretch(ts[3]), measure=m)
    f = f1 + f2
    T_ = lambda a: torch.tensor(a, dtype=dtype, device=dev)  # noqa: E731
    ty2 = T_(y2).requires_grad_(True)
    lp = m.logpdf((f1(T_(x1), ts[4].to(dtype=dtype, device=dev)), T_(y1)), (f(T_(x2), ts[5].to(dtype=dtype, device=dev)), ty2))
    assert lp.requires_grad and abs(float(lp) - value(p0)) <= tol * abs(value(p0))
    lp.backward()
    got = np.array([float(t.grad) for t in ts])
    assert np.max(np.abs(got - ref)) <= tol * max(np.max(np.abs(ref)), 1.0), (got, ref)

    def value_y(yy):
        nonlocal y2
        keep, y2 = y2, yy.reshape(n2, 1)
        try:
            return value(p0)
        finally:
            y2 = keep

    ref_y = fd_grad(value_y, y2.copy().ravel(), h=1e-5)
    assert np.max(np.abs(ty2.grad.double().cpu().numpy().ravel() - ref_y)) <= tol * max(np.max(np.abs(ref_y)), 1.0)


def test_joint_logpdf_gradients_host_logic(oracle_backend):
    run_joint_logpdf("cpu", torch.float64, 2e-6)


@pytest.mark.gpu
def test_joint_logpdf_gradients_gpu(hip_backend):
    run_joint_logpdf("cuda", torch.float64, 5e-6)


def run_dense_noise(dev, dtype, tol):
    """``f(x, N).logpdf(y)`` with a dense noise covariance ``N = c B B^T + 0.1 I``: d/dc (through the matrix), d/d kernel, d/dx."""
    rng = np.random.default_rng(23)
    n, d = 40, 2
    x, y = rng.standard_normal((n, d)), rng.standard_normal((n, 1))
    Bm = rng.standard_normal((n, 3))
    p0 = np.array([1.2, 0.8, 0.5])                          # variance, scale, c

    def value(p, xx=None):
        xx = x if xx is None else xx
        d2 = ((xx[:, None, :] - xx[None, :, :]) ** 2).sum(-1)
        k = p[0] * np.exp(-0.5 * d2 / p[1] ** 2) + p[2] * Bm @ Bm.T + 0.1 * np.eye(n)
        return float(O.normal_logpdf(None, k, y))

    ref = fd_grad(value, p0.copy())
    ts = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in p0]
    tx = torch.tensor(x, dtype=dtype, device=dev, requires_grad=True)
    tb = torch.tensor(Bm, dtype=dtype, device=dev)
    noise = ts[2].to(dtype=dtype, device=dev) * (tb @ tb.T) + 0.1 * torch.eye(n, dtype=dtype, device=dev)
    lp = st.GP(ts[0] * st.EQ().stretch(ts[1]))(tx, noise).logpdf(torch.tensor(y, dtype=dtype, device=dev))
    assert lp.requires_grad and abs(float(lp) - value(p0)) <= tol * abs(value(p0))
    lp.backward()
    got = np.array([float(t.grad) for t in ts])
    assert np.max(np.abs(got - ref)) <= tol * max(np.max(np.abs(ref)), 1.0), (got, ref)
    i, cd = n // 3, 1
    xp, xm = x.copy(), x.copy()
    xp[i, cd] += 1e-6; xm[i, cd] -= 1e-6
    fdx = (value(p0, xp) - value(p0, xm)) / 2e-6
    assert abs(float(tx.grad[i, cd]) - fdx) <= tol * max(abs(fdx), 1.0)


def test_dense_noise_gradients_host_logic(oracle_backend):
    run_dense_noise("cpu", torch.float64, 2e-6)


@pytest.mark.gpu
def test_dense_noise_gradients_gpu(hip_backend):
    run_dense_noise("cuda", torch.float64, 5e-6)


def run_posterior_logpdf(dev, dtype, tol):
    """Predictive log-density ``(f | (f(x1, n1), y1))(x2, n2).logpdf(y2)`` under learnable hyper-parameters (the held-out
    log-likelihood objective): chain rule over two prior log-densities; values against the non-differentiable posterior path,
    gradients against finite differences of the conditional Gaussian."""
    rng = np.random.default_rng(29)
    n1, n2, d = 30, 12, 2
    x1, x2 = rng.standard_normal((n1, d)), rng.standard_normal((n2, d))
    y1, y2 = rng.standard_normal((n1, 1)), rng.standard_normal((n2, 1))
    p0 = np.array([1.1, 0.9, 0.25])                          # variance, scale, noise

    def value(p):
        def km(a, b):
            d2 = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
            return p[0] * np.exp(-0.5 * d2 / p[1] ** 2)
        k11 = km(x1, x1) + p[2] * np.eye(n1)
        k21 = km(x2, x1)
        sol = np.linalg.solve(k11, np.concatenate([y1, k21.T], axis=1))
        mean = k21 @ sol[:, :1]
        cov = km(x2, x2) + p[2] * np.eye(n2) - k21 @ sol[:, 1:]
        return float(O.normal_logpdf(mean, cov, y2))

    ref = fd_grad(value, p0.copy())
    ts = [torch.tensor(v, dtype=torch.float64, requires_grad=True) for v in p0]
    T_ = lambda a: torch.tensor(a, dtype=dtype, device=dev)  # noqa: E731
    f = st.GP(ts[0] * st.EQ().stretch(ts[1]))
    nz = ts[2].to(dtype=dtype, device=dev)
    post = f | (f(T_(x1), nz), T_(y1))
    lp = post(T_(x2), nz).logpdf(T_(y2))
    assert lp.requires_grad and abs(float(lp) - value(p0)) <= tol * abs(value(p0))
    with torch.no_grad():
        plain = post(T_(x2), nz).logpdf(T_(y2))
    assert abs(float(plain) - float(lp)) <= 1e-8 * abs(float(lp))
    lp.backward()
    got = np.array([float(t.grad) for t in ts])
    assert np.max(np.abs(got - ref)) <= tol * max(np.max(np.abs(ref)), 1.0), (got, ref)


def test_posterior_logpdf_gradients_host_logic(oracle_backend):
    run_posterior_logpdf("cpu", torch.float64, 5e-6)


@pytest.mark.gpu
def test_posterior_logpdf_gradients_gpu(hip_backend):
    run_posterior_logpdf("cuda", torch.float64, 1e-5)
