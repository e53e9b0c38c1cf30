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
    """A posterior log-density under learnable per-dimension length scales is not covered: it refuses (no detached value)."""
    x = torch.randn(20, 2, dtype=torch.float64)
    y = torch.randn(20, 1, dtype=torch.float64)
    ls = torch.tensor([0.9, 1.3], dtype=torch.float64, requires_grad=True)
    f = st.GP(st.EQ().stretch(ls))
    post = f | (f(x[:10], 0.1), y[:10])
    with pytest.raises(NotImplementedError):
        post(x[10:], 0.1).logpdf(y[10:])
    with torch.no_grad():
        assert torch.isfinite(post(x[10:], 0.1).logpdf(y[10:]))
    # more than 8 input dimensions: refused up front
    x9 = torch.randn(12, 9, dtype=torch.float64, requires_grad=True)
    with pytest.raises(NotImplementedError):
        st.GP(st.EQ())(x9, 0.1).logpdf(torch.randn(12, 1, dtype=torch.float64))
