"""Seeded sweep of ragged shapes through the whole path on HIP against the CPU oracle: sizes that straddle every tile
edge (N not a multiple of 16 / 64 / 128, D from 1 to 11, 1-4 kernel terms of mixed kinds, scalar / per-point noise,
1-3 columns of y, N* ragged too), dense logpdf + posterior marginals, and the pseudo-point bound + posterior with ragged
M.  The golden fixtures pin a handful of round shapes; this pins the edges of the kernels' bounds-checked variants.

Tolerances: the stated ones (1e-6 fp64 / 1e-3 fp32, norm-wise for vectors).
"""
import numpy as np
import pytest
import torch

import stheno_amd as st
from oracle import gp_oracle as O

from .test_gpu_parity import EPS, KINDS, TOL, dev, eps, rel

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("hip_backend")]

ALL_KINDS = ["eq", "matern12", "matern32", "matern52", "linear"]


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([1, 2, 15, 17, 63, 65, 127, 129, 200, 255, 257, 383, 500, 641, 777]))
    d = int(rng.integers(1, 12))
    nt = int(rng.integers(1, 5))
    kinds = [str(k) for k in rng.choice(ALL_KINDS, size=nt, replace=True)]
    terms = [(k, float(rng.uniform(0.3, 1.5)), float(rng.uniform(0.6, 2.5)) * np.sqrt(d)) for k in kinds]
    c = int(rng.choice([1, 1, 2, 3]))
    ns = int(rng.choice([1, 7, 64, 130, 301]))
    per_point = bool(rng.integers(0, 2))
    x, xs = rng.standard_normal((n, d)), rng.standard_normal((ns, d))
    y = rng.standard_normal((n, c))
    noise = rng.uniform(0.05, 0.5, n) if per_point else float(rng.uniform(0.05, 0.5))
    return terms, x, xs, y, noise


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("seed", range(24))
def test_dense_ragged_shapes(seed, dtype):
    terms, x, xs, y, noise = _case(seed)
    tol, e = TOL[dtype], EPS[dtype]
    with eps(e):
        kern = sum(v * KINDS[k]().stretch(s) for k, v, s in terms)
        f = st.GP(kern)
        nz = dev(noise, dtype) if isinstance(noise, np.ndarray) else noise
        lp = f(dev(x, dtype), nz).logpdf(dev(y, dtype))
        want = np.atleast_1d(O.gp_logpdf(terms, x, noise, y, eps=e))
        assert rel(lp.reshape(-1), want) < tol, (terms, x.shape, y.shape)
        post = f | (f(dev(x, dtype), nz), dev(y[:, :1], dtype))
        mean, var = post(dev(xs, dtype)).marginals()
        rm, _, rv = O.gp_posterior(terms, x, noise, y[:, :1], xs, full_cov=False, eps=e)
        # (norm-wise against the prior scale for the variance: where the data pin the function it is a difference of O(1) numbers)
        prior = sum(v for _, v, _ in terms) * max(1.0, float((xs ** 2).sum(-1).max()) if any(k == "linear" for k, _, _ in terms) else 1.0)
        # (norm-wise against the scale of the data: a single test point far from the data has a posterior mean near zero)
        assert float(np.max(np.abs(mean.cpu().double().numpy() - rm))) < tol * max(float(np.max(np.abs(rm))), float(np.max(np.abs(y))))
        assert float(np.max(np.abs(var.cpu().double().numpy() - np.maximum(rv, 0)))) < tol * prior


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("seed", range(10))
def test_pseudo_point_ragged_shapes(seed, dtype):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([130, 257, 500, 1023, 1500]))
    m = int(rng.choice([1, 3, 17, 64, 129, 200]))
    d = int(rng.integers(1, 9))
    method = str(rng.choice(["vfe", "fitc", "dtc"]))
    kinds = [str(k) for k in rng.choice(["eq", "matern32", "matern52"], size=int(rng.integers(1, 3)), replace=True)]
    terms = [(k, float(rng.uniform(0.5, 1.5)), float(rng.uniform(0.8, 2.0)) * np.sqrt(d)) for k in kinds]
    x, z = rng.standard_normal((n, d)), rng.standard_normal((m, d))
    y = rng.standard_normal((n, 1))
    noise = rng.uniform(0.1, 0.5, n)
    # fp32 at its jitter 1e-6: a well-conditioned K_z is needed for the stated 1e-3 (random z in d >= 1 is)
    tol, e = TOL[dtype], (1e-10 if dtype == torch.float64 else 1e-6)
    with eps(e):
        kern = sum(v * KINDS[k]().stretch(s) for k, v, s in terms)
        f = st.GP(kern)
        cls = {"vfe": st.PseudoObs, "fitc": st.PseudoObsFITC, "dtc": st.PseudoObsDTC}[method]
        obs = cls(f(dev(z, dtype)), f(dev(x, dtype), dev(noise, dtype)), dev(y, dtype))
        ref = O.pseudo_obs(terms, x, noise, y, z, method=method, eps=e)
        assert abs(float(obs.elbo(f.measure)) - ref["elbo"]) <= tol * abs(ref["elbo"]), (method, n, m, d)
        assert rel(obs.mu(f.measure), ref["mu"]) < tol


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("seed", range(8))
def test_batched_ragged_shapes(seed, dtype):
    """Leading batch axis (``tests/model/test_cases.py:134-155``) with ragged B, N, D: one launch sequence for all data sets."""
    rng = np.random.default_rng(9000 + seed)
    bsz = int(rng.choice([1, 3, 8, 17, 70]))
    n = int(rng.choice([5, 64, 129, 200, 333]))
    d = int(rng.integers(1, 9))
    kinds = [str(k) for k in rng.choice(ALL_KINDS, size=int(rng.integers(1, 4)), replace=True)]
    terms = [(k, float(rng.uniform(0.3, 1.5)), float(rng.uniform(0.6, 2.5)) * np.sqrt(d)) for k in kinds]
    x, y = rng.standard_normal((bsz, n, d)), rng.standard_normal((bsz, n, 1))
    noise = float(rng.uniform(0.05, 0.5))
    tol, e = TOL[dtype], EPS[dtype]
    with eps(e):
        f = st.GP(sum(v * KINDS[k]().stretch(s) for k, v, s in terms))
        lp = f(dev(x, dtype), noise).logpdf(dev(y, dtype))
        assert lp.shape == (bsz,)
        assert rel(lp, O.gp_logpdf_batched(terms, x, noise, y, eps=e)) < tol, (terms, x.shape)
