"""BASELINE.json's configurations at FULL size on the MI355X, checked through
size-independent properties (the oracle cannot finish these sizes in seconds) plus
oracle comparisons at a reduced N drawn from the same generator.

Properties used:
 * factor residual: rows of ``L L^T`` reproduce rows of ``K + sigma^2 I``;
 * solve residual: ``L (L^{-1} b) = b``;
 * chain rule: ``logpdf(y) = logpdf(y_1) + logpdf(y_2 | y_1)`` with the conditional
   evaluated through the posterior kernel path (a different code path: GEMM-updated
   posterior covariance, second factorisation);
 * marginal variance = diagonal of the full posterior covariance on a subset;
 * fp32 vs fp64 agreement of the same computation;
 * batched == loop over single GPs; sharded == unsharded.
"""
import numpy as np
import pytest
import torch

import stheno_amd as st
from bench import NOISE, make_inputs, make_step
from oracle import gp_oracle as O
from stheno_amd import B, ops

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("hip_backend")]
DEV = torch.device("cuda")


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def test_config2_dense_f64_n16384():
    w, t = make_inputs("dense_f64", DEV)
    assert t["x"].shape == (16384, 8) and t["x"].dtype == torch.float64
    lp, mean, var = make_step("dense_f64", w, t)()
    assert torch.isfinite(lp) and torch.isfinite(mean).all() and (var >= 0).all() and (var <= 1.0 + 1e-9).all()

    # factor + solve residuals on the factor the model layer cached
    f = st.GP(st.EQ())
    fdd = f(t["x"], NOISE)
    lp2 = fdd.logpdf(t["y"])
    assert abs(float(lp2) - float(lp)) <= 1e-12 * abs(float(lp))          # deterministic
    chol = fdd.var.chol()
    L = chol.lower()
    rows = torch.tensor([0, 1, 127, 128, 129, 5000, 8191, 8192, 12345, 16383], device=DEV)
    k_rows = f.kernel.pairwise(t["x"][rows], t["x"])                       # (10, N)
    k_rows[torch.arange(10, device=DEV), rows] += NOISE + B.epsilon
    llt_rows = L[rows] @ L.T
    assert rel(llt_rows, k_rows) < 1e-12
    b = torch.randn(16384, 3, dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))
    assert rel(L @ chol.solve(b), b) < 1e-10
    b2 = torch.randn(16384, 64, dtype=torch.float64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
    assert rel(L @ chol.solve(b2), b2) < 1e-10                             # blocked TRSM path (merged inverses)

    # marginal variance == diagonal of the full posterior covariance (subset of test points)
    post = f | (fdd, t["y"])
    sub = t["xs"][:256]
    full = B.dense(post(sub).var)
    assert rel(torch.diagonal(full), var[:256]) < 1e-9
    assert rel(post(sub).mean[:, 0], mean[:256]) < 1e-12


def test_config2_chain_rule_and_oracle_at_reduced_n():
    w, t = make_inputs("dense_f64", DEV)
    n1 = 4096
    x1, y1, x2, y2 = t["x"][:n1], t["y"][:n1], t["x"][n1:2 * n1], t["y"][n1:2 * n1]
    f = st.GP(st.EQ())
    joint = f(t["x"][:2 * n1], NOISE).logpdf(t["y"][:2 * n1])
    post = f | (f(x1, NOISE), y1)
    chain = f(x1, NOISE).logpdf(y1) + post(x2, NOISE).logpdf(y2)
    assert abs(float(chain) - float(joint)) <= 1e-9 * abs(float(joint))
    # oracle at N = 2048 on the same draws: the 1e-6 parity bar of north_star
    n0 = 2048
    xo, yo, xso = (a.cpu().numpy() for a in (t["x"][:n0], t["y"][:n0], t["xs"][:200]))
    ref_lp = O.gp_logpdf([("eq", 1.0, 1.0)], xo, NOISE, yo)
    ref_m, _, ref_v = O.gp_posterior([("eq", 1.0, 1.0)], xo, NOISE, yo, xso, full_cov=False)
    fd = f(t["x"][:n0], NOISE)
    assert abs(float(fd.logpdf(t["y"][:n0])) - ref_lp) <= 1e-6 * abs(ref_lp)
    m, v = (f | (fd, t["y"][:n0]))(t["xs"][:200]).marginals()
    assert np.max(np.abs(m.cpu().numpy() - ref_m)) <= 1e-6 * np.max(np.abs(ref_m))
    assert np.max(np.abs(v.cpu().numpy() - ref_v)) <= 1e-6 * np.max(np.abs(ref_v))


def test_config3_sum_kernel_f32_n32768():
    B.epsilon = 1e-6
    try:
        w, t = make_inputs("sum_f32", DEV)
        lp, mean, var = make_step("sum_f32", w, t)()
        assert torch.isfinite(lp) and torch.isfinite(mean).all() and torch.isfinite(var).all()
        # fp32 vs fp64 on the same inputs at N = 8192 (the 1e-3 bar)
        n0 = 8192
        k = st.EQ() + st.Linear()
        f = st.GP(k)
        x32, y32, xs32 = t["x"][:n0], t["y"][:n0], t["xs"]
        fd32 = f(x32, NOISE)
        lp32 = fd32.logpdf(y32)
        m32, v32 = (f | (fd32, y32))(xs32).marginals()
        B.epsilon = 1e-12
        fd64 = f(x32.double(), NOISE)
        lp64 = fd64.logpdf(y32.double())
        m64, v64 = (f | (fd64, y32.double()))(xs32.double()).marginals()
        assert abs(float(lp32) - float(lp64)) <= 1e-3 * abs(float(lp64))
        assert rel(m32, m64) < 1e-3 and rel(v32, v64) < 1e-3
        # oracle at N = 1024
        n1 = 1024
        ref = O.gp_logpdf([("eq", 1.0, 1.0), ("linear", 1.0, 1.0)], x32[:n1].double().cpu().numpy(), NOISE,
                          y32[:n1].double().cpu().numpy())
        assert abs(float(f(x32[:n1].double(), NOISE).logpdf(y32[:n1].double())) - ref) <= 1e-6 * abs(ref)
    finally:
        B.epsilon = 1e-12


def test_config4_batched_512x2048_f32():
    B.epsilon = 1e-6
    try:
        w, t = make_inputs("batched_f32", DEV)
        assert t["x"].shape == (512, 2048, 3)
        lp = make_step("batched_f32", w, t)()
        assert lp.shape == (512,) and torch.isfinite(lp).all()
        f = st.GP(st.EQ())
        for b in (0, 17, 511):          # batched == single, and the oracle (fp64) within 1e-3
            single = f(t["x"][b], NOISE).logpdf(t["y"][b])
            assert abs(float(single) - float(lp[b])) <= 1e-5 * abs(float(single))
            ref = O.gp_logpdf([("eq", 1.0, 1.0)], t["x"][b].double().cpu().numpy(), NOISE,
                              t["y"][b].double().cpu().numpy(), eps=1e-6)
            assert abs(float(lp[b]) - ref) <= 1e-3 * abs(ref)
        # a sharded run of two "ranks" in one process equals the unsharded vector
        from stheno_amd.dist import shard_bounds
        parts = []
        for r in range(2):
            lo, hi = shard_bounds(512, 2, r)
            parts.append(f(t["x"][lo:hi], NOISE).logpdf(t["y"][lo:hi]))
        assert rel(torch.cat(parts), lp) < 1e-6
    finally:
        B.epsilon = 1e-12


def test_config5_sparse_vfe_f32_n200000_m4096():
    B.epsilon = 1e-6
    try:
        w, t = make_inputs("sparse_f32", DEV)
        elbo32 = make_step("sparse_f32", w, t)()
        assert torch.isfinite(elbo32)
        # the same ELBO in fp64 on the device (fp32 bar: 1e-3)
        B.epsilon = 1e-10
        prior = st.Measure()
        f = st.GP(st.EQ(), measure=prior)
        elbo64 = st.PseudoObs(f(t["z"].double()), f(t["x"].double(), NOISE), t["y"].double()).elbo(prior)
        assert abs(float(elbo32) - float(elbo64)) <= 1e-3 * abs(float(elbo64))
        # oracle at N = 20000, M = 512 on the same draws (fp64 bar: 1e-6)
        n0, m0 = 20000, 512
        xo, yo, zo = (a.double().cpu().numpy() for a in (t["x"][:n0], t["y"][:n0], t["z"][:m0]))
        ref = O.pseudo_obs([("eq", 1.0, 1.0)], xo, NOISE, yo, zo, eps=1e-10)["elbo"]
        prior2 = st.Measure()
        f2 = st.GP(st.EQ(), measure=prior2)
        got = st.PseudoObs(f2(t["z"][:m0].double()), f2(t["x"][:n0].double(), NOISE), t["y"][:n0].double()).elbo(prior2)
        assert abs(float(got) - ref) <= 1e-6 * abs(ref)
        # the ELBO is a lower bound on the exact log-density (checked where the exact one is cheap)
        n1 = 4096
        prior3 = st.Measure()
        f3 = st.GP(st.EQ(), measure=prior3)
        exact = f3(t["x"][:n1].double(), NOISE).logpdf(t["y"][:n1].double())
        bound = st.PseudoObs(f3(t["z"][:m0].double()), f3(t["x"][:n1].double(), NOISE), t["y"][:n1].double()).elbo(prior3)
        assert float(bound) <= float(exact) + 1e-6 * abs(float(exact))
    finally:
        B.epsilon = 1e-12
