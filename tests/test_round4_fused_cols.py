"""``gpk_gemm_colscale`` (round 4): the column scaling ``K_n^{-1/2}`` and the column sums of squares ``Q_x_diag`` of
``V = L_z^{-1} K_zx`` folded into the store of the GEMM that forms ``V`` (``stheno/model/observations.py:301, 305, 322, 327``).
Op level against torch in fp64, model level (VFE / DTC / FITC bounds where the whole factor is inverted: M >= 1024, N >= 4 M)
against the oracle."""
import numpy as np
import pytest
import torch

import stheno_amd as st
from oracle import gp_oracle as O
from stheno_amd import B, ops

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("hip_backend")]
DEV = torch.device("cuda")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
@pytest.mark.parametrize("m,n", [(1024, 4096), (1152, 5037), (2048 + 64, 9000)])
def test_gemm_colscale_against_torch(dtype, m, n):
    be = ops.get_backend()
    g = torch.Generator().manual_seed(m + n)
    w = torch.tril(torch.randn(m, m, generator=g, dtype=torch.float64)).to(dtype).to(DEV)        # lower triangular A (k-contiguous)
    b = torch.randn(m, n, generator=g, dtype=torch.float64).to(dtype).to(DEV)                    # B stored K x N
    s = (0.5 + torch.rand(n, generator=g, dtype=torch.float64)).to(dtype).to(DEV)
    ref = w.double() @ b.double()
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    for scale, want in ((s, True), (None, True), (s, False)):
        out, ss = be.gemm_colscale(w, b, scale, want_colss=want, a_kmajor=True, b_kmajor=False, tri_k_lower=True)
        want_out = ref * s.double()[None, :] if scale is not None else ref
        assert float((out.double() - want_out).abs().max() / want_out.abs().max()) < tol
        if want:
            want_ss = (ref * ref).sum(0)
            assert float((ss.double() - want_ss).abs().max() / want_ss.abs().max()) < tol * 10
        else:
            assert ss is None
    # the unfused passes give the same numbers
    plain = be.gemm(w, b, a_kmajor=True, b_kmajor=False, tri_k_lower=True)
    out, _ = be.gemm_colscale(w, b, None, want_colss=False, a_kmajor=True, b_kmajor=False, tri_k_lower=True)
    assert torch.equal(out, plain)


@pytest.mark.parametrize("method", ["vfe", "dtc", "fitc"])
def test_pseudo_point_bounds_through_the_fused_epilogue(method):
    rng = np.random.default_rng(17)
    n, m, d = 6001, 1024, 3
    x, z = rng.standard_normal((n, d)), rng.standard_normal((m, d)) * 1.5
    y = rng.standard_normal((n, 1))
    noise = rng.uniform(0.05, 0.3, size=n)
    terms = [("eq", 1.3, 0.9)]
    eps0 = B.epsilon
    try:
        B.epsilon = 1e-10
        ref = O.pseudo_obs(terms, x, noise, y, z, method=method, eps=1e-10)
        prior = st.Measure()
        f = st.GP(1.3 * st.EQ().stretch(0.9), measure=prior)
        cls = {"vfe": st.PseudoObs, "dtc": st.PseudoObsDTC, "fitc": st.PseudoObsFITC}[method]
        tx, tz, ty, tn = (torch.as_tensor(a, device=DEV) for a in (x, z, y, noise))
        obs = cls(f(tz), f(tx, tn), ty)
        elbo = float(obs.elbo(prior))
        assert abs(elbo - ref["elbo"]) <= 1e-6 * abs(ref["elbo"]), (method, elbo, ref["elbo"])
        mu = obs.mu(prior).reshape(-1).cpu().numpy()
        assert np.max(np.abs(mu - ref["mu"][:, 0])) <= 1e-6 * np.max(np.abs(ref["mu"]))
    finally:
        B.epsilon = eps0
