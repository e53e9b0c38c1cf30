"""Round 6: one step of iterative refinement behind the solves against an ill-conditioned factor (``matrix.config.refine_solves``).

The blocked triangular solves multiply by explicit inverses of 512 ... 2048-wide diagonal blocks (``matrix._solve_block``); in the
nearly noise-free regime (``README.md:43-86``: a GP observed without noise, ``B.epsilon`` alone on the diagonal) that loses one to
two orders of magnitude against LAPACK's substitution (``stheno/random.py:272-279`` via ``B.solve``).  Which of two fp64 paths is off
cannot be read from their difference, so the yardstick here is an 80-bit evaluation (``tests/golden/make_golden_illcond.py``).

CPU part: the host logic (when the step is taken, that it is a fixed point for an exact solve, both call orders, the rows path).
GPU part: the HIP path against the 80-bit reference, held to a small multiple of the fp64 oracle's own error.
"""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import stheno_amd as st
from oracle import gp_oracle as O
from stheno_amd import B, matrix, ops

from .conftest import DEVICE, OracleBackend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _golden(n):
    with open(os.path.join(ROOT, "tests", "golden", f"illcond_n{n}.json")) as fh:
        return json.load(fh)


def _inputs(n, noise=1e-6):
    """The generator calls of ``tests/golden/make_golden_illcond.py`` (and of the conditioning sweep in ``test_round5_evidence.py``)."""
    rng = np.random.default_rng(n + int(-np.log10(noise)))
    x = np.sort(rng.uniform(0.0, n / 204.8, size=(n, 1)), axis=0)
    y = np.sin(x) + 0.1 * rng.standard_normal((n, 1))
    xs = rng.uniform(0.0, n / 204.8, size=(64, 1))
    return x, y, xs


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def _rel(a, ref):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    a, ref = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(ref, dtype=np.float64).reshape(-1)
    return float(np.max(np.abs(a - ref)) / np.max(np.abs(ref)))


@pytest.fixture()
def rows_from_128():
    old = (matrix.config.posterior_rows_from, matrix.config.posterior_rows_min_points)
    matrix.config.posterior_rows_from, matrix.config.posterior_rows_min_points = 128, 8
    yield
    matrix.config.posterior_rows_from, matrix.config.posterior_rows_min_points = old


# ----------------------------------------------------------------------------------------------------------------------------------
# when the step is taken
# ----------------------------------------------------------------------------------------------------------------------------------
def test_the_a_priori_bound_uses_what_the_host_knows(oracle_backend):
    x = torch.linspace(0.0, 5.0, 400, dtype=torch.float64)[:, None]
    f = st.GP(2.0 * st.EQ() + st.Linear())
    k = f(x, 1e-3).var
    assert isinstance(k, matrix.KernelDense)
    # n * (the stationary variances) / (noise + epsilon); the linear term stays out of the trace
    assert k.cond_bound() == pytest.approx(400 * 2.0 / (1e-3 + B.epsilon))
    assert not k.wants_refinement()
    assert f(x, 1e-9).var.wants_refinement()
    assert f(x).var.wants_refinement()                          # noise-free: the jitter alone
    assert f(x, torch.full((400,), 1e-9, dtype=torch.float64)).var.cond_bound() is None       # (its minimum lives in a tensor)
    assert not f(x, torch.full((400,), 1e-9, dtype=torch.float64)).var.wants_refinement()
    assert not st.GP(st.EQ())(x.float(), 1e-9).var.wants_refinement()       # fp32: no threshold by default
    old = matrix.config.refine_solves
    try:
        matrix.config.refine_solves = True
        assert k.wants_refinement()
        matrix.config.refine_solves = False
        assert not f(x).var.wants_refinement()
    finally:
        matrix.config.refine_solves = old


def test_noisy_models_solve_once_and_noise_free_ones_refine(any_backend):
    dev = DEVICE[0]
    rng = np.random.default_rng(5)
    x = torch.as_tensor(np.sort(rng.uniform(0, 3, (300, 1)), axis=0), device=dev)
    y = torch.sin(x)
    xs = torch.as_tensor(rng.uniform(0, 3, (20, 1)), device=dev)
    f = st.GP(st.EQ())
    for noise, expect in ((0.1, False), (1e-8, True)):
        fdd = f(x, noise)
        fdd.logpdf(y)
        post = f | (fdd, y)
        post(xs).marginals()
        chol = fdd.var.chol()
        assert chol.refine is expect
        assert (chol.refined > 0) is expect


# ----------------------------------------------------------------------------------------------------------------------------------
# what it computes
# ----------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("order", ["logpdf-first", "posterior-first"])
def test_refined_posterior_matches_the_oracle_in_both_call_orders(oracle_backend, rows_from_128, order):
    """Over the test backend (exact substitution) the step is a fixed point up to rounding: the refined path must reproduce the
    oracle's posterior as closely as the unrefined one does, through the separate solve AND through the rows under the matrix."""
    x, y, xs = _inputs(512)
    terms = [("eq", 1.0, 1.0)]
    ref_mean, _, ref_var = O.gp_posterior(terms, x, 1e-7, y, xs, full_cov=False)
    ref_lp = O.gp_logpdf(terms, x, 1e-7, y)
    tx, ty, txs = (torch.as_tensor(a) for a in (x, y, xs))
    got = {}
    for mode in (False, "auto"):
        old = matrix.config.refine_solves
        matrix.config.refine_solves = mode
        try:
            f = st.GP(st.EQ())
            fdd = f(tx, 1e-7)         # (n / noise = 5e9: past the fp64 threshold)
            if order == "logpdf-first":
                lp = float(fdd.logpdf(ty))
                mean, var = (f | (fdd, ty))(txs).marginals()
            else:
                mean, var = (f | (fdd, ty))(txs).marginals()
                lp = float(fdd.logpdf(ty))
            chol = fdd.var.chol()
            assert chol.refine is (mode == "auto")
            if order == "posterior-first":
                assert chol.rows_under == 64
            if mode == "auto":
                # the single column (log-density and mean: solved once on the device, twice for host tensors -- `solve_residual`
                # remembers nothing about host memory) and the 64 columns / rows
                assert chol.refined in (2, 3)
            got[mode] = (abs(lp - ref_lp) / abs(ref_lp), _rel(mean, ref_mean), _rel(var, ref_var))
        finally:
            matrix.config.refine_solves = old
    # (unrefined, the test backend IS the oracle's arithmetic; refined, it is a second valid fp64 result: kappa eps ~ 5e-7 apart at most)
    for mode in got:
        assert got[mode][0] <= 1e-9 and got[mode][1] <= 1e-6 and got[mode][2] <= 1e-5, got


def test_refinement_repairs_a_perturbed_solve(oracle_backend):
    """The step itself: a backend whose solves are off by a relative 1e-6 (what a wide explicit inverse does at kappa(L) ~ 1e5 x eps x
    growth) comes back to ~1e-12 after one step."""

    class Sloppy(OracleBackend):
        def tri_solve_(self, l, dinv_sb, sb, b):
            out = super().tri_solve_(l, dinv_sb, sb, b)
            g = torch.Generator().manual_seed(1)
            return out.mul_(1.0 + 1e-6 * torch.randn(out.shape, generator=g, dtype=out.dtype))

    prev = ops.set_backend(Sloppy())
    try:
        rng = np.random.default_rng(2)
        a = rng.standard_normal((200, 200))
        k = torch.as_tensor(a @ a.T + 200 * np.eye(200))
        b1, b40 = torch.as_tensor(rng.standard_normal((200, 1))), torch.as_tensor(rng.standard_normal((200, 40)))
        exact = matrix.Chol.factor_(k.clone())
        lo = torch.linalg.cholesky(k)
        for b in (b1, b40):
            want = torch.linalg.solve_triangular(lo, b, upper=False)
            exact.refine = False
            e0 = _rel(exact.solve(b), want)
            exact.refine = True
            e1 = _rel(exact.solve(b), want)
            e2 = _rel(exact.solve_(b.clone()), want)
            assert 1e-7 < e0 < 1e-4 and e1 < 1e-10 and e2 < 1e-10, (e0, e1, e2)
        # rows: zt ~ kxs L^{-T}
        kxs = torch.as_tensor(rng.standard_normal((24, 200)))
        want = torch.linalg.solve_triangular(lo, kxs.T.contiguous(), upper=False).T
        zt = (want * (1.0 + 1e-6 * torch.randn(want.shape, dtype=want.dtype, generator=torch.Generator().manual_seed(3)))).contiguous()
        e0 = _rel(zt, want)
        exact.refine_rows_(zt, kxs.clone())
        assert 1e-7 < e0 < 1e-4 and _rel(zt, want) < 1e-10
    finally:
        ops.set_backend(prev)


def test_oracle_error_recorded_in_the_golden_is_reproduced():
    """The fixture pins the fp64 oracle against the 80-bit numbers: the same oracle, here, lands on the recorded error (to a factor --
    BLAS builds differ in summation order), and the inputs regenerate bit for bit."""
    g = _golden(1536)
    x, y, xs = _inputs(1536)
    assert (_sha(x), _sha(y), _sha(xs)) == (g["x_sha256"], g["y_sha256"], g["xs_sha256"])
    terms = [("eq", 1.0, 1.0)]
    mean, _, var = O.gp_posterior(terms, x, g["noise"], y, xs, full_cov=False)
    e_m, e_v = _rel(mean, g["mean"]), _rel(var, g["var"])
    assert e_m <= 20 * g["oracle_fp64_error"]["mean"] + 1e-9 and e_v <= 20 * g["oracle_fp64_error"]["var"] + 1e-9, (e_m, e_v)
    assert abs(O.gp_logpdf(terms, x, g["noise"], y) - g["logpdf"]) / abs(g["logpdf"]) <= 1e-8


# ----------------------------------------------------------------------------------------------------------------------------------
# MI355X: the HIP path against the 80-bit reference
# ----------------------------------------------------------------------------------------------------------------------------------
OUT_DIR = os.path.join(ROOT, "gpurun_out")


def _note(key, **vals):
    try:
        os.makedirs(OUT_DIR, exist_ok=True)
        path = os.path.join(OUT_DIR, "r06_refinement_errors.json")
        rec = {}
        if os.path.exists(path):
            with open(path) as fh:
                rec = json.load(fh)
        rec[key] = vals
        with open(path, "w") as fh:
            json.dump(rec, fh, indent=1, sort_keys=True)
    except OSError:
        pass


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1536, 8192])
@pytest.mark.parametrize("order", ["logpdf-first", "posterior-first"])
def test_hip_path_against_the_80_bit_reference(hip_backend, n, order):
    """kappa ~ 5e8 (n = 1536, 256-wide explicit inverses) and ~1e10 (n = 8192: 1024-wide inverses, the regime of the conditioning
    sweep): with the refinement step the HIP path sits within a small multiple of the fp64 oracle's OWN error against the 80-bit
    numbers (both orders: the separate solves, and -- from 2048 observations -- the rows under the matrix); without it, the separate
    solves do not (asserted for the variance at n = 8192, so that the test notices when the step stops mattering)."""
    path = os.path.join(ROOT, "tests", "golden", f"illcond_n{n}.json")
    if not os.path.exists(path):
        pytest.skip(f"{path} has not been generated")
    g = _golden(n)
    x, y, xs = _inputs(n)
    assert (_sha(x), _sha(y), _sha(xs)) == (g["x_sha256"], g["y_sha256"], g["xs_sha256"])
    tx, ty, txs = (torch.as_tensor(a, device="cuda") for a in (x, y, xs))
    errs = {}
    for mode in (False, "auto"):
        old = matrix.config.refine_solves
        matrix.config.refine_solves = mode
        try:
            f = st.GP(st.EQ())
            fdd = f(tx, g["noise"])
            if order == "logpdf-first":
                lp = float(fdd.logpdf(ty))
                mean, var = (f | (fdd, ty))(txs).marginals()
            else:
                mean, var = (f | (fdd, ty))(txs).marginals()
                lp = float(fdd.logpdf(ty))
            chol = fdd.var.chol()
            assert chol.refine is (mode == "auto") and (chol.refined > 0) is (mode == "auto")
            if order == "posterior-first" and n >= 2048:
                assert chol.rows_under == 64
            errs[str(mode)] = {"logpdf": abs(lp - g["logpdf"]) / abs(g["logpdf"]), "mean": _rel(mean, g["mean"]), "var": _rel(var, g["var"])}
        finally:
            matrix.config.refine_solves = old
    _note(f"n{n}_{order}", oracle_fp64=g["oracle_fp64_error"], **errs)
    ref = g["oracle_fp64_error"]
    e = errs["auto"]
    assert e["logpdf"] <= 1e-8, errs
    # the fp64 oracle's own distance from the truth is one draw of a rounding-error walk: a small multiple of it, and north_star's 1e-6
    assert e["mean"] <= max(30 * ref["mean"], 2e-8) and e["mean"] <= 1e-6, (errs, ref)
    assert e["var"] <= max(100 * ref["var"], 2e-7) and e["var"] <= 1e-6, (errs, ref)
    if n == 8192 and order == "logpdf-first":      # (posterior-first at this order: pipelined panels, no wide inverse to repair)
        assert errs["False"]["var"] > 3 * e["var"], errs
