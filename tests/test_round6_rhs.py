"""Round 6: the observations as a right-hand side UNDER the matrix (``gpk_potrf_rows_rhs``, ``matrix.config.posterior_rows_rhs``).

When the posterior is what factorises the observations' kernel matrix, ``K(x*, x)`` rides through the factorisation as rows
(round 5); now ``y - m(x)`` rides along as one more row, so ``L^{-1} (y - m(x))`` -- which the posterior mean
(``stheno/model/observations.py:161-168``) and the log-density (``stheno/random.py:272-279``) both need -- comes out of the same
call instead of a 32-launch single-column sweep behind it.  Host logic over the test backend here; the HIP path against the
oracle and against the separate solve on the MI355X."""
import numpy as np
import pytest
import torch

import stheno_amd as st
from oracle import gp_oracle as O
from stheno_amd import matrix, ops

from .conftest import DEVICE


@pytest.fixture()
def rows_from_128():
    old = (matrix.config.posterior_rows_from, matrix.config.posterior_rows_min_points)
    matrix.config.posterior_rows_from, matrix.config.posterior_rows_min_points = 128, 8
    yield
    matrix.config.posterior_rows_from, matrix.config.posterior_rows_min_points = old


def _rel(a, ref):
    a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    ref = ref.detach().cpu().numpy() if torch.is_tensor(ref) else ref
    a, ref = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(ref, dtype=np.float64).reshape(-1)
    return float(np.max(np.abs(a - ref)) / np.max(np.abs(ref)))


def _case(n, ns, d, seed, dtype=np.float64):
    rng = np.random.default_rng(seed)
    x, xs = rng.standard_normal((n, d)), rng.standard_normal((ns, d))
    y = np.sin(x.sum(-1, keepdims=True)) + 0.1 * rng.standard_normal((n, 1))
    return x.astype(dtype), y.astype(dtype), xs.astype(dtype)


def test_the_observations_ride_along_and_serve_mean_and_logpdf(any_backend, rows_from_128):
    dev = DEVICE[0]
    x, y, xs = _case(384, 40, 2, 1)
    terms = [("eq", 1.0, 1.0)]
    ref_mean, _, ref_var = O.gp_posterior(terms, x, 0.05, y, xs, full_cov=False)
    ref_lp = O.gp_logpdf(terms, x, 0.05, y)
    tx, ty, txs = (torch.as_tensor(a, device=dev) for a in (x, y, xs))
    solves = []
    orig = matrix.Chol.solve

    def counting(self, b):
        solves.append(tuple(b.shape))
        return orig(self, b)

    matrix.Chol.solve = counting
    try:
        for on in (True, False):
            matrix.config.posterior_rows_rhs = on
            del solves[:]
            f = st.GP(st.EQ())
            fdd = f(tx, 0.05)
            mean, var = (f | (fdd, ty))(txs).marginals()
            lp = float(fdd.logpdf(ty))
            chol = fdd.var.chol()
            assert chol.rows_under == 40 and chol.rhs_rode is on
            assert _rel(mean, ref_mean) <= 1e-9 and _rel(var, ref_var) <= 1e-9 and abs(lp - ref_lp) <= 1e-9 * abs(ref_lp)
            if on and dev != "cpu":
                assert solves == []          # neither the mean nor the log-density solved for the observations again
            if not on:
                assert (384, 1) in solves
    finally:
        matrix.Chol.solve = orig
        matrix.config.posterior_rows_rhs = True


def test_a_mean_function_rides_along_too_and_a_later_solve_merges_its_own_inverses(oracle_backend, rows_from_128):
    x, y, xs = _case(300, 24, 1, 2)          # 300: padded to 384 inside
    tx, ty, txs = (torch.as_tensor(a) for a in (x, y, xs))
    f = st.GP(lambda t: 0.5 * t[:, :1], st.EQ())
    fdd = f(tx, 0.1)
    post = f | (fdd, ty)
    mean, var = post(txs).marginals()
    chol = fdd.var.chol()
    assert chol.rhs_rode and chol.rows_under == 24
    terms = [("eq", 1.0, 1.0)]
    ref_mean, _, ref_var = O.gp_posterior(terms, x, 0.1, y - 0.5 * x[:, :1], xs, full_cov=False)
    assert _rel(mean.reshape(-1) - 0.5 * txs[:, 0], ref_mean) <= 1e-9 and _rel(var, ref_var) <= 1e-9
    # a second set of points: the factor is there, the separate solve runs (and finds its block inverses)
    xs2 = torch.as_tensor(np.random.default_rng(3).standard_normal((16, 1)))
    mean2, var2 = post(xs2).marginals()
    ref_mean2, _, ref_var2 = O.gp_posterior(terms, x, 0.1, y - 0.5 * x[:, :1], xs2.numpy(), full_cov=False)
    assert _rel(mean2.reshape(-1) - 0.5 * xs2[:, 0], ref_mean2) <= 1e-9 and _rel(var2, ref_var2) <= 1e-9


def test_refined_factors_solve_for_the_observations_separately(oracle_backend, rows_from_128):
    x, y, xs = _case(256, 16, 1, 4)
    tx, ty, txs = (torch.as_tensor(a) for a in (x, y, xs))
    f = st.GP(st.EQ())
    fdd = f(tx, 1e-9)
    (f | (fdd, ty))(txs).marginals()
    chol = fdd.var.chol()
    assert chol.refine and not chol.rhs_rode and chol.rows_under == 16 and chol.refined >= 2


# ----------------------------------------------------------------------------------------------------------------------------------
# MI355X
# ----------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("n,ns,d,dtype,tol", [
    (2048, 64, 2, np.float64, 1e-9),          # pipelined panels: the right-hand side is a row like the others
    (4100, 300, 3, np.float64, 1e-9),         # padded order
    (11392, 200, 4, np.float64, 1e-9),        # look-ahead (1024-wide inverses) + plain tail, side stream
    (12288, 2048, 8, np.float64, 1e-9),
    (11520, 129, 4, np.float32, 2e-4),        # fp32: 512-wide inverses inside 1024-wide outer blocks
])
def test_hip_right_hand_side_under_the_matrix_equals_the_separate_solve(hip_backend, n, ns, d, dtype, tol):
    x, y, xs = _case(n, ns, d, n)
    x, y, xs = (a.astype(dtype) for a in (x / np.sqrt(d), y, xs / np.sqrt(d)))
    tx, ty, txs = (torch.as_tensor(a, device="cuda") for a in (x, y, xs))
    out = {}
    try:
        for on in (True, False):
            matrix.config.posterior_rows_rhs = on
            f = st.GP(st.EQ())
            fdd = f(tx, 0.1)
            mean, var = (f | (fdd, ty))(txs).marginals()
            lp = float(fdd.logpdf(ty))
            chol = fdd.var.chol()
            assert chol.rows_under == ns and chol.rhs_rode is on
            w = chol.solve_residual(ty.clone(), ty)          # what the factor remembers for y (on) / the separate sweep (off)
            out[on] = (mean.double().cpu().numpy(), var.double().cpu().numpy(), lp, w.double().cpu().numpy())
            # a later solve against the same factor (its merged inverses were skipped with the row): still right
            if on:
                b = torch.ones((n, 3), dtype=tx.dtype, device="cuda")
                lo = torch.tril(chol.l)
                r = lo @ chol.solve(b) - b
                assert float(r.abs().max()) <= (1e-9 if dtype == np.float64 else 2e-3)
    finally:
        matrix.config.posterior_rows_rhs = True
    assert _rel(out[True][3], out[False][3]) <= tol
    assert _rel(out[True][0], out[False][0]) <= tol and _rel(out[True][1], out[False][1]) <= tol
    assert abs(out[True][2] - out[False][2]) <= tol * abs(out[False][2])
    if n <= 4100:
        terms = [("eq", 1.0, 1.0)]
        ref_mean, _, ref_var = O.gp_posterior(terms, x.astype(np.float64), 0.1, y.astype(np.float64), xs.astype(np.float64), full_cov=False)
        assert _rel(out[True][0], ref_mean) <= 1e-8 and _rel(out[True][1], ref_var) <= 1e-8


# ----------------------------------------------------------------------------------------------------------------------------------
# batches: the log-density's residual solved along with the batched factorisation (``gpk_potrf_rhs``, ``matrix.config.logpdf_rhs``)
# ----------------------------------------------------------------------------------------------------------------------------------
def _batched_case(b, n, d, seed, dtype):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((b, n, d)) / np.sqrt(d)
    y = np.sin(x.sum(-1, keepdims=True)) + 0.3 * rng.standard_normal((b, n, 1))
    return x.astype(dtype), y.astype(dtype)


def test_batched_logpdf_hands_its_residual_to_the_factorisation(any_backend):
    dev = DEVICE[0]
    x, y = _batched_case(5, 96, 2, 7, np.float64)
    terms = [("eq", 1.0, 1.0)]
    ref = np.array([O.gp_logpdf(terms, x[i], 0.2, y[i]) for i in range(5)])
    tx, ty = torch.as_tensor(x, device=dev), torch.as_tensor(y, device=dev)
    try:
        for on in (True, False):
            matrix.config.logpdf_rhs = on
            fdd = st.GP(st.EQ())(tx, 0.2)
            lp = fdd.logpdf(ty)
            chol = fdd.var.chol()
            assert chol.rhs_rode is on
            assert _rel(lp, ref) <= 1e-10
            # a mean function: the residual is not the data, it rides all the same
            fdd2 = st.GP(lambda t: 0.25 * t[..., :1], st.EQ())(tx, 0.2)
            lp2 = fdd2.logpdf(ty)
            assert fdd2.var.chol().rhs_rode is on
            ref2 = np.array([O.gp_logpdf(terms, x[i], 0.2, y[i] - 0.25 * x[i][:, :1]) for i in range(5)])
            assert _rel(lp2, ref2) <= 1e-10
    finally:
        matrix.config.logpdf_rhs = True


@pytest.mark.gpu
@pytest.mark.parametrize("b,n,dtype,tol", [
    (64, 1024, np.float32, 2e-5),         # mixed-phase steps: the sweep's steps on the side stream
    (200, 2048, np.float32, 2e-5),
    (70, 512, np.float64, 1e-11),         # fp64 batches keep the lockstep launches: factorise, then sweep (inside the native call)
    (8, 640, np.float32, 2e-5),
])
def test_hip_batched_logpdf_with_the_residual_under_way_equals_the_separate_sweep(hip_backend, b, n, dtype, tol):
    x, y = _batched_case(b, n, 3, b + n, dtype)
    tx, ty = torch.as_tensor(x, device="cuda"), torch.as_tensor(y, device="cuda")
    got = {}
    try:
        for on in (True, False):
            matrix.config.logpdf_rhs = on
            fdd = st.GP(st.EQ())(tx, 0.2)
            lp = fdd.logpdf(ty)
            chol = fdd.var.chol()
            assert chol.rhs_rode is on
            got[on] = (lp.double().cpu().numpy(), chol.l.clone())
    finally:
        matrix.config.logpdf_rhs = True
    assert torch.equal(torch.tril(got[True][1]), torch.tril(got[False][1]))          # the same factors, bit for bit
    assert _rel(got[True][0], got[False][0]) <= tol
    terms = [("eq", 1.0, 1.0)]
    ref = np.array([O.gp_logpdf(terms, x[i].astype(np.float64), 0.2, y[i].astype(np.float64)) for i in (0, b // 2, b - 1)])
    assert _rel(got[True][0][[0, b // 2, b - 1]], ref) <= (1e-3 if dtype == np.float32 else 1e-9)


# ----------------------------------------------------------------------------------------------------------------------------------
# one large matrix, log-density FIRST: the residual is the only thing under the matrix (its strip), the posterior's solve follows
# ----------------------------------------------------------------------------------------------------------------------------------
def test_logpdf_first_hands_its_residual_to_the_factorisation_of_one_matrix(any_backend, rows_from_128):
    dev = DEVICE[0]
    x, y, xs = _case(300, 20, 2, 11)
    terms = [("eq", 1.0, 1.0)]
    ref_lp = O.gp_logpdf(terms, x, 0.05, y)
    ref_mean, _, ref_var = O.gp_posterior(terms, x, 0.05, y, xs, full_cov=False)
    tx, ty, txs = (torch.as_tensor(a, device=dev) for a in (x, y, xs))
    try:
        for on in (True, False):
            matrix.config.logpdf_rhs = on
            f = st.GP(st.EQ())
            fdd = f(tx, 0.05)
            lp = float(fdd.logpdf(ty))
            chol = fdd.var.chol()
            assert chol.rhs_rode is on and chol.rows_under == 0
            mean, var = (f | (fdd, ty))(txs).marginals()
            assert abs(lp - ref_lp) <= 1e-9 * abs(ref_lp) and _rel(mean, ref_mean) <= 1e-9 and _rel(var, ref_var) <= 1e-9
    finally:
        matrix.config.logpdf_rhs = True


@pytest.mark.gpu
@pytest.mark.parametrize("n,ns,d,dtype,tol", [
    (2048, 70, 2, np.float64, 1e-9),
    (4100, 64, 3, np.float64, 1e-9),          # padded order
    (12288, 2048, 8, np.float64, 1e-9),       # look-ahead: the posterior's 2048-column solve finds the merged inverses it needs
    (11520, 300, 4, np.float32, 2e-4),
])
def test_hip_logpdf_first_with_the_residual_under_the_matrix_equals_the_separate_sweep(hip_backend, n, ns, d, dtype, tol):
    x, y, xs = _case(n, ns, d, n + 1)
    x, y, xs = (a.astype(dtype) for a in (x / np.sqrt(d), y, xs / np.sqrt(d)))
    tx, ty, txs = (torch.as_tensor(a, device="cuda") for a in (x, y, xs))
    out = {}
    try:
        for on in (True, False):
            matrix.config.logpdf_rhs = on
            f = st.GP(st.EQ())
            fdd = f(tx, 0.1)
            lp = float(fdd.logpdf(ty))
            chol = fdd.var.chol()
            assert chol.rhs_rode is on and chol.rows_under == 0
            merged_before = set(chol._dinv_sb)
            mean, var = (f | (fdd, ty))(txs).marginals()
            if n >= matrix.config.potrf_lookahead_from and dtype == np.float64:
                assert set(chol._dinv_sb) == merged_before          # the look-ahead's inverses served the solve: nothing merged again
            out[on] = (mean.double().cpu().numpy(), var.double().cpu().numpy(), lp)
    finally:
        matrix.config.logpdf_rhs = True
    assert _rel(out[True][0], out[False][0]) <= tol and _rel(out[True][1], out[False][1]) <= tol
    assert abs(out[True][2] - out[False][2]) <= tol * abs(out[False][2])


@pytest.mark.gpu
def test_right_hand_side_under_the_matrix_inside_a_stream_capture(hip_backend):
    """``gpk_potrf_rows_rhs`` forks onto the helper stream AND onto its sibling (the right-hand side's products): under stream capture
    both join the capture (``include/gpk.h``).  Capture one look-ahead factorisation with rows and a right-hand side, replay it on
    fresh data, compare with the eager call."""
    n, ns = 4096, 192
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 4, generator=g, dtype=torch.float64).to("cuda")
    xs = torch.randn(ns, 4, generator=g, dtype=torch.float64).to("cuda")
    y = torch.randn(n, generator=g, dtype=torch.float64).to("cuda")
    a0 = torch.empty((n + ns + 64, n), dtype=torch.float64, device="cuda")
    a0[:n] = st.EQ().pairwise(x, None)
    a0[:n].diagonal().add_(0.1)
    a0[n:n + ns] = st.EQ().pairwise(xs, x)
    a0[n + ns:] = 0
    a0[n + ns] = y
    be = ops.get_backend()
    eager = a0.clone()
    be.potrf_rows_(eager, lookahead_nb=512, rhs_row=True, tail_inverses=False)      # (also creates the helper streams outside the capture)
    torch.cuda.synchronize()
    lo = torch.tril(eager[:n])
    want_w = torch.linalg.solve_triangular(lo, y[:, None], upper=False)[:, 0]
    assert _rel(eager[n + ns], want_w) < 1e-10
    buf = a0.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            outs = be.potrf_rows_(buf, lookahead_nb=512, rhs_row=True, tail_inverses=False)
    torch.cuda.current_stream().wait_stream(side)
    buf.copy_(a0)
    outs[1].zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert int(outs[1].max()) == 0
    assert torch.equal(torch.tril(buf[:n]), torch.tril(eager[:n]))
    assert _rel(buf[n:n + ns], eager[n:n + ns]) < 1e-13 and _rel(buf[n + ns], eager[n + ns]) < 1e-13


# ----------------------------------------------------------------------------------------------------------------------------------
# determinism / concurrency of the right-hand sides that ride along
# ----------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_batched_right_hand_sides_on_four_concurrent_streams_give_the_single_stream_bits(hip_backend):
    """``gpk_potrf_rhs`` inside the mixed-phase steps: sub-batches of 64 fp32 matrices on four streams at once, 60 rounds -- factors AND
    solved right-hand sides equal the single-stream run bit for bit (each entry of b is written by exactly one solve tile per step, in
    step order: nothing about the result depends on what runs beside)."""
    be = ops.get_backend()
    terms = ops.KTerms([("eq", 1.0, 1.0)])
    g = torch.Generator().manual_seed(1)
    parts, per, n = 4, 64, 1024
    x = torch.randn(parts * per, n, 3, generator=g, dtype=torch.float64).float().cuda()
    b = torch.randn(parts * per, n, generator=g, dtype=torch.float64).float().cuda()

    def factor(i):
        a = be.kmat(terms, x[i * per:(i + 1) * per], lower=True, diag_add=0.1 + 1e-6)
        rhs = b[i * per:(i + 1) * per].clone()
        dinv, info = be.potrf_(a, rhs=rhs)
        return torch.tril(a), rhs, info

    ref = [factor(i) for i in range(parts)]
    torch.cuda.synchronize()
    # the solved vectors against the separate sweep on the same factors
    a0 = be.kmat(terms, x[:per], lower=True, diag_add=0.1 + 1e-6)
    dinv0, _ = be.potrf_(a0)
    want = be.tri_solve_(a0, dinv0, 128, b[:per].clone().unsqueeze(-1))[..., 0]
    assert _rel(ref[0][1], want) < 2e-5
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for it in range(60):
        cur = torch.cuda.current_stream()
        outs = [None] * parts
        for i, s in enumerate(streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                outs[i] = factor(i)
        for s in streams:
            cur.wait_stream(s)
        torch.cuda.synchronize()
        for i in range(parts):
            for got, wanted, what in zip(outs[i], ref[i], ("factor", "solved right-hand side", "info")):
                assert torch.equal(got, wanted), f"round {it}, sub-batch {i}: {what} differs from the single-stream run"


@pytest.mark.gpu
def test_look_ahead_with_rows_and_a_right_hand_side_is_reproducible_and_survives_company(hip_backend):
    """``gpk_potrf_rows_rhs`` shares ONE helper stream and ONE side stream per device between all callers: the same posterior evaluated
    ten times in a row, and then from two host threads on two streams at once, gives the same bits every time."""
    import threading

    n, ns = 11392, 256
    x, y, xs = _case(n, ns, 4, 99)
    tx, ty, txs = (torch.as_tensor(a / (2.0 if i != 1 else 1.0), device="cuda") for i, a in enumerate((x, y, xs)))

    def evaluate():
        f = st.GP(st.EQ())
        fdd = f(tx, 0.1)
        mean, var = (f | (fdd, ty))(txs).marginals()
        lp = fdd.logpdf(ty)
        assert fdd.var.chol().rhs_rode
        return mean.clone(), var.clone(), lp.clone()

    ref = evaluate()
    torch.cuda.synchronize()
    for _ in range(10):
        got = evaluate()
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(got, ref))
    results, errors = {}, []

    def worker(k):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for _ in range(4):
                    results[k] = evaluate()
                s.synchronize()
        except Exception as e:        # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    torch.cuda.synchronize()
    for k in range(2):
        assert all(torch.equal(a, b) for a, b in zip(results[k], ref)), f"thread {k}"
