"""Host-side model logic (GP / Measure / FDD / Normal / Obs / PseudoObs) on the CPU box.

Every test runs TWICE: on the CPU box (``-m "not gpu"``) over the TEST-ONLY ``OracleBackend`` from
conftest.py, pinning what the Python layer composes -- lazy resolution, caching, shapes, errors, the
formulas above the kernels -- and on the MI355X (``-m gpu``) over ``libgpk.so`` with every tensor on the
device, mirroring the reference's own tests (cited per test) on the product path.
"""
import time

import numpy as np
import pytest
import torch
from scipy.stats import multivariate_normal

import stheno_amd as st
from oracle import gp_oracle as O
from stheno_amd import B
from stheno_amd.matrix import Dense, Diagonal, KernelDense, Zero

from stheno_amd import ops

from .conftest import OracleBackend, golden

f64 = torch.float64
_DEV = ["cpu"]


@pytest.fixture(autouse=True, params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def backend(request):
    """The op backend (and the device new tensors are created on) for one test."""
    if request.param == "oracle":
        prev = ops.set_backend(OracleBackend())
        _DEV[0] = "cpu"
        yield request.param
        ops.set_backend(prev)
    else:
        prev = ops.set_backend(None)
        assert ops.get_backend().name == "hip"
        _DEV[0] = "cuda"
        torch.set_default_device("cuda")
        try:
            yield request.param
        finally:
            torch.set_default_device(None)      # (NOT "cpu": that leaves a device mode behind under which torch.as_tensor(cuda_tensor) copies to the host)
            _DEV[0] = "cpu"
            ops.set_backend(prev)


def t(a):
    return torch.as_tensor(np.asarray(a), dtype=f64, device=_DEV[0])


def gen(seed):
    return torch.Generator(device=_DEV[0]).manual_seed(seed)


def approx(a, b, atol=1e-8, rtol=1e-8):
    a = B.to_numpy(a) if torch.is_tensor(a) or hasattr(a, "dense") else np.asarray(a)
    b = B.to_numpy(b) if torch.is_tensor(b) or hasattr(b, "dense") else np.asarray(b)
    np.testing.assert_allclose(a, b, atol=atol, rtol=rtol)


# ---------------------------------------------------------------- Normal (tests/test_random.py)
def test_normal_lazy_zero_mean():          # test_random.py:68-83
    dist = st.Normal(lambda: torch.eye(3, dtype=f64))
    assert dist.mean_is_zero
    assert dist.computed("mean") and not dist.computed("var")
    approx(dist.mean, np.zeros((3, 1)))
    assert dist.computed("var")
    approx(dist.var, np.eye(3))


def test_normal_lazy_nonzero_mean():       # test_random.py:86-96
    dist = st.Normal(lambda: torch.ones(3, 1, dtype=f64), lambda: torch.eye(3, dtype=f64))
    assert not dist.computed("mean") and not dist.computed("var")
    approx(dist.mean, np.ones((3, 1)))
    assert not dist.computed("var")
    approx(dist.var, np.eye(3))


def test_normal_lazy_var_diag():           # test_random.py:99-108
    dist = st.Normal(lambda: torch.eye(3, dtype=f64))
    approx(dist.var_diag, np.ones(3))
    assert dist.computed("var")
    dist = st.Normal(lambda: torch.eye(3, dtype=f64), var_diag=lambda: 9)
    assert dist.var_diag == 9 and not dist.computed("var")


def test_normal_lazy_mean_var_called_only_when_both_missing():    # test_random.py:111-133
    calls = []

    def mv():
        calls.append(1)
        return torch.full((3, 1), 8.0, dtype=f64), 9 * torch.eye(3, dtype=f64)

    dist = st.Normal(lambda: torch.ones(3, 1, dtype=f64), lambda: torch.eye(3, dtype=f64), mean_var=mv)
    m, v = dist.mean_var
    approx(m, 8 * np.ones((3, 1))); approx(v, 9 * np.eye(3)); assert len(calls) == 1
    dist = st.Normal(lambda: torch.ones(3, 1, dtype=f64), lambda: torch.eye(3, dtype=f64), mean_var=mv)
    approx(dist.mean, np.ones((3, 1)))
    m, v = dist.mean_var
    approx(m, np.ones((3, 1))); approx(v, np.eye(3)); assert len(calls) == 1


def test_normal_lazy_mean_var_diag():      # test_random.py:136-158
    mvd = lambda: (torch.full((3, 1), 8.0, dtype=f64), torch.full((3,), 9.0, dtype=f64))  # noqa: E731
    dist = st.Normal(lambda: torch.ones(3, 1, dtype=f64), lambda: torch.eye(3, dtype=f64), mean_var_diag=mvd)
    m, v = dist.marginals()
    approx(m, 8 * np.ones(3)); approx(v, 9 * np.ones(3))
    dist = st.Normal(lambda: torch.ones(3, 1, dtype=f64), lambda: torch.eye(3, dtype=f64), mean_var_diag=mvd)
    approx(dist.var_diag, np.ones(3))
    m, v = dist.marginals()
    approx(m, np.ones(3)); approx(v, np.ones(3))


@pytest.fixture()
def normal1():
    g = gen(0)
    mean = torch.randn(3, 1, dtype=f64, generator=g)
    chol = torch.randn(3, 3, dtype=f64, generator=g)
    return st.Normal(mean, chol @ chol.T)


def test_normal_marginals_and_bounds(normal1):     # test_random.py:165-175
    mean, var = normal1.marginals()
    approx(mean, normal1.mean[:, 0]); approx(var, torch.diagonal(B.dense(normal1.var)))
    m, lo, hi = normal1.marginal_credible_bounds()
    approx(lo, normal1.mean[:, 0] - 1.96 * torch.diagonal(B.dense(normal1.var)) ** 0.5)
    approx(hi, normal1.mean[:, 0] + 1.96 * torch.diagonal(B.dense(normal1.var)) ** 0.5)


def test_normal_logpdf_vs_scipy(normal1):          # test_random.py:185-192
    B.epsilon = 0.0
    try:
        sp = multivariate_normal(B.to_numpy(normal1.mean)[:, 0], B.to_numpy(normal1.var))
        x = torch.randn(3, 10, dtype=f64, generator=gen(1))
        approx(normal1.logpdf(x), sp.logpdf(B.to_numpy(x).T), rtol=1e-6)
        assert normal1.logpdf(torch.ones(3, 1, dtype=f64)).shape == ()
        assert normal1.logpdf(torch.ones(3, 2, dtype=f64)).shape == (2,)
        approx(normal1.entropy(), sp.entropy(), rtol=1e-8)
    finally:
        B.epsilon = 1e-12


def test_normal_logpdf_missing_data(normal1):      # test_random.py:195-204
    x = torch.randn(3, 1, dtype=f64)
    x[1] = float("nan")
    sub = st.Normal(normal1.mean[[0, 2]], B.dense(normal1.var)[[0, 2]][:, [0, 2]])
    approx(normal1.logpdf(x), sub.logpdf(x[[0, 2]]))


# ---------------------------------------------------------------- FDD (tests/model/test_fdd.py)
def test_fdd_noise_typing_and_var():       # test_fdd.py:15-82, test_gp.py:57-92
    p = st.GP(st.EQ())
    x = t(np.linspace(0, 5, 5))
    assert isinstance(p(x).noise, Zero)
    assert isinstance(p(x, 0.1).noise, Diagonal)
    approx(p(x, 0.1).noise.diag(), 0.1 * np.ones(5))
    assert isinstance(p(x, t(np.full(5, 0.2))).noise, Diagonal)
    assert isinstance(p(x, 0.3 * torch.eye(5, dtype=f64)).noise, Dense)
    d = p(x, 1.0)
    assert not d.computed("var") and not d.computed("mean")            # nothing computed at construction
    approx(d.var, O.kernel_matrix([("eq", 1, 1)], np.linspace(0, 5, 5)) + np.eye(5))
    approx(d.mean, np.zeros((5, 1)))
    assert isinstance(d.var, KernelDense)
    assert p(x).dtype == f64


def test_fdd_properties_under_posterior():  # test_fdd.py:111-134
    p = st.GP(1, st.EQ())
    x = t(np.linspace(0, 5, 5))
    y = p(x, 0.1).sample()
    post = p | (p(x, 0.1), y)
    xs = t(np.linspace(0, 5, 10))
    fdd = post(xs, 0.2)
    mean, var = fdd.mean, B.dense(fdd.var)
    approx(post(xs, 0.2).var_diag, torch.diagonal(var))
    m2, v2 = post(xs, 0.2).mean_var
    approx(m2, mean); approx(v2, var)
    m3, vd3 = post(xs, 0.2).marginals()
    approx(m3, mean[:, 0]); approx(vd3, torch.diagonal(var))


def test_marginals_do_not_form_the_covariance():   # test_gp.py:201-211
    p = st.GP(st.EQ())
    x = t(np.linspace(0, 5, 5))
    y = p(x, 0.1).sample()
    p = p | (p(x, 0.1), y)
    xs = t(np.linspace(0, 5, 10_000))
    p(xs, 0.2).marginal_credible_bounds()       # warm (lazy imports inside torch / scipy)
    start = time.time()
    p(xs, 0.2).marginal_credible_bounds()
    assert time.time() - start < 1


# ---------------------------------------------------------------- conditioning (tests/model/test_model.py)
def test_conditioning_spellings_agree():   # test_model.py:123-178
    m = st.Measure()
    p = st.GP(1, st.EQ(), measure=m)
    x = t(np.linspace(0, 2, 3))
    y = p(x, 0.1).sample()
    xs = t(np.linspace(0, 5, 5))
    posts = [m.condition(p(x, 0.1), y), m.condition((p(x, 0.1), y)), m | (p(x, 0.1), y), m | ((p(x, 0.1), y),),
             m | st.Obs(p(x, 0.1), y), m | st.Obs((p(x, 0.1), y))]
    ref_m, ref_v = posts[0](p)(xs).mean, B.dense(posts[0](p)(xs).var)
    for post in posts[1:]:
        approx(post(p)(xs).mean, ref_m); approx(post(p)(xs).var, ref_v)
    post = m | (p(x, 0.1), y)
    assert isinstance(post(p(x, 0.1)), st.FDD)
    approx(post(p(x, 0.1)).var, post(p)(x, 0.1).var)
    # agreement with the oracle
    om, ov, _ = O.gp_posterior([("eq", 1, 1)], B.to_numpy(x), 0.1, B.to_numpy(y) - 1.0, B.to_numpy(xs))
    approx(ref_m[:, 0], om + 1.0); approx(ref_v, ov)


def test_conditioning_interpolates_and_chains():   # test_model.py:211-228
    p = st.GP(1, st.EQ())
    x = t(np.linspace(0, 5, 10))
    y = p(x).sample()
    approx(p.condition(p(x), y).mean(x), y, atol=1e-5)
    p1 = p | (p(x), y)
    x2 = t(np.linspace(10, 20, 10))
    y2 = p1(x2).sample()
    p2 = p1 | (p1(x2), y2)
    approx(p2.mean(x2), y2, atol=1e-5)
    approx(p2.mean(x), y, atol=1e-5)


@pytest.mark.parametrize("shape", [(0,), (0, 1)])
def test_conditioning_on_nothing_returns_prior_objects(shape):   # test_model.py:198-208
    p = st.GP(1, st.EQ())
    x = torch.zeros(*shape, dtype=f64)
    post = p | (p(x), torch.zeros(0, 1, dtype=f64))
    assert post.mean is p.mean and post.kernel is p.kernel


def test_conditioning_missing_data_and_shape_check():   # test_model.py:231-246
    p = st.GP(1, st.EQ())
    x = t(np.linspace(0, 5, 10))
    y = p(x, 0.1).sample()
    y_nan = y.clone(); y_nan[:3] = float("nan")
    xs = t(np.linspace(0, 5, 7))
    a, b = (p | (p(x, 0.1), y_nan))(xs), (p | (p(x[3:], 0.1), y[3:]))(xs)
    approx(a.mean, b.mean); approx(a.var, b.var)
    f = st.GP(1, st.EQ())
    xx = torch.randn(2, dtype=f64)
    f | (f(xx), torch.randn(2, 1, dtype=f64))
    with pytest.raises(ValueError):
        f | (f(xx), torch.randn(2, 2, dtype=f64))


def test_chain_rule_of_logpdfs():          # test_model.py:391-398 (single-process form)
    m = st.Measure()
    p = st.GP(st.EQ() + 2 * st.Exp(), measure=m)
    x1, x2 = t(np.linspace(0, 2, 5)), t(np.linspace(1.1, 3, 6))
    y = p(torch.cat([x1, x2]), 0.2).sample()
    y1, y2 = y[:5], y[5:]
    d2 = m | (p(x1, 0.2), y1)
    approx(p(x1, 0.2).logpdf(y1) + d2(p)(x2, 0.2).logpdf(y2), p(torch.cat([x1, x2]), 0.2).logpdf(y))
    approx(m.logpdf(p(x1, 0.2), y1), p(x1, 0.2).logpdf(y1))
    approx(m.logpdf(st.Obs(p(x1, 0.2), y1)), p(x1, 0.2).logpdf(y1))


# ---------------------------------------------------------------- pseudo-points (test_model.py:249-332)
@pytest.mark.parametrize("cls", [st.PseudoObs, st.PseudoObsFITC, st.PseudoObsDTC])
@pytest.mark.parametrize("noise_form", ["scalar", "vector", "Diagonal"])
def test_pseudoobs_exact_when_inducing_equals_inputs(cls, noise_form):
    m = st.Measure()
    p = st.GP(1, st.EQ() + 2 * st.Exp(), measure=m)
    x = t(np.linspace(3, 5, 6))
    nz = {"scalar": 0.3, "vector": t(np.linspace(0.2, 0.5, 6)), "Diagonal": Diagonal(t(np.linspace(0.2, 0.5, 6)))}[noise_form]
    y = p(x, nz).sample()
    xs = t(np.linspace(0, 5, 5))
    exact, approx_post = m | (p(x, nz), y), m | cls(p(x), p(x, nz), y)
    approx(approx_post(p)(xs).mean, exact(p)(xs).mean, atol=1e-7)
    approx(approx_post(p)(xs).var, exact(p)(xs).var, atol=1e-7)
    m1, v1 = approx_post(p)(xs).marginals()
    approx(v1, torch.diagonal(B.dense(exact(p)(xs).var)), atol=1e-7)
    approx(cls(p(x), p(x, nz), y).elbo(m), m.logpdf(st.Obs(p(x, nz), y)))
    approx(m.logpdf(cls(p(x), p(x, nz), y)), p(x, nz).logpdf(y))
    obs = cls(p(x), p(x, nz), y)
    for name in ["K_z", "elbo", "mu", "A"]:
        assert getattr(obs, name)(m) is getattr(obs, name)(m)
    with pytest.raises(RuntimeError):
        cls(p(x), (p(x, B.dense(p(x).var)), y)).elbo(m)


@pytest.mark.parametrize("name", ["sparse_eq_n400_m50_d2", "sparse_matern32_linear_n300_m40_d3", "sparse_matern52_n350_m45_d2"])
def test_sparse_golden_through_the_api(name):
    g = golden(name + ".npz")
    kinds = {"eq": st.EQ, "matern12": st.Matern12, "matern32": st.Matern32, "matern52": st.Matern52, "linear": st.Linear}
    B.epsilon = float(g["epsilon"])
    try:
        m = st.Measure()
        f = st.GP(sum(float(v) * kinds[str(kd)]().stretch(float(sc)) for kd, v, sc in zip(g["kinds"], g["variances"], g["scales"])), measure=m)
        x, z, xs, y = (t(g[k]) for k in ("x", "z", "xs", "y"))
        for cls, tag in [(st.PseudoObs, "vfe"), (st.PseudoObsFITC, "fitc"), (st.PseudoObsDTC, "dtc")]:
            obs = cls(f(z), f(x, float(g["noise"])), y)
            approx(obs.elbo(m), g[f"elbo_{tag}"][0], rtol=1e-9)
            approx(obs.mu(m), g[f"mu_{tag}"], atol=1e-7)
            approx(obs.A(m), g[f"A_{tag}"], rtol=1e-7, atol=1e-7)
            mean, vd = (m | obs)(f)(xs).marginals()
            approx(mean, g[f"post_mean_{tag}"], atol=1e-7)
            approx(vd, g[f"post_var_diag_{tag}"], atol=1e-7)
        assert st.SparseObs is st.PseudoObs and st.SparseObservations is st.PseudoObs
    finally:
        B.epsilon = 1e-12


# ---------------------------------------------------------------- golden / batched / misc
@pytest.mark.parametrize("name", ["dense_eq_n256_d8", "dense_eq_n300_d1_c3", "dense_matern12_n200_d3",
                                  "dense_matern32_n200_d3", "dense_matern52_n333_d5", "dense_eq_linear_n512_d4"])
def test_dense_golden_through_the_api(name, backend):
    g = golden(name + ".npz")
    # the fixtures follow upstream's |a|^2 + |b|^2 - 2ab distances (as the oracle backend does); the HIP kernel takes direct
    # differences: <= 3e-8 relative on Matern diagonals -- the device run is held to the 1e-6 bar
    loose = dict(atol=1e-6, rtol=1e-6) if backend == "hip" else {}
    kinds = {"eq": st.EQ, "matern12": st.Matern12, "matern32": st.Matern32, "matern52": st.Matern52, "linear": st.Linear}
    k = sum(float(v) * kinds[str(kd)]().stretch(float(s)) for kd, v, s in zip(g["kinds"], g["variances"], g["scales"]))
    f = st.GP(k)
    x, xs, y = t(g["x"]), t(g["xs"]), t(g["y"])
    approx(np.atleast_1d(B.to_numpy(f(x, float(g["noise"])).logpdf(y))), g["logpdf"], rtol=1e-6 if backend == "hip" else 1e-10)
    post = f | (f(x, float(g["noise"])), y[:, :1])
    mean, vd = post(xs).marginals()
    approx(mean, g["post_mean"], **(loose or dict(atol=1e-9))); approx(vd, np.maximum(g["post_var_diag"], 0), **(loose or dict(atol=1e-9)))
    approx(post(xs).var, g["post_var"], **(loose or dict(atol=1e-9)))
    approx(f.kernel.elwise(x)[:, 0], g["kdiag"], **loose); approx(f.kernel(x[:8], xs[:8]), g["k_corner"], **loose)


def test_batched_shapes_and_values():      # tests/model/test_cases.py:134-155, README.md:744-766
    g = golden("batched_eq_b16_n100_d3.npz")
    p = st.GP(2 * st.EQ().stretch(0.5))
    x, y = t(g["x"]), t(g["y"])
    lp = p(x, 0.1).logpdf(y)
    assert lp.shape == (16,)
    approx(lp, g["logpdf"], rtol=1e-10)
    ys = p(x, 0.1).sample()
    assert ys.shape == (16, 100, 1)
    post = p | (p(x, 0.1), y)
    assert torch.all(post(x, 0.1).logpdf(y) > lp)


def test_measures_names_and_defaults():    # test_model.py:62-121
    m = st.Measure()
    p1, p2 = st.GP(st.EQ(), measure=m), st.GP(st.EQ(), name="two", measure=m)
    p1.name = "one"
    assert m["one"] is p1 and m["two"] is p2 and p1.name == "one" and m[p2] == "two"
    with pytest.raises(RuntimeError):
        p1.name = "two"
    with st.Measure() as prior:
        q = st.GP(st.EQ())
        assert q.measure is prior
    assert st.Measure.default is None
    assert m.kernels[p1, p2] == st.ZeroKernel()
    with pytest.raises(AssertionError):
        p1 + q
    s = p1 + p2
    approx(s(t([0.0, 1.0])).var, 2 * O.kernel_matrix([("eq", 1, 1)], np.array([0.0, 1.0])))
    approx((2 * p1)(t([0.0, 1.0])).var, 4 * O.kernel_matrix([("eq", 1, 1)], np.array([0.0, 1.0])))


def test_non_positive_definite_raises():
    with pytest.raises(torch.linalg.LinAlgError):
        st.Normal(torch.tensor([[1.0, 2.0], [2.0, 1.0]], dtype=f64)).logpdf(torch.zeros(2, 1, dtype=f64))


def test_epsilon_is_read_at_factorisation_time(backend):    # README.md:820-831
    x = t(np.linspace(0, 2, 10))
    f = st.GP(st.EQ())
    y = x**2
    mean = (f | (f(x), y))(t([1.0, 2.0, 3.0])).mean
    # kappa(K) ~ 1e12 (noise-free conditioning): LAPACK builds agree to ~1e-7 here; the device is
    # held to the 1e-6 bar
    approx(mean[:, 0], [1.00000068, 3.99999999, 8.4825932], rtol=1e-6 if backend == "hip" else 2e-7)
    B.epsilon = 1e-8
    try:
        mean8 = (f | (f(x), y))(t([1.0, 2.0, 3.0])).mean
    finally:
        B.epsilon = 1e-12
    assert abs(float(mean8[2, 0]) - 8.4825932) > 1e-3


def test_big_buffers_are_released_without_the_cyclic_gc():
    """Kernel matrices / Cholesky factors are multi-GB on the device: the object graph
    (GP <-> Measure, FDD constructors, posterior back-references) must not contain reference
    cycles that keep them alive until Python's cyclic garbage collector happens to run."""
    import gc
    import weakref

    x, y, xs = t(np.linspace(0, 5, 20)), t(np.random.randn(20, 1)), t(np.linspace(0, 5, 7))
    gc.collect()
    gc.disable()
    try:
        def once():
            f = st.GP(st.EQ())
            fdd = f(x, 0.1)
            fdd.logpdf(y)
            post = f | (fdd, y)
            post(xs).marginals()
            return [weakref.ref(o) for o in (fdd.var, fdd.var.chol(), f, post, f.measure)]

        assert all(r() is None for r in once())
        # a long-lived prior GP must not accumulate its posteriors
        f = st.GP(st.EQ())
        refs = []
        for _ in range(3):
            fdd = f(x, 0.1)
            post = f | (fdd, y)
            post(xs).marginals()
            refs.append(weakref.ref(fdd.var))
        del fdd, post
        assert all(r() is None for r in refs) and len(f._measures) == 1
    finally:
        gc.enable()
    # derived processes keep what their rules refer to alive
    m = st.Measure()
    p_sum = st.GP(st.EQ(), measure=m) + st.GP(2 * st.EQ(), measure=m)     # parents only referenced by p_sum
    approx(p_sum(t([0.0, 1.0])).var, 3 * O.kernel_matrix([("eq", 1, 1)], np.array([0.0, 1.0])))
    q = (m | (p_sum(x, 0.1), y))(p_sum)                                   # prior handle dropped, posterior used
    assert torch.isfinite(q(xs).mean).all()


# ---------------------------------------------------------------- GP bookkeeping (tests/model/test_gp.py)
def test_gp_construction_and_resolution():    # test_gp.py:55-92
    from stheno_amd import kernels as K

    x = t(np.random.default_rng(0).standard_normal((10, 1)))
    k = st.EQ()
    m = lambda u: u ** 2
    assert isinstance(st.GP(k).mean, K.ZeroMean)
    assert isinstance(st.GP(5, k).mean, K.ScaledMean)
    assert isinstance(st.GP(0, k).mean, K.ZeroMean)
    assert isinstance(st.GP(1, k).mean, K.OneMean) and str(st.GP(1, k)) == "GP(1, EQ())"
    assert isinstance(st.GP(m, k).mean, K.FunctionMean)
    assert isinstance(st.GP(k).kernel, st.EQ)
    assert isinstance(st.GP(5).kernel, K.Scaled)
    assert isinstance(st.GP(0).kernel, st.ZeroKernel)
    p = st.GP(m, k)
    approx(p.kernel(x, x), O.kernel_matrix([("eq", 1.0, 1.0)], B.to_numpy(x)))
    approx(p.kernel.elwise(x), np.ones((10, 1)))
    d = p(x)                                   # without noise
    approx(B.dense(d.var), O.kernel_matrix([("eq", 1.0, 1.0)], B.to_numpy(x)))
    approx(d.mean, B.to_numpy(x) ** 2)
    d = p(x, 1)                                # with noise
    approx(B.dense(d.var), O.kernel_matrix([("eq", 1.0, 1.0)], B.to_numpy(x)) + np.eye(10))
    approx(d.mean, B.to_numpy(x) ** 2)


def test_gp_sum_and_mul_with_other_things():  # test_gp.py:95-152
    x = t(np.random.default_rng(1).standard_normal((5, 1)))
    p = st.GP(lambda u: u ** 2, st.EQ())
    five = lambda u: 5 * torch.ones(u.shape[0], 1, dtype=u.dtype)
    kx = B.to_numpy(B.dense(p.kernel(x)))
    for p_sum in (p + 5.0, 5.0 + p, p + five, five + p):
        approx(p_sum.mean(x), B.to_numpy(p.mean(x)) + 5.0)
        approx(B.dense(p_sum.kernel(x)), kx)
    for p_mul in (p * 5.0, 5.0 * p):
        approx(p_mul.mean(x), 5.0 * B.to_numpy(p.mean(x)))
        approx(B.dense(p_mul.kernel(x)), 25.0 * kx)
    with pytest.raises(NotImplementedError):   # products with functions / processes: outside the path
        p * five
    with pytest.raises(NotImplementedError):
        p * p
    with pytest.raises(AssertionError):        # different measures (test_gp.py:24-33)
        p + st.GP(st.EQ())
    with pytest.raises(RuntimeError):          # test_gp.py:41-43
        st.GP().measure


def test_gp_stationarity_and_display():       # test_gp.py:46-50,155-172
    m = st.Measure()
    p1, p2 = st.GP(st.EQ(), measure=m), st.GP(st.EQ().stretch(2), measure=m)
    p = p1 + 2 * p2
    assert p.stationary
    assert not (p + st.GP(st.Linear(), measure=m)).stationary
    assert str(st.GP()) == "GP()"
    assert str(st.GP(st.EQ())) == "GP(0, EQ())"


def test_recycled_process_ids_do_not_inherit_rules():
    """CPython recycles object ids: a process handle created after another one died may get its id; it
    must not pick up the dead handle's cross-kernel rules (found by the multi-process mirrors on the GPU box)."""
    rng = np.random.default_rng(3)
    x = t(rng.standard_normal((6, 1)))
    y = t(rng.standard_normal((6, 1)))
    m = st.Measure()
    p1 = st.GP(st.EQ(), measure=m)
    p2 = st.GP(st.Matern32(), measure=m)
    q = p1 + p2
    post = m | (q(x, 0.1), y)
    want = B.to_numpy(post.kernels[p2, p1].pairwise(x, x))
    for _ in range(50):
        a = post(p1)
        del a
        b = post(p2)                     # frequently lands on the address `a` just vacated
        approx(post.kernels[b, p1].pairwise(x, x), want, atol=1e-12)
        approx(post.kernels[p1, b].pairwise(x, x), want.T, atol=1e-12)
        del b

    # the mechanism, deterministically: index 7 dies and is re-used by a process with other rules
    from stheno_amd.lazy import LazyMatrix
    lm = LazyMatrix()
    lm[1] = "k11"
    lm[7] = "old77"
    lm.add_left_rule(7, {1}, lambda j: "old rule (7, %d)" % j)
    lm.add_right_rule(7, {1}, lambda i: "old rule (%d, 7)" % i)
    assert lm[7, 1] == "old rule (7, 1)" and lm[1, 7] == "old rule (1, 7)"
    lm.purge(7)
    lm[7] = "new77"
    lm.add_left_rule(7, {1}, lambda j: "new rule (7, %d)" % j)
    lm.add_right_rule(7, {1}, lambda i: "new rule (%d, 7)" % i)
    assert lm[7, 1] == "new rule (7, 1)" and lm[1, 7] == "new rule (1, 7)" and lm[7] == "new77"
    # ... and an older rule must not claim to know a newcomer that recycled a dead index
    lm.add_left_rule(1, {7}, lambda j: "rule of 1 about the OLD 7")
    lm.purge(7)
    lm[7] = "newest"
    lm.add_right_rule(7, {1}, lambda i: "newest rule (%d, 7)" % i)
    assert lm[1, 7] == "newest rule (1, 7)"


def test_measure_groups():                    # tests/model/test_model.py:23-59
    prior = st.Measure()
    f1 = st.GP(st.EQ(), measure=prior)
    f2 = st.GP(st.EQ(), measure=prior)
    assert f1._measures == f2._measures == [prior]
    x = t(np.linspace(0, 5, 10))
    y = f1(x).sample()
    post = prior | (f1(x), y)
    assert f1._measures == f2._measures == [prior, post]
    f_sum = f1 + f2                            # known to both measures
    assert f_sum._measures == [prior, post]
    f3 = st.GP(st.EQ(), measure=prior)         # created after the conditioning: the posterior does not know it
    f_sum = f1 + f3
    assert f3._measures == f_sum._measures == [prior]
    with pytest.raises(AssertionError):
        post(f1) + f3
    f_sum = post(f1) + post(f2)                # extend the posterior
    assert f_sum._measures == [post]
    f3 = st.GP(st.EQ(), measure=post)
    f_sum = post(f1) + f3
    assert f3._measures == f_sum._measures == [post]
    with pytest.raises(AssertionError):
        f1 + f3
    del post                                   # (this package: a posterior dies with its last handle)
    import gc
    del f_sum, f3
    gc.collect()
    assert f1._measures == [prior]


def test_normal_mean_is_zero_entropy_and_sampling(normal1):   # test_random.py:53-65,207-209,228-245
    d = st.Normal(torch.eye(3, dtype=f64))
    assert d.mean_is_zero
    approx(d.mean, np.zeros((3, 1)))
    assert st.Normal(Zero(f64, 3, 1), torch.eye(3, dtype=f64)).mean_is_zero
    assert not st.Normal(torch.randn(3, 1, dtype=f64), torch.eye(3, dtype=f64)).mean_is_zero
    sp = multivariate_normal(B.to_numpy(normal1.mean)[:, 0], B.to_numpy(B.dense(normal1.var)))
    approx(normal1.entropy(), sp.entropy(), rtol=1e-10)
    g = gen(0)
    for mean in (0.0, 1.0):
        dist = st.Normal(mean * torch.ones(200, 1, dtype=f64), 3 * torch.eye(200, dtype=f64))
        s = dist.sample(2000, generator=g)
        assert s.shape == (200, 2000)
        assert abs(float(s.mean()) - mean) < 5e-2 and abs(float(s.std()) ** 2 - 3) < 5e-2
        s = dist.sample(2000, noise=2, generator=g)
        assert abs(float(s.mean()) - mean) < 5e-2 and abs(float(s.std()) ** 2 - 5) < 5e-2
    a = dist.sample(generator=gen(7))
    b = dist.sample(generator=gen(7))
    approx(a, b)


def test_pseudoobs_kernel_call_count():       # tests/model/test_model.py:335-365
    """A pseudo-point posterior prediction evaluates the kernel exactly as often as the reference does:
    pairwise (x_ind, x_obs), (x_ind, x_ind), (x_ind, x_new); elwise at x_obs and at x_new."""
    from stheno_amd import ops

    be = ops.get_backend()
    calls = {"kmat": [], "kdiag": []}
    kmat0, kdiag0 = be.kmat, be.kdiag

    def kmat(terms, x, y=None, **kw):
        calls["kmat"].append((x.shape[-2], None if y is None else y.shape[-2]))
        return kmat0(terms, x, y, **kw)

    def kdiag(terms, x):
        calls["kdiag"].append(x.shape[-2])
        return kdiag0(terms, x)

    be.kmat, be.kdiag = kmat, kdiag
    try:
        rng = np.random.default_rng(0)
        x_obs, y_obs = t(np.linspace(0, 5, 10)), t(rng.standard_normal(10))
        x_ind, x_new = t(np.linspace(0, 5, 5)), t(rng.standard_normal(1))
        p = st.GP(1, st.EQ())
        p_post = p | st.PseudoObs(p(x_ind), (p(x_obs, 0.1), y_obs))
        mean, var = p_post(x_new).marginals()
        assert mean.shape == var.shape == (1,)
    finally:
        be.kmat, be.kdiag = kmat0, kdiag0
    assert sorted(calls["kmat"], key=str) == sorted([(5, 10), (5, None), (5, 1)], key=str), calls
    assert sorted(calls["kdiag"]) == [1, 10], calls


def test_normal_arithmetic(normal1):           # test_random.py:248-293
    rng = np.random.default_rng(5)
    normal2 = st.Normal(t(rng.standard_normal((3, 1))), t(np.eye(3) * 0.7 + 0.1))
    a = Dense(t(rng.standard_normal((3, 3))))
    an, m1, v1 = B.to_numpy(a), B.to_numpy(normal1.mean), B.to_numpy(B.dense(normal1.var))
    m2, v2 = B.to_numpy(normal2.mean), B.to_numpy(B.dense(normal2.var))
    approx(normal1.lmatmul(a).mean, an @ m1); approx(B.dense(normal1.lmatmul(a).var), an @ v1 @ an.T)
    approx(normal1.rmatmul(a).mean, an.T @ m1); approx(B.dense(normal1.rmatmul(a).var), an.T @ v1 @ an)
    b = 5.0
    for d in (normal1 * b, b * normal1):
        approx(d.mean, m1 * b); approx(B.dense(d.var), v1 * b ** 2)
    with pytest.raises(TypeError):
        normal1 * normal1
    approx((normal1 + normal2).mean, m1 + m2); approx(B.dense((normal1 + normal2).var), v1 + v2)
    approx((b + normal1).mean, m1 + b)
    with pytest.raises(TypeError):
        normal1 + st.RandomVector()
    approx((-normal1).mean, -m1); approx(B.dense((-normal1).var), v1)
    approx((normal1 - normal2).mean, m1 - m2); approx(B.dense((normal1 - normal2).var), v1 + v2)
    approx((normal2 - normal1).mean, m2 - m1)
    approx((normal1 / b).mean, m1 / b); approx(B.dense((normal1 / b).var), v1 / b ** 2)


def test_normal_m2_diagonalise_kl(normal1):    # test_random.py:160-163,178-182,212-215
    rng = np.random.default_rng(9)
    c = rng.standard_normal((3, 3))
    normal2 = st.Normal(t(rng.standard_normal((3, 1))), t(c @ c.T + 0.5 * np.eye(3)))
    m1, v1 = B.to_numpy(normal1.mean), B.to_numpy(B.dense(normal1.var))
    m2, v2 = B.to_numpy(normal2.mean), B.to_numpy(B.dense(normal2.var))
    approx(B.dense(normal1.m2), v1 + m1 @ m1.T)
    d = normal1.diagonalise()
    assert isinstance(d.var, Diagonal)
    approx(d.mean, m1); approx(B.dense(d.var), np.diag(np.diag(v1)))
    assert float(normal1.kl(normal1)) < 1e-5 and float(normal1.kl(normal2)) > 0.1
    want = 0.5 * (np.trace(np.linalg.solve(v2, v1)) + ((m2 - m1).T @ np.linalg.solve(v2, m2 - m1))[0, 0] - 3
                  + np.linalg.slogdet(v2)[1] - np.linalg.slogdet(v1)[1])
    approx(normal1.kl(normal2), want, rtol=1e-8)
