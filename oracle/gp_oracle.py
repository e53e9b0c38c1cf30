"""CPU oracle for the dense GP inference hot path of wesselb/stheno.

TEST INFRASTRUCTURE ONLY.  Nothing under ``stheno_amd/`` imports this module; the
only permitted callers are ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  It is never the thing that is shipped or
measured as the product.

What it restates
----------------
Stheno's NumPy path for: kernel-matrix construction, ``chol(K + noise + eps I)``,
``Normal.logpdf``, exact posterior mean / covariance / marginal variance, the
batched variants, and the VFE / FITC / DTC pseudo-point ELBO and posterior.

The reference repository (``/root/reference``) holds only the model algebra.  The
arithmetic lives in un-vendored dependencies that are absent from this image and
cannot be installed (no network); pins are lower bounds only (``setup.py:3-12``):

* ``mlkernels>=0.3.6``      -- kernels, ``pairwise``/``elwise``, ``PosteriorMean``,
  ``PosteriorKernel``, ``SubspaceKernel``, ``mean_var_diag``
* ``backends-matrix>=1.2.11`` -- ``Dense``/``Diagonal``, ``cholesky``, ``logdet``,
  ``iqf``, ``iqf_diag``, ``ratio``, ``matmul_diag``
* ``backends>=1.4.11`` (``lab``) -- ``B.epsilon``, ``B.reg``, ``B.pw_dists2``

Their published algorithms are restated here with NumPy/SciPy (LAPACK-backed, the
same vendor math the reference ends up in), anchored on the reference's own call
sites, each cited as ``file:line`` relative to the reference repository.

Pinning (see tests/test_oracle_pins.py and tests/golden/):
  * ``Normal.logpdf`` == ``scipy.stats.multivariate_normal.logpdf`` -- the check the
    reference's own test makes (``tests/test_random.py:185-192``);
  * README known answers: kernel matrix, two logpdf values, posterior mean and
    covariance (``README.md:43-86``, ``README.md:470-497``);
  * inducing points == inputs  =>  ELBO == logpdf and approximate posterior == exact
    posterior for VFE/FITC/DTC (``tests/model/test_model.py:283-308``);
  * ``marginals`` == diag of the full posterior (``tests/model/test_fdd.py:111-134``).
"""
import numpy as np
import scipy.linalg as sla

EPSILON_DEFAULT = 1e-12  # lab's ``B.epsilon`` default (README.md:820-831)

LOG_2_PI = float(np.log(2 * np.pi))


# ---------------------------------------------------------------------------
# lab: B.uprank / B.pw_dists2 / B.reg
# ---------------------------------------------------------------------------
def uprank(x):
    """``B.uprank``: rank-0/1 inputs become column matrices (random.py:259, fdd.py)."""
    x = np.asarray(x)
    if x.ndim == 0:
        return x.reshape(1, 1)
    if x.ndim == 1:
        return x[:, None]
    return x


def pw_dists2(x, y):
    """``B.pw_dists2``: squared Euclidean distances between rows of ``x`` and ``y``.

    Upstream uses the direct difference for one-dimensional inputs and
    ``|a|^2 + |b|^2 - 2 a.b`` otherwise (formula evidence for the EQ kernel built on
    it: ``tests/model/test_model.py:342-345``).
    """
    x, y = uprank(x), uprank(y)
    if x.shape[-1] == 1:
        return (x - np.swapaxes(y, -1, -2)) ** 2
    nx = np.sum(x**2, axis=-1)[..., :, None]
    ny = np.sum(y**2, axis=-1)[..., None, :]
    d2 = nx + ny - 2 * np.matmul(x, np.swapaxes(y, -1, -2))
    return np.maximum(d2, 0)


def ew_dists2(x, y):
    x, y = uprank(x), uprank(y)
    return np.sum((x - y) ** 2, axis=-1)[..., None]


def reg(a, eps):
    """``B.reg``: add ``B.epsilon`` to the diagonal before every Cholesky."""
    n = a.shape[-1]
    return a + eps * np.eye(n, dtype=a.dtype)


# ---------------------------------------------------------------------------
# mlkernels: primitive kernels, as a sum of (kind, variance, length scale) terms
# ---------------------------------------------------------------------------
KINDS = ("eq", "matern12", "matern32", "matern52", "linear", "const")


def _kappa(kind, d2, dot):
    if kind == "eq":
        return np.exp(-0.5 * d2)
    if kind == "matern12":
        return np.exp(-np.sqrt(d2))
    if kind == "matern32":
        r = np.sqrt(3.0) * np.sqrt(d2)
        return (1 + r) * np.exp(-r)
    if kind == "matern52":
        r = np.sqrt(5.0) * np.sqrt(d2)
        return (1 + r + r**2 / 3) * np.exp(-r)
    if kind == "linear":
        return dot
    if kind == "const":
        return np.ones_like(d2)
    raise ValueError(kind)


def kernel_matrix(terms, x, y=None):
    """``k(x, y)`` for ``k = sum_t variance_t * kind_t.stretch(scale_t)``.

    ``terms``: iterable of ``(kind, variance, scale)``.  Stretching divides the inputs
    by the length scale (mlkernels ``Stretched``).  Call sites: ``fdd.py:79``,
    ``observations.py:139,285-286``.
    """
    x = uprank(x)
    y = x if y is None else uprank(y)
    out = None
    for kind, variance, scale in terms:
        xs, ys = x / scale, y / scale
        if kind == "linear":
            k = _kappa(kind, None, np.matmul(xs, np.swapaxes(ys, -1, -2)))
        else:
            k = _kappa(kind, pw_dists2(xs, ys), None)
        k = variance * k
        out = k if out is None else out + k
    return out


def kernel_diag(terms, x):
    """``k.elwise(x)`` (``fdd.py:66``, ``observations.py:304``) as a vector."""
    x = uprank(x)
    out = np.zeros(x.shape[:-1], dtype=x.dtype)
    for kind, variance, scale in terms:
        if kind == "linear":
            out = out + variance * np.sum((x / scale) ** 2, axis=-1)
        else:
            out = out + variance
    return out


def noise_matrix(noise, n, dtype):
    """``_noise_as_matrix`` (``fdd.py:14-41``) densified."""
    if noise is None:
        return np.zeros((n, n), dtype=dtype)
    noise = np.asarray(noise, dtype=dtype)
    if noise.ndim == 0:
        return noise * np.eye(n, dtype=dtype)
    if noise.ndim == 1:
        return np.diag(noise)
    return noise


def noise_diag(noise, n, dtype):
    if noise is None:
        return np.zeros(n, dtype=dtype)
    noise = np.asarray(noise, dtype=dtype)
    if noise.ndim == 0:
        return np.full(n, noise, dtype=dtype)
    if noise.ndim == 1:
        return noise
    return np.diag(noise)


# ---------------------------------------------------------------------------
# matrix: cholesky / logdet / iqf / iqf_diag on Dense
# ---------------------------------------------------------------------------
def cholesky(a, eps=EPSILON_DEFAULT):
    """``B.cholesky(B.reg(a))`` -- LAPACK potrf (implicit under random.py:274-276)."""
    return np.linalg.cholesky(reg(a, eps))


def logdet_chol(chol):
    return 2 * np.sum(np.log(np.diagonal(chol, axis1=-2, axis2=-1)), axis=-1)


def solve_lower(chol, b):
    """``B.solve(L, b)`` / ``B.triangular_solve`` (observations.py:301)."""
    return sla.solve_triangular(chol, b, lower=True, check_finite=False)


def iqf_diag(chol, b):
    """``B.iqf_diag(K, b)`` = column-wise ``|L^{-1} b|^2`` (random.py:276)."""
    v = solve_lower(chol, b)
    return np.sum(v * v, axis=0)


# ---------------------------------------------------------------------------
# stheno.random.Normal.logpdf  (random.py:248-280)
# ---------------------------------------------------------------------------
def normal_logpdf(mean, var, y, eps=EPSILON_DEFAULT):
    """``Normal(mean, var).logpdf(y)``; ``y``: ``(N,)`` or ``(N, C)``.  Returns a scalar
    for one column, ``(C,)`` otherwise (random.py:280)."""
    y = uprank(y)
    n = var.shape[-1]
    mean = np.zeros((n, 1), dtype=var.dtype) if mean is None else uprank(mean)
    chol = cholesky(var, eps)
    out = -(logdet_chol(chol) + n * LOG_2_PI + iqf_diag(chol, y - mean)) / 2
    return out[0] if out.shape[0] == 1 else out


def gp_logpdf(terms, x, noise, y, eps=EPSILON_DEFAULT, mean=None):
    """``f(x, noise).logpdf(y)`` for ``f = GP(kernel)`` (gp.py:134-144, fdd.py:59-83,
    random.py:248-280).  Unbatched."""
    x = uprank(x)
    n = x.shape[0]
    var = kernel_matrix(terms, x) + noise_matrix(noise, n, x.dtype)
    return normal_logpdf(mean, var, y, eps)


def gp_logpdf_batched(terms, x, noise, y, eps=EPSILON_DEFAULT):
    """Batched computation (README.md:744-766, tests/model/test_cases.py:134-155):
    ``x`` ``(B, N, D)``, ``y`` ``(B, N, 1)`` -> ``(B,)``."""
    return np.array([gp_logpdf(terms, x[b], noise, y[b], eps) for b in range(x.shape[0])])


# ---------------------------------------------------------------------------
# exact conditioning: Observations + PosteriorMean / PosteriorKernel
# (observations.py:127-168; formulas of mlkernels' PosteriorMean/PosteriorKernel)
# ---------------------------------------------------------------------------
def gp_posterior(terms, x, noise, y, xs, eps=EPSILON_DEFAULT, noise_s=None, full_cov=True):
    """Posterior of ``f`` given ``(f(x, noise), y)`` evaluated at ``f_post(xs, noise_s)``.

    Returns ``(mean (Ns,), var (Ns, Ns) or None, var_diag (Ns,))``; ``var_diag`` is
    computed through the ``elwise`` route of ``mean_var_diag`` (fdd.py:72-74) and is
    NOT clamped (``marginals()`` clamps at zero, random.py:226).
    """
    x, xs = uprank(x), uprank(xs)
    y = uprank(y)
    n, ns = x.shape[0], xs.shape[0]
    k_x = kernel_matrix(terms, x) + noise_matrix(noise, n, x.dtype)   # observations.py:139
    chol = cholesky(k_x, eps)
    k_xs = kernel_matrix(terms, x, xs)
    v = solve_lower(chol, k_xs)                  # L^{-1} k(x, xs)
    w = solve_lower(chol, y)                     # L^{-1} (y - m(x)), zero prior mean
    mean = (v.T @ w)[:, 0]
    var_diag = kernel_diag(terms, xs) - np.sum(v * v, axis=0) + noise_diag(noise_s, ns, x.dtype)
    var = None
    if full_cov:
        var = kernel_matrix(terms, xs) - v.T @ v + noise_matrix(noise_s, ns, x.dtype)
    return mean, var, var_diag


def marginals(mean, var_diag):
    """``Normal.marginals`` clamp (random.py:224-227)."""
    return mean, np.maximum(var_diag, 0)


def credible_bounds(mean, var_diag):
    """``Normal.marginal_credible_bounds`` (random.py:229-238)."""
    mean, var = marginals(mean, var_diag)
    err = 1.96 * np.sqrt(var)
    return mean, mean - err, mean + err


# ---------------------------------------------------------------------------
# pseudo-point approximations (observations.py:279-336)
# ---------------------------------------------------------------------------
def pseudo_obs(terms, x, noise, y, z, noise_z=None, method="vfe", eps=EPSILON_DEFAULT):
    """``AbstractPseudoObservations._compute``.  Returns a dict with ``elbo``, ``mu``
    (M, 1), ``A`` (= L_z A L_z^T, M x M), ``K_z``, and the pieces needed for the
    approximate posterior."""
    x, z = uprank(x), uprank(z)
    y = uprank(y)
    n, m = x.shape[0], z.shape[0]
    k_zx = kernel_matrix(terms, z, x)                                   # :285
    k_z = kernel_matrix(terms, z) + noise_matrix(noise_z, m, x.dtype)   # :286
    k_n = noise_diag(noise, n, x.dtype).copy()                          # :290 (must be diagonal, :293-297)
    l_z = cholesky(k_z, eps)                                            # :300
    ilz_kzx = solve_lower(l_z, k_zx)                                    # :301
    if method in ("vfe", "fitc"):
        k_x_diag = kernel_diag(terms, x)                                # :304
        q_x_diag = np.sum(ilz_kzx * ilz_kzx, axis=0)                    # :305
        corr = k_x_diag - q_x_diag                                      # :306
    if method == "vfe":
        trace_part = np.sum(corr / k_n)                                 # :310  B.ratio(Diag, Diag)
    elif method == "fitc":
        k_n = k_n + corr                                                # :312
        trace_part = 0.0
    elif method == "dtc":
        trace_part = 0.0
    else:
        raise ValueError(method)
    a = np.eye(m, dtype=x.dtype) + (ilz_kzx / k_n) @ ilz_kzx.T          # :322
    a_big = l_z @ a @ l_z.T                                             # :323
    y_bar = y                                                           # :326, zero prior mean
    prod_y_bar = (ilz_kzx / k_n) @ y_bar                                # :327
    l_a = cholesky(a, eps)
    t = solve_lower(l_a, l_z.T)
    u = solve_lower(l_a, prod_y_bar)
    mu = t.T @ u                                                        # :329
    det_part = np.sum(np.log(2 * np.pi * k_n)) + logdet_chol(l_a)       # :334
    iqf_part = np.sum(y_bar[:, 0] ** 2 / k_n) - np.sum(u[:, 0] ** 2)    # :335
    elbo = -0.5 * (det_part + iqf_part + trace_part)                    # :336
    return dict(elbo=float(elbo), mu=mu, A=a_big, K_z=k_z, l_z=l_z)


def pseudo_posterior(terms, x, noise, y, z, xs, noise_z=None, method="vfe", eps=EPSILON_DEFAULT, full_cov=True):
    """Approximate posterior (observations.py:255-277): kernel
    ``PosteriorKernel(K_z) + SubspaceKernel(A)``, mean ``PosteriorMean(K_z, mu)``."""
    z, xs = uprank(z), uprank(xs)
    r = pseudo_obs(terms, x, noise, y, z, noise_z, method, eps)
    l_z = cholesky(r["K_z"], eps)
    k_zs = kernel_matrix(terms, z, xs)
    v = solve_lower(l_z, k_zs)
    mean = (v.T @ solve_lower(l_z, r["mu"]))[:, 0]
    l_big = cholesky(r["A"], eps)
    s = solve_lower(l_big, k_zs)
    var_diag = kernel_diag(terms, xs) - np.sum(v * v, axis=0) + np.sum(s * s, axis=0)
    var = None
    if full_cov:
        var = kernel_matrix(terms, xs) - v.T @ v + s.T @ s
    return mean, var, var_diag


# ---------------------------------------------------------------------------
# sampling (adjacent; random.py:331-363): chol(var) @ xi
# ---------------------------------------------------------------------------
def sample(var, xi, eps=EPSILON_DEFAULT, mean=None):
    out = cholesky(var, eps) @ xi
    return out if mean is None else out + uprank(mean)
