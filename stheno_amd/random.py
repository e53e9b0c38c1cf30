"""``Normal``: the multivariate normal of the reference (``stheno/random.py``), restated
over torch tensors on a HIP device, with the lazy mean / variance / variance-diagonal
resolution semantics of ``random.py:96-117,149-227`` and ``logpdf`` of
``random.py:248-280``.

What is computed once: the variance is resolved once; its Cholesky factor is cached on
the matrix object and shared by ``logdet`` and ``iqf_diag`` (one POTRF per logpdf, one
per conditioning).
"""
import types

import torch

from . import ops
from .matrix import LOG_2_PI, AbstractMatrix, Dense, Diagonal, KernelDense, Zero, any_missing, config, deferred_checks, to_matrix

__all__ = ["Random", "RandomProcess", "RandomVector", "Normal"]


class Random:
    """A random object."""

    def __radd__(self, other):
        return self + other

    def __rmul__(self, other):
        return self * other

    def __neg__(self):
        return -1 * self

    def __sub__(self, other):
        return self + (-other)

    def __rsub__(self, other):
        return (-self) + other

    def __truediv__(self, other):
        return self * (1 / other)


class RandomProcess(Random):
    """A random process."""


class RandomVector(Random):
    """A random vector."""


def _x_requires_grad(x):
    """Whether (any component of) an input specification carries gradients."""
    if torch.is_tensor(x):
        return x.requires_grad
    return any(_x_requires_grad(q) for _, q in getattr(x, "parts", ()))


def _is_zero(x):
    return isinstance(x, (int, float)) and not isinstance(x, bool) and x == 0


class _Deferred:
    """One quantity of a distribution that may not exist yet: given up front, or made on first use by a zero-argument callable
    (the reference's ``Normal`` takes such constructors so that a prior's ``f(x)`` costs nothing until something is asked of it,
    ``random.py:56-117``)."""

    __slots__ = ("value", "make")

    def __init__(self, value=None, make=None):
        self.value, self.make = value, make

    @property
    def known(self):
        return self.value is not None

    def get(self):
        if self.value is None and self.make is not None:
            self.value = self.make()
        return self.value


class Normal(RandomVector):
    """Normal random variable.

    ``Normal(var)``, ``Normal(mean, var)`` with tensors / structured matrices, or
    ``Normal(mean_fn, var_fn, var_diag=..., mean_var=..., mean_var_diag=...)`` with
    zero-argument constructors that are called lazily (``random.py:56-94``).

    State: three deferred quantities (mean, variance, marginal variances) and two optional JOINT constructors that produce the mean
    together with the variance / the marginal variances in one go (a posterior shares the whitened cross-kernel between them).  A
    joint constructor is only worth calling while neither of its two quantities exists; after that the single ones are used.
    """

    def __init__(self, *args, var_diag=None, mean_var=None, mean_var_diag=None):
        if len(args) == 1:
            mean, var = (lambda: 0) if isinstance(args[0], types.FunctionType) else 0, args[0]
        elif len(args) == 2:
            mean, var = args
        else:
            raise TypeError("Normal(var) or Normal(mean, var)")
        lazy = isinstance(var, types.FunctionType)
        if lazy and not isinstance(mean, types.FunctionType):
            raise TypeError("mean and var must both be constructors or both be values")
        self._m = _Deferred(make=mean) if lazy else _Deferred(mean)
        self._v = _Deferred(make=var) if lazy else _Deferred(var)
        self._d = _Deferred(make=var_diag if lazy else None)
        self._joint_var = mean_var if lazy else None
        self._joint_diag = mean_var_diag if lazy else None
        self._mean_vanishes = None       # whether the mean is identically zero, decided when the mean first exists

    def computed(self, what):
        """Whether ``"mean"``, ``"var"`` or ``"var_diag"`` has been given or computed already (nothing is computed by asking)."""
        return {"mean": self._m, "var": self._v, "var_diag": self._d}[what].known

    # -- the deferred quantities (random.py:96-202) ----------------------------
    def _settle_mean(self, as_tensor):
        """The mean exists after this; a scalar zero stays a scalar unless ``as_tensor`` (then it becomes a zero column of the
        variance's shape and type, which makes the variance exist too)."""
        m = self._m.get()
        if self._mean_vanishes is None:
            self._mean_vanishes = _is_zero(m) or isinstance(m, Zero)
        if as_tensor and _is_zero(m):
            var = self.var
            shape = tuple(var.shape)
            self._m.value = torch.zeros(shape[:-2] + (shape[-1], 1), dtype=var.dtype, device=var.device)

    def _settle_jointly(self, other, joint):
        """``other`` is the variance or the marginal-variance slot: while neither it nor the mean exists, one call of ``joint``
        (if there is one) makes both."""
        if joint is not None and not self._m.known and not other.known:
            self._m.value, other.value = joint()

    def __repr__(self):
        m = repr(self._m.value) if self._m.known else "unresolved"
        v = repr(self._v.value) if self._v.known else "unresolved"
        return f"<Normal:\n mean={m},\n var={v}>"

    __str__ = __repr__

    @property
    def _mean_t(self):
        """The mean as a plain column-vector tensor (what every computation below uses)."""
        self._settle_mean(as_tensor=True)
        m = self._m.value
        return m.dense() if isinstance(m, AbstractMatrix) else m

    @property
    def mean(self):
        """Column vector: mean.  A plain tensor -- or, with ``config.mean_as_matrix``, the reference's return type (``matrix.Dense``,
        README.md:58-68; ``B.dense`` strips it)."""
        m = self._mean_t
        return Dense(m) if config.mean_as_matrix else m

    @property
    def mean_is_zero(self):
        self._settle_mean(as_tensor=False)
        return self._mean_vanishes

    @property
    def var(self):
        """Variance as a structured matrix (``B.dense(d.var)`` for the plain tensor)."""
        self._v.value = to_matrix(self._v.get())
        return self._v.value

    @property
    def var_diag(self):
        if not self._d.known and self._d.make is None:
            self._d.value = self.var.diag()
        return self._d.get()

    @property
    def mean_var(self):
        self._settle_jointly(self._v, self._joint_var)
        return self.mean, self.var

    @property
    def dtype(self):
        return self.var.dtype

    @property
    def dim(self):
        return self.var.shape[-1]

    # -- marginals (random.py:204-238) -----------------------------------------
    def marginals(self):
        """Marginal means and variances (the covariance is not formed when a
        ``mean_var_diag`` constructor is available)."""
        self._settle_jointly(self._d, self._joint_diag)
        mean, var_diag = self._mean_t, self.var_diag
        if isinstance(var_diag, AbstractMatrix):
            var_diag = var_diag.dense()
        # Variances can come out slightly negative through round-off (random.py:221-227).
        if torch.is_tensor(mean) and mean.dim() >= 1:
            mean = mean[..., 0]
        if torch.is_tensor(var_diag):
            var_diag = torch.clamp(var_diag, min=0)
        return mean, var_diag

    def marginal_credible_bounds(self):
        """Marginal means with lower and upper 95% central credible bounds."""
        mean, var = self.marginals()
        error = 1.96 * torch.sqrt(var)
        return mean, mean - error, mean + error

    # -- logpdf (random.py:248-280) --------------------------------------------
    def logpdf(self, x):
        """Log-density at ``x``: ``(N,)``/``(N, 1)`` -> scalar tensor, ``(N, C)`` -> ``(C,)``,
        batched ``(B, N, 1)`` -> ``(B,)``."""
        return self._logpdf(x)[0]

    def _logpdf(self, x):
        """``(value, differentiable)``: whether the value came from one of the autograd paths (``stheno_amd/autograd.py``) --
        everything else is computed by ``libgpk.so`` on detached buffers, whatever ``value.requires_grad`` says."""
        if not torch.is_tensor(x):
            x = torch.as_tensor(x, dtype=self.dtype, device=self.var.device)
        if x.dim() <= 1:
            x = x.reshape(-1, 1)

        # Missing data (not for batched computation): random.py:261-270.
        if config.check_nan and x.dim() == 2 and x.shape[1] == 1:
            if any_missing(x):
                available = ~torch.isnan(x[:, 0])
                idx = torch.nonzero(available)[:, 0]
                mean = self._mean_t[idx]
                var = self.var
                if isinstance(var, KernelDense) and torch.is_tensor(var.x) and var.x.dim() == 2 and var._mat is None and not isinstance(var.noise, Dense) \
                        and var.kernel.num_outputs(var.x) == var.x.shape[0]:
                    # a kernel matrix restricted to the observed points is the kernel matrix OF those points:
                    # stay lazy (lower-only build, in-place factor) and differentiable
                    noise = var.noise
                    if isinstance(noise, Diagonal):
                        noise = Diagonal(noise.diag()[idx])
                    elif isinstance(noise, Zero):
                        noise = Zero(noise.dtype, idx.numel(), idx.numel(), device=noise.device)
                    sub = KernelDense(var.kernel, var.x[idx], noise)
                else:
                    sub = var.dense()[idx][:, idx]
                return Normal(mean, sub)._logpdf(x[idx])

        var = self.var
        n = self.dim
        # (a zero prior mean: the residual IS the data -- no zeros materialised, nothing subtracted; aliasing is safe: every consumer
        #  below reads `r` or copies it before solving in place)
        if (getattr(self, "_zero_mean", False) and x.dtype == self.dtype and x.dim() == 2 and x.shape[0] == n and not x.requires_grad
                and x.device == self.var.device):
            r = x
        else:
            r = x - self._mean_t
        # hyper-parameter learning: differentiable path for a kernel-matrix variance
        batched_ok = (r.dim() == 3 and r.shape[-1] == 1 and torch.is_tensor(getattr(var, "x", None)) and var.x.dim() == 3
                      and tuple(var.x.shape[:-2]) == tuple(r.shape[:-2]))
        if isinstance(var, KernelDense) and (r.dim() == 2 or batched_ok):
            from . import autograd as _ag

            noise_vec, noise_mat = var.differentiable_noise(), None
            if noise_vec is NotImplemented and r.dim() == 2 and isinstance(var.noise, Dense) and var.noise.mat is not None \
                    and var.noise.mat.dim() == 2:
                noise_vec, noise_mat = None, var.noise.mat       # a dense noise covariance: its cotangent is that of K
            if noise_vec is not None and noise_vec is not NotImplemented and noise_vec.dim() != r.dim() - 1:
                noise_vec = NotImplemented
            # k(x) = k0(x / l) with k0 a sum of primitives (l: per-dimension length scales, or none): the fused path runs
            # k0 on the divided inputs; torch differentiates the division (d/dl, d/dx)
            view = var.kernel.input_scaled_view() if torch.is_tensor(var.x) else None
            if noise_vec is not NotImplemented and view is not None:
                kern, scales = view
                xin = var.x if scales is None else var.x / scales.to(dtype=var.x.dtype, device=var.x.device)
                tt = kern.tensor_terms()
                if tt is not None and _ag.needs_grad(tt, noise_vec, r, xin, noise_mat):
                    if xin.requires_grad and torch.is_grad_enabled() and xin.shape[-1] > 8:
                        raise NotImplementedError("gradients with respect to the inputs (or per-dimension length scales) "
                                                  "are implemented for at most 8 input dimensions")
                    lp = _ag.gp_logpdf(kern, xin, noise_vec, r, noise_mat)
                    if r.dim() == 3:
                        return lp, True
                    return (lp[0] if lp.shape[0] == 1 else lp), True
        if torch.is_grad_enabled() and isinstance(var, KernelDense) and r.dim() == 2:
            from . import autograd as _ag
            from . import kernels as _kk

            if isinstance(var.kernel, _kk.MultiOutputKernel):      # several processes observed jointly
                noise_vec = var.differentiable_noise()
                if noise_vec is None or (noise_vec is not NotImplemented and noise_vec.dim() == 1):
                    lp = _ag.joint_logpdf(var.kernel, var.x, noise_vec, r, config.epsilon)
                    if lp is not None:
                        return (lp[0] if lp.shape[0] == 1 else lp), True
        if torch.is_grad_enabled() and isinstance(var, KernelDense):
            from . import autograd as _ag

            nz = var.noise
            noisy = (isinstance(nz, Diagonal) and nz.diag().requires_grad) or (isinstance(nz, Dense) and nz.mat is not None
                                                                               and nz.mat.requires_grad)
            if noisy or r.requires_grad or _x_requires_grad(var.x) or _ag.kernel_requires_grad(var.kernel):
                raise NotImplementedError(
                    "gradients of logpdf are implemented for one process (or one batch of independent data sets) whose kernel "
                    "is a sum of primitives with scalar or per-point noise; this call (multi-process, posterior or "
                    "dense-noise) would return a value cut off from the autograd graph -- wrap it in torch.no_grad() "
                    "if that is intended"
                )
        if isinstance(var, Zero):
            raise torch.linalg.LinAlgError("the variance is identically zero")
        with deferred_checks():      # the factor's `info` is read after the log-determinant and the solve are queued behind it
            # (a zero prior mean: `r` IS `x`; naming it lets the factor hand the same L^{-1} y to the posterior mean later)
            src = x if (getattr(self, "_zero_mean", False) and isinstance(var, Dense)) else None
            if hasattr(var, "chol_with_rhs"):
                var.chol_with_rhs(r, src)      # (a batch that has not been factorised yet: L^{-1} r comes out of the factorisation)
            logdet = var.logdet()
            iqf = var.iqf_diag(r, src) if src is not None else var.iqf_diag(r)
            logpdfs = torch.add(logdet[..., None], iqf).add_(n * LOG_2_PI).mul_(-0.5)       # -(logdet + n log 2 pi + iqf) / 2 in three launches
        return (logpdfs[..., 0] if logpdfs.shape[-1] == 1 else logpdfs), False

    def entropy(self):
        return (self.var.logdet() + self.dim * (LOG_2_PI + 1)) / 2

    @property
    def m2(self):
        """Second moment ``V + m m^T`` (``random.py:200-202``)."""
        m = self._mean_t.contiguous()
        be = ops.get_backend()
        return Dense(be.gemm(m, m, a_kmajor=True, b_kmajor=True, alpha=1.0, beta=1.0, out=be.copy(self.var.dense())))

    def diagonalise(self):
        """The distribution with its correlations set to zero (``random.py:240-246``)."""
        return Normal(self._mean_t, Diagonal(self.var_diag))

    def kl(self, other):
        """``KL(self || other)`` (``random.py:294-311``): ``tr(V_o^{-1} V_s) = |L_o^{-1} L_s|_F^2`` from the two
        Cholesky factors (one TRSM), the quadratic term by a TRSV."""
        vs, vo = to_matrix(self.var), to_matrix(other.var)
        if isinstance(vs, Diagonal) or isinstance(vo, Diagonal):
            vs, vo = Dense(vs.dense()), Dense(vo.dense())
        w = vo.chol().solve(vs.chol().lower())
        _, ss = ops.get_backend().colreduce(w, want_ss=True)
        ratio = ss.sum(-1)
        iqf = vo.iqf_diag(other._mean_t - self._mean_t)[..., 0]
        return (iqf + ratio + vo.logdet() - vs.logdet() - self.dim) / 2

    # -- sampling (random.py:331-363; adjacent to the hot path) ----------------
    def sample(self, num=1, noise=None, generator=None, xi=None):
        """Samples as column vectors (..., N, num): ``chol(var) @ xi`` on the MFMA GEMM.

        ``xi``: the standard-normal draws to transform, (..., N, num) -- what ``B.randn`` produces inside the
        reference's ``B.sample`` (``random.py:351``); given, the result is a deterministic function of the
        distribution (used to compare samples with the oracle)."""
        var = self.var
        if xi is not None:
            if xi.dim() == 1:
                xi = xi[:, None]
            if xi.shape[-2] != self.dim:
                raise ValueError(f"xi has {xi.shape[-2]} rows, the distribution has dimension {self.dim}")
            num = xi.shape[-1]
        if isinstance(var, Diagonal):
            d = var.diag() + (noise if noise is not None else 0)
            if xi is None:
                xi = torch.randn(d.shape + (num,), dtype=d.dtype, device=d.device, generator=generator)
            out = torch.sqrt(d)[..., None] * xi
        else:
            if noise is not None:
                var = var + Diagonal(torch.full(tuple(var.shape[:-1]), float(noise), dtype=var.dtype, device=var.device))
            chol = var.chol() if isinstance(var, Dense) else to_matrix(var.dense()).chol()
            l = chol.lower()
            if xi is None:
                xi = torch.randn(tuple(var.shape[:-1]) + (num,), dtype=var.dtype, device=var.device, generator=generator)
            out = ops.get_backend().gemm(l, xi.to(var.dtype), a_kmajor=True, b_kmajor=False)
        if not self.mean_is_zero:
            out = out + self._mean_t
        return out

    # -- arithmetic (random.py:365-393) ----------------------------------------
    def __add__(self, other):
        if isinstance(other, Normal):
            return Normal(self._mean_t + other._mean_t, self.var + other.var)
        if isinstance(other, Random):
            raise TypeError(f"cannot add a {type(other).__name__} to a Normal")
        return Normal(self._mean_t + other, self.var)

    def __mul__(self, other):
        if isinstance(other, Random):
            raise TypeError(f"cannot multiply a Normal by a {type(other).__name__}")
        return Normal(self._mean_t * other, Dense(self.var.dense() * (other * other)))

    def lmatmul(self, a):
        """Distribution of ``a @ x`` (``random.py:365-371``): mean ``a m``, variance ``a V a^T`` (two GEMMs)."""
        a = a.dense() if isinstance(a, AbstractMatrix) else a
        be = ops.get_backend()
        av = be.gemm(a, self.var.dense(), a_kmajor=True, b_kmajor=False)          # a V
        return Normal(be.gemm(a, self._mean_t, a_kmajor=True, b_kmajor=False), Dense(be.gemm(av, a, a_kmajor=True, b_kmajor=True)))

    def rmatmul(self, a):
        """Distribution of ``a^T @ x`` (``random.py:373-379``)."""
        a = a.dense() if isinstance(a, AbstractMatrix) else a
        return self.lmatmul(a.transpose(-1, -2).contiguous())
