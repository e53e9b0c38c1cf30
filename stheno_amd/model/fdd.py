"""``FDD``: a GP at finite inputs (``stheno/model/fdd.py``)."""
import torch

from .. import kernels as _k
from ..matrix import AbstractMatrix, Dense, Diagonal, KernelDense, Zero
from ..random import Normal

__all__ = ["FDD"]


def _noise_as_matrix(noise, x, n):
    """Represent noise as a structured matrix (``fdd.py:14-41``): ``None`` -> ``Zero``,
    scalar -> ``Diagonal`` (``B.fill_diag``), vector -> ``Diagonal``, matrix -> ``Dense``."""
    if noise is None:
        return Zero(x.dtype, n, n, device=x.device, batch=tuple(x.shape[:-2]))
    if isinstance(noise, AbstractMatrix):
        return noise
    host_min = None
    if isinstance(noise, (int, float)) and not isinstance(noise, bool):
        host_min = float(noise)       # (what the host knows about the diagonal without reading the device: KernelDense.cond_bound)
        noise = torch.full((), float(noise), dtype=x.dtype, device=x.device)      # (a fill on the device: no host-to-device copy to wait for)
    if not torch.is_tensor(noise):
        noise = torch.as_tensor(noise, dtype=x.dtype, device=x.device)
    noise = noise.to(dtype=x.dtype, device=x.device)
    if noise.dim() == 0:
        d = Diagonal(noise.expand(tuple(x.shape[:-2]) + (n,)).contiguous())
        d.host_min = host_min
        return d
    if noise.dim() == 1 or (x.dim() > 2 and noise.dim() == x.dim() - 1):
        return Diagonal(noise)
    return Dense(noise)


def _multi_parts(p, x):
    """Flatten a (nested) tuple of FDDs and plain inputs into ``(process, input)`` parts; a plain input
    stands for every component process of the product process ``p`` at that input
    (``stheno/mo/kernel.py:58-76``; ``tests/model/test_fdd.py:85-95`` builds such specifications)."""
    if isinstance(x, FDD):
        xr = _k.uprank(x.x)
        return list(xr.parts) if isinstance(xr, _k.MultiInput) else [(x.p, x.x)]
    if isinstance(x, (tuple, list)):
        return [part for e in x for part in _multi_parts(p, e)]
    comps = p._parents if isinstance(p.kernel, _k.MultiOutputKernel) and p._parents else (p,)
    return [(q, x) for q in comps]


class FDD(Normal):
    """Finite-dimensional distribution of process ``p`` at inputs ``x`` with additive
    ``noise``.  Nothing is computed at construction (``fdd.py:59-83``)."""

    def __init__(self, p, x, noise=None):
        self.p = p
        if isinstance(p, int):
            self.x = x
            self.noise = None
            return
        if isinstance(x, (tuple, list)):
            x = _k.MultiInput(_multi_parts(p, x))          # inputs of a product process
        xr = _k.uprank(x)
        self.x = x
        self._xr = xr
        # NB: the constructors below must not capture ``self`` -- a reference cycle would keep
        # the (multi-GB) kernel matrix / Cholesky factor alive until Python's cyclic GC runs.
        nz = self.noise = _noise_as_matrix(noise, xr, p.kernel.num_outputs(xr))

        def var_diag():
            return p.kernel.elwise(xr)[..., 0] + nz.diag()

        def mean_var():
            mean, var = _k.mean_var(p.mean, p.kernel, xr)
            return mean, var + nz

        def mean_var_diag():
            mean, vd = _k.mean_var_diag(p.mean, p.kernel, xr)
            return mean, vd[..., 0] + nz.diag()

        def var():
            k = p.kernel
            if k.input_scaled_view() is not None or isinstance(k, _k.MultiOutputKernel):
                return KernelDense(k, xr, nz)      # K + noise fused, factorised in place
            return k(xr) + nz

        Normal.__init__(self, lambda: p.mean(xr), var, var_diag=var_diag, mean_var=mean_var,
                        mean_var_diag=mean_var_diag)
        self._zero_mean = isinstance(p.mean, _k.ZeroMean)

    def logpdf(self, x):
        """``Normal.logpdf`` plus a guard: a value that came out cut off from the autograd graph although the process
        has learnable quantities (a posterior / multi-process / dense-noise case outside the differentiable paths)
        is refused, never returned silently detached."""
        posterior = not isinstance(self.p, int) and getattr(self.p.measure, "_conditioned_on", None) is not None
        if posterior and torch.is_grad_enabled() and self._learnable(x):
            # a posterior density under learnable quantities: the plain path would factorise the posterior covariance,
            # only for its value to be discarded -- go to the chain-rule form (two prior densities) at once
            via_prior = self._posterior_logpdf_via_prior(x)
            if via_prior is not None:
                return via_prior
        lp, differentiable = Normal._logpdf(self, x)
        if torch.is_grad_enabled() and not isinstance(self.p, int) and not differentiable and self._learnable(x):
            via_prior = None if posterior else self._posterior_logpdf_via_prior(x)
            if via_prior is not None:
                return via_prior
            raise NotImplementedError(
                "this log-density is outside the differentiable paths (one process -- or one batch of independent "
                "data sets -- whose kernel is a sum of primitives, scalar / per-point noise): its value would be "
                "cut off from the autograd graph.  Wrap the call in torch.no_grad() if that is intended"
            )
        return lp

    def _learnable(self, x):
        """Does anything behind this log-density require a gradient (kernel hyper-parameters, inputs, noise, the data)?"""
        from .. import autograd as _ag
        from ..random import _x_requires_grad

        nz = self.noise
        noisy = ((isinstance(nz, Diagonal) and nz.diag().requires_grad)
                 or (isinstance(nz, Dense) and nz.mat is not None and nz.mat.requires_grad))
        return bool(noisy or _x_requires_grad(self._xr) or _ag.kernel_requires_grad(self.p.kernel)
                    or (torch.is_tensor(x) and x.requires_grad))

    def _posterior_logpdf_via_prior(self, y):
        """A posterior log-density under learnable quantities, by the chain rule: with the process conditioned on exact
        observations ``(fdd_obs, y_obs)`` of its prior measure,
        ``log p(y | y_obs) = log p(y_obs, y) - log p(y_obs)`` -- two PRIOR log-densities, both on the differentiable paths
        (one process: the fused one; several: the block one).  None when that form does not apply (pseudo-point posteriors,
        processes built under the posterior measure, several columns of ``y``)."""
        from .observations import Observations

        link = getattr(self.p.measure, "_conditioned_on", None)
        parents = getattr(self.p, "_parents", None)
        if link is None or not parents or len(parents) != 1 or not torch.is_tensor(y):
            return None
        prior, obs = link
        if not isinstance(obs, Observations) or parents[0].measure is not prior:
            return None
        y2 = y if y.dim() >= 2 else y[:, None]
        if y2.dim() != 2 or y2.shape[-1] != 1 or bool(torch.isnan(y2).any()):
            return None
        here = FDD(parents[0], self.x, self.noise)
        joint = prior.logpdf((obs.fdd, obs.y), (here, y2))
        marginal = prior.logpdf(obs.fdd, obs.y)
        if not (torch.is_tensor(joint) and joint.requires_grad):
            return None
        return joint - marginal

    @property
    def dtype(self):
        return self._xr.dtype if not isinstance(self.p, int) else self.x.dtype

    def __repr__(self):
        return f"<FDD:\n process={self.p!r},\n input={self.x!r},\n noise={self.noise!r}>"

    __str__ = __repr__


def take(fdd, mask):
    """``B.take(fdd, mask)`` (``fdd.py:125-132``): sub-select observations by a boolean mask."""
    if mask.dtype != torch.bool:
        raise AssertionError("Can only take from finite-dimensional distributions according to a mask.")
    idx = torch.nonzero(mask)[:, 0]
    noise = fdd.noise
    if isinstance(noise, Diagonal):
        noise = Diagonal(noise.diag()[idx])
    elif isinstance(noise, Dense):
        noise = Dense(noise.mat[idx][:, idx])
    elif isinstance(noise, Zero):
        noise = None
    return FDD(fdd.p, _k.uprank(fdd.x)[idx], noise)
