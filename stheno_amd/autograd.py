"""Differentiable ``f(x, noise).logpdf(y)`` -- row 8(f)-1 of SURVEY.md: the dominant use of
``stheno.torch`` is hyper-parameter learning (``readme_example13_optimisation_torch.py:47-53``).

Forward: the same HIP path as the plain logpdf (fused kernel matrix, in-place Cholesky, GEMV
sweep).  Backward, from the stored factor:

    G = d logpdf / dK = 1/2 (A diag(g) A^T - sum(g) K^{-1}),     A = K^{-1} (y - m)

``K^{-1} = W^T W`` with ``W = L^{-1}`` (blocked TRSM on the identity + lower SYRK, both on the
MFMA GEMM), then ONE pass over the lower triangle of ``K^{-1}`` (``gpk_kmat_vjp``) yields the
gradients w.r.t. every variance, length scale and the noise; ``d/d(y - m) = -A g``.
Gradients w.r.t. the inputs ``x`` are not provided.
"""
import torch

from . import ops
from .matrix import LOG_2_PI, Chol, config

__all__ = ["gp_logpdf"]


class _GPLogpdf(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, noise_vec, kinds, *params):
        """``params`` = variances then scales (scalar tensors, any device); ``noise_vec`` (n,)
        or None; ``r = y - mean`` (n, C).  Returns (C,)."""
        be = ops.get_backend()
        nt = len(kinds)
        variances, scales = params[:nt], params[nt:]
        terms = ops.KTerms([(k, float(v), float(s)) for k, v, s in zip(kinds, variances, scales)])
        n, C = r.shape
        k = be.kmat(terms, x, None, lower=True, diag_add=config.epsilon, diag_vec=noise_vec)
        chol = Chol.factor_(k)
        w = chol.solve(r)                                        # L^{-1} r
        _, ss = be.colreduce(w, want_ss=True)
        out = -(chol.logdet() + n * LOG_2_PI + ss) / 2
        ctx.chol, ctx.w, ctx.x, ctx.terms = chol, w, x, terms
        ctx.nt, ctx.has_noise = nt, noise_vec is not None
        ctx.param_meta = [(p.device, p.dtype) for p in params]
        ctx.values = ([float(v) for v in variances], [float(s) for s in scales])
        return out

    @staticmethod
    def backward(ctx, grad_out):
        be = ops.get_backend()
        chol, w, x, terms, nt = ctx.chol, ctx.w, ctx.x, ctx.terms, ctx.nt
        n, C = w.shape
        if C > 8:
            raise NotImplementedError("backward through logpdf supports at most 8 columns of y")
        g = [float(v) for v in grad_out.reshape(-1).tolist()]    # host sync: C scalars
        # W = L^{-1} (lower triangular), K^{-1} = W^T W, A = K^{-1} r = W^T w
        W = chol.inverse_lower()                                              # N^3/3 flops
        kinv = be.gemm(W, W, a_kmajor=False, b_kmajor=False, lower_only=True, tri_k=True)   # N^3/3 flops
        alpha = torch.stack([be.colreduce(W, w[:, c], want_dot=True, want_ss=False)[0] for c in range(C)], dim=1)
        S, trace_g, diag_g = be.kmat_vjp(terms, x, kinv, alpha, g)
        variances, scales = ctx.values
        grads = []
        for t in range(nt):                                       # d/d variance_t
            dev, dt = ctx.param_meta[t]
            grads.append(S[t, 0].to(device=dev, dtype=dt))
        for t in range(nt):                                       # d/d scale_t
            dev, dt = ctx.param_meta[nt + t]
            grads.append((-2.0 * variances[t] / scales[t] * S[t, 1]).to(device=dev, dtype=dt))
        grad_r = -(alpha * grad_out.reshape(1, -1).to(alpha.dtype))
        grad_noise = diag_g if ctx.has_noise else None
        return (None, grad_r, grad_noise, None, *grads)


def needs_grad(tensor_terms, noise_vec, r):
    if not torch.is_grad_enabled():
        return False
    for _, v, s in tensor_terms:
        if (torch.is_tensor(v) and v.requires_grad) or (torch.is_tensor(s) and s.requires_grad):
            return True
    return (noise_vec is not None and noise_vec.requires_grad) or r.requires_grad


def gp_logpdf(kernel, x, noise_vec, r):
    """Differentiable log-density of ``r = y - m(x)`` under ``N(0, k(x) + diag(noise_vec) + eps I)``."""
    tt = kernel.tensor_terms()
    kinds = tuple(k for k, _, _ in tt)
    as_t = lambda v: v if torch.is_tensor(v) else torch.tensor(float(v), dtype=torch.float64)  # noqa: E731
    params = [as_t(v) for _, v, _ in tt] + [as_t(s) for _, _, s in tt]
    return _GPLogpdf.apply(x, r, noise_vec, kinds, *params)
