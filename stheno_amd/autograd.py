"""Differentiable ``f(x, noise).logpdf(y)`` -- row 8(f)-1 of SURVEY.md: the dominant use of
``stheno.torch`` is hyper-parameter learning (``readme_example13_optimisation_torch.py:47-53``).

Forward: the same HIP path as the plain logpdf (fused kernel matrix, in-place Cholesky, GEMV
sweep).  Backward, from the stored factor:

    G = d logpdf / dK = 1/2 (A diag(g) A^T - sum(g) K^{-1}),     A = K^{-1} (y - m)

``K^{-1} = W^T W`` with ``W = L^{-1}`` (blocked TRSM on the identity + lower SYRK, both on the
MFMA GEMM), then ONE pass over the lower triangle of ``K^{-1}`` (``gpk_kmat_vjp``) yields the
gradients w.r.t. every variance, length scale and the noise; ``d/d(y - m) = -A g``.
Gradients w.r.t. the inputs ``x`` (learnt input warps, latent inputs, per-dimension length scales --
``k.stretch(vector)`` divides the inputs) take one more pass: the explicit symmetric cotangent
``G`` is formed in the buffer of ``K^{-1}`` (a rank-C GEMM update) and ``gpk_kmat_vjp_dense`` reduces it
against ``dK_ij/dx_i``; both arguments of ``k(x, x)`` move with ``x``, hence the factor two.
"""
import math

import torch

from . import ops
from .matrix import LOG_2_PI, Chol, config

__all__ = ["gp_logpdf", "joint_logpdf", "sparse_elbo"]


def _cotangent(be, kinv_lower, alpha, g):
    """``G = d logpdf / dK = 1/2 (alpha diag(g) alpha^T - sum(g) K^{-1})`` as an explicit symmetric (n, n) matrix, formed in the
    buffer of the lower triangle of ``K^{-1}`` (consumed): ``alpha = K^{-1} r`` (n, C), ``g`` the C output cotangents."""
    gt = torch.as_tensor(g, dtype=alpha.dtype, device=alpha.device)
    G = be.symmetrize_(kinv_lower)
    return be.gemm(alpha * gt[None, :], alpha, a_kmajor=True, b_kmajor=True, alpha=0.5, beta=-0.5 * float(sum(g)), out=G)


def _grad_inputs(be, terms, x, G):
    """``d logpdf / dx`` (n, D) from the explicit cotangent: ``dx_i = 2 sum_j G_ij dk(x_i, x_j)/dx_i`` (both arguments of
    ``k(x, x)`` move with ``x``)."""
    if x.shape[-1] > 8:
        raise NotImplementedError("gradients with respect to the inputs are implemented for at most 8 input dimensions")
    _, _, gx = be.kmat_vjp_dense(terms, x, x, G, want_gradx=True)
    return 2.0 * gx


class _GPLogpdf(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, noise_vec, noise_mat, kinds, *params):
        """``params`` = variances then scales (scalar tensors, any device); ``noise_vec`` (n,)
        or None; ``noise_mat`` (n, n) dense noise covariance or None; ``r = y - mean`` (n, C).  Returns (C,)."""
        be = ops.get_backend()
        nt = len(kinds)
        variances, scales = params[:nt], params[nt:]
        terms = ops.KTerms([(k, float(v), float(s)) for k, v, s in zip(kinds, variances, scales)])
        n, C = r.shape
        k = be.kmat(terms, x, None, lower=True, diag_add=config.epsilon, diag_vec=noise_vec)
        if noise_mat is not None:
            k += noise_mat                                       # (only the lower triangle is read from here on)
        chol = Chol.factor_(k)
        w = chol.solve(r)                                        # L^{-1} r
        _, ss = be.colreduce(w, want_ss=True)
        out = -(chol.logdet() + n * LOG_2_PI + ss) / 2
        ctx.chol, ctx.w, ctx.x, ctx.terms = chol, w, x, terms
        ctx.nt, ctx.has_noise = nt, noise_vec is not None
        ctx.has_noise_mat = noise_mat is not None
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
        grad_x = grad_nm = None
        if ctx.needs_input_grad[0] or (ctx.has_noise_mat and ctx.needs_input_grad[3]):
            G = _cotangent(be, kinv, alpha, g)                   # explicit, in the buffer of K^{-1}
            if ctx.needs_input_grad[0]:
                grad_x = _grad_inputs(be, terms, x, G)
            if ctx.has_noise_mat and ctx.needs_input_grad[3]:
                grad_nm = G                                      # d/d noise matrix: the cotangent of K itself
        return (grad_x, grad_r, grad_noise, grad_nm, None, *grads)


class _GPLogpdfBatched(torch.autograd.Function):
    """The same for stheno's batched computation (``README.md:744-766``): ``x`` (B, N, D), ``r`` (B, N, 1), optional
    per-point noise (B, N), hyper-parameters SHARED by the B independent GPs -- learning over a batch of data sets
    (the configuration that shards over GPUs).  Forward = the batched HIP path (one launch sequence for all B);
    backward walks the batch: ``K_b^{-1}`` from the stored factor of entry b, one ``gpk_kmat_vjp`` pass, sums over b."""

    @staticmethod
    def forward(ctx, x, r, noise_vec, kinds, *params):
        be = ops.get_backend()
        nt = len(kinds)
        variances, scales = params[:nt], params[nt:]
        terms = ops.KTerms([(k, float(v), float(s)) for k, v, s in zip(kinds, variances, scales)])
        n = r.shape[-2]
        k = be.kmat(terms, x, None, lower=True, diag_add=config.epsilon, diag_vec=noise_vec)
        chol = Chol.factor_(k)
        w = chol.solve(r)                                        # (B, N, 1)
        _, ss = be.colreduce(w, want_ss=True)                    # (B, 1)
        out = -(chol.logdet() + n * LOG_2_PI + ss[..., 0]) / 2   # (B,)
        ctx.chol, ctx.w, ctx.x, ctx.terms = chol, w, x, terms
        ctx.nt, ctx.has_noise = nt, noise_vec is not None
        ctx.param_meta = [(p.device, p.dtype) for p in params]
        ctx.values = ([float(v) for v in variances], [float(s) for s in scales])
        return out

    @staticmethod
    def backward(ctx, grad_out):
        be = ops.get_backend()
        chol, w, x, terms, nt = ctx.chol, ctx.w, ctx.x, ctx.terms, ctx.nt
        B = x.shape[0]
        g_host = [float(v) for v in grad_out.reshape(-1).tolist()]            # host sync: B scalars
        S_tot = None
        grad_r = torch.empty_like(w)
        grad_noise = torch.empty(w.shape[:-1], dtype=w.dtype, device=w.device) if ctx.has_noise else None
        grad_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        for b in range(B):
            sl = lambda t: None if t is None else t[b:b + 1]      # noqa: E731
            cb = Chol(chol.l[b], sl(chol.dinv), sl(chol.info))                 # entry b as an unbatched factor (views)
            W = cb.inverse_lower()
            kinv = be.gemm(W, W, a_kmajor=False, b_kmajor=False, lower_only=True, tri_k=True)
            alpha = be.colreduce(W, w[b, :, 0], want_dot=True, want_ss=False)[0][:, None]
            S, _, diag_g = be.kmat_vjp(terms, x[b], kinv, alpha, [g_host[b]])
            S_tot = S if S_tot is None else S_tot + S
            grad_r[b] = -alpha * g_host[b]
            if grad_noise is not None:
                grad_noise[b] = diag_g
            if grad_x is not None:
                grad_x[b] = _grad_inputs(be, terms, x[b], _cotangent(be, kinv, alpha, [g_host[b]]))
        variances, scales = ctx.values
        grads = []
        for t in range(nt):
            dev, dt = ctx.param_meta[t]
            grads.append(S_tot[t, 0].to(device=dev, dtype=dt))
        for t in range(nt):
            dev, dt = ctx.param_meta[nt + t]
            grads.append((-2.0 * variances[t] / scales[t] * S_tot[t, 1]).to(device=dev, dtype=dt))
        return (grad_x, grad_r, grad_noise, None, *grads)


class _JointLogpdf(torch.autograd.Function):
    """Log-density of SEVERAL processes of one measure observed jointly (``measure.logpdf((f1(x1), y1), (f2(x2), y2))``,
    ``f(x).logpdf(y)`` of a product process): the variance is the block matrix of ``MultiOutputKernel``, block (i, j) =
    ``kernels[p_i, p_j](x_i, x_j)``, every block a sum of primitives.  Forward = the plain HIP path (blocks written into one
    buffer, factorised in place).  Backward: the explicit cotangent ``G = 1/2 (alpha diag(g) alpha^T - sum(g) K^{-1})`` once, then one
    ``gpk_kmat_vjp_dense`` pass per block of the lower block triangle over its view of ``G`` (off-diagonal blocks count twice: ``G`` and
    the block matrix are symmetric).  ``layout`` = [(i, j, number of terms)], ``params`` = the variances then scales of block
    after block: autograd carries them back to the user's leaves through whatever kernel algebra produced them."""

    @staticmethod
    def forward(ctx, r, noise_vec, build, parts, layout, kinds, *params):
        be = ops.get_backend()
        n = r.shape[0]
        k = build()                                              # lower block triangle + eps + noise on the diagonal
        chol = Chol.factor_(k)
        w = chol.solve(r)
        _, ss = be.colreduce(w, want_ss=True)
        out = -(chol.logdet() + n * LOG_2_PI + ss) / 2
        ctx.chol, ctx.w, ctx.parts, ctx.layout, ctx.kinds = chol, w, parts, layout, kinds
        ctx.has_noise = noise_vec is not None
        ctx.param_meta = [(p.device, p.dtype) for p in params]
        ctx.values = [float(p) for p in params]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        be = ops.get_backend()
        chol, w = ctx.chol, ctx.w
        g = [float(v) for v in grad_out.reshape(-1).tolist()]    # host sync: C scalars
        W = chol.inverse_lower()
        G = be.gemm(W, W, a_kmajor=False, b_kmajor=False, lower_only=True, tri_k=True)      # K^{-1}, lower triangle
        alpha = torch.stack([be.colreduce(W, w[:, c], want_dot=True, want_ss=False)[0] for c in range(w.shape[1])], dim=1)
        G = _cotangent(be, G, alpha, g)
        offs = [0]
        for _, xi in ctx.parts:
            offs.append(offs[-1] + xi.shape[0])
        grads, pos, kpos = [None] * len(ctx.values), 0, 0
        for (i, j, nt) in ctx.layout:
            vals = ctx.values[pos: pos + 2 * nt]
            variances, scales = vals[:nt], vals[nt:]
            terms = ops.KTerms([(kd, v, sc) for kd, v, sc in zip(ctx.kinds[kpos: kpos + nt], variances, scales)])
            gb = G[offs[i]: offs[i + 1], offs[j]: offs[j + 1]]
            S, _, _ = be.kmat_vjp_dense(terms, ctx.parts[i][1], ctx.parts[j][1], gb)
            wgt = 1.0 if i == j else 2.0
            for t in range(nt):
                dev, dt = ctx.param_meta[pos + t]
                grads[pos + t] = (wgt * S[t, 0]).to(device=dev, dtype=dt)
                dev, dt = ctx.param_meta[pos + nt + t]
                grads[pos + nt + t] = (wgt * -2.0 * variances[t] / scales[t] * S[t, 1]).to(device=dev, dtype=dt)
            pos += 2 * nt
            kpos += nt
        grad_r = -(alpha * grad_out.reshape(1, -1).to(alpha.dtype)) if ctx.needs_input_grad[0] else None
        grad_noise = torch.diagonal(G).clone() if ctx.has_noise else None
        return (grad_r, grad_noise, None, None, None, None, *grads)


def joint_logpdf(mok, x, noise_vec, r, eps):
    """Differentiable joint log-density under ``MultiOutputKernel`` ``mok`` at the multi-input ``x``; None when a block of the
    lower block triangle is not a sum of primitives with scalar hyper-parameters (the caller then refuses)."""
    kernels = mok.kernels
    parts = [(pid, xi) for pid, xi in mok._split(x)]
    if any((not torch.is_tensor(xi)) or xi.dim() != 2 or kernels[pid].num_outputs(xi) != xi.shape[0] for pid, xi in parts):
        return None                                   # batched inputs / nested product processes: not covered
    if any(xi.requires_grad for _, xi in parts):
        return None                                   # d/dx through the block matrix: not covered (the caller refuses)
    layout, kinds, variances_scales = [], [], []
    as_t = lambda v: v if torch.is_tensor(v) else torch.tensor(float(v), dtype=torch.float64)  # noqa: E731
    for i, (pi, _) in enumerate(parts):
        for j in range(i + 1):
            kern = kernels[pi] if i == j else kernels[pi, parts[j][0]]
            tt = kern.tensor_terms() if hasattr(kern, "tensor_terms") else None
            if tt is None:
                return None
            layout.append((i, j, len(tt)))
            kinds.extend(k for k, _, _ in tt)
            variances_scales.extend([as_t(v) for _, v, _ in tt] + [as_t(sc) for _, _, sc in tt])
    if not torch.is_grad_enabled() or not (any(p.requires_grad for p in variances_scales) or r.requires_grad
                                            or (noise_vec is not None and noise_vec.requires_grad)):
        return None

    def build():
        return mok.pairwise(x, None, lower=True, diag_add=eps, diag_vec=noise_vec)

    return _JointLogpdf.apply(r, noise_vec, build, parts, tuple(layout), tuple(kinds), *variances_scales)


def needs_grad(tensor_terms, noise_vec, r, x=None, noise_mat=None):
    if not torch.is_grad_enabled():
        return False
    if x is not None and torch.is_tensor(x) and x.requires_grad:
        return True
    if noise_mat is not None and noise_mat.requires_grad:
        return True
    for _, v, s in tensor_terms:
        if (torch.is_tensor(v) and v.requires_grad) or (torch.is_tensor(s) and s.requires_grad):
            return True
    return (noise_vec is not None and noise_vec.requires_grad) or r.requires_grad


def kernel_requires_grad(kernel, _depth=0):
    """Does any hyper-parameter reachable from ``kernel`` carry a gradient?  (Used to refuse -- loudly --
    the cases the differentiable paths do not cover, instead of returning a value cut off from the graph.)"""
    from . import kernels as _k

    if _depth > 16:
        return False
    tt = kernel.tensor_terms() if isinstance(kernel, _k.Kernel) else None
    if tt is not None:
        return any((torch.is_tensor(v) and v.requires_grad) or (torch.is_tensor(s) and s.requires_grad) for _, v, s in tt)
    if isinstance(kernel, _k.MultiOutputKernel):
        ks = kernel.kernels
        return any(kernel_requires_grad(ks[p], _depth + 1) for p in kernel.pids)
    for val in vars(kernel).values():
        if torch.is_tensor(val) and val.requires_grad:
            return True
        if isinstance(val, _k.Kernel) and kernel_requires_grad(val, _depth + 1):
            return True
    return False


def gp_logpdf(kernel, x, noise_vec, r, noise_mat=None):
    """Differentiable log-density of ``r = y - m(x)`` under ``N(0, k(x) + diag(noise_vec) + noise_mat + eps I)``."""
    tt = kernel.tensor_terms()
    kinds = tuple(k for k, _, _ in tt)
    as_t = lambda v: v if torch.is_tensor(v) else torch.tensor(float(v), dtype=torch.float64)  # noqa: E731
    params = [as_t(v) for _, v, _ in tt] + [as_t(s) for _, _, s in tt]
    if x.dim() == 3:
        return _GPLogpdfBatched.apply(x, r, noise_vec, kinds, *params)
    return _GPLogpdf.apply(x, r, noise_vec, noise_mat, kinds, *params)


# ---------------------------------------------------------------------------------------------
# Pseudo-point ELBO (VFE / FITC / DTC), stheno/model/observations.py:279-336, differentiable w.r.t. the
# kernel variances / length scales, the (diagonal) observation noise, the inducing inputs z and
# the residual r = y - m(x).
#
# With K_z = L L^T, V = L^{-1} K_zx, D = diag(noise), A = I + V D^{-1} V^T, c = A^{-1} V D^{-1} r,
# beta = r - V^T c (so Sigma^{-1} r = D^{-1} beta), tau = 1 (VFE) or 0 (DTC):
#     dELBO/dK_zx = L^{-T} [ (tau I - A^{-1}) V + c beta^T ] D^{-1}                  (M x N)
#     dELBO/dK_z  = -1/2 L^{-T} [ tau A - (tau + 1) I + A^{-1} + c c^T ] L^{-1}      (M x M)
#     dELBO/dD_j  = -1/2 [ 1/d_j - ((V^T A^{-1} V)_jj + beta_j^2 + tau (k_jj - q_jj)) / d_j^2 ]
#     dELBO/dk_jj = -tau / (2 d_j),      dELBO/dr = -D^{-1} beta
# The M x N cotangent costs ONE extra MFMA GEMM (2 M^2 N flops) on the stored, whitened and scaled
# V; gpk_kmat_vjp_dense then reads it once and returns the per-term sums, the column sums that
# give (V^T A^{-1} V)_jj without another TRSM, and d/dz.  Everything else is M x M.
# FITC = the DTC bound at the noise d + (k_jj - q_jj); its dependence on V through q_jj adds one M x N GEMM,
# one weighted M x M SYRK and a second pass of gpk_kmat_vjp_dense.
# ---------------------------------------------------------------------------------------------
class _SparseELBO(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, z, r, noise_vec, tau, kinds, *params):
        from .model.observations import _syrk_lower

        be = ops.get_backend()
        nt = len(kinds)
        variances, scales = params[:nt], params[nt:]
        terms = ops.KTerms([(k, float(v), float(s)) for k, v, s in zip(kinds, variances, scales)])
        n, m = x.shape[0], z.shape[0]
        d = noise_vec.detach()
        v = be.kmat(terms, z, x)                                               # K_zx
        chol_z = Chol.factor_(be.kmat(terms, z, None, lower=True, diag_add=config.epsilon))
        v = chol_z.solve_(v)                                                   # V
        _, q = be.colreduce(v, want_ss=True)
        corr = be.kdiag(terms, x) - q
        fitc = tau < 0
        if fitc:                     # FITC: the DTC bound with the noise d + (k_jj - q_jj)   (observations.py:312)
            d, tau = d + corr, 0.0
        s = torch.rsqrt(d)
        be.scale_cols_(v, s)                                                   # V D^{-1/2}
        a = torch.zeros((m, m), dtype=x.dtype, device=x.device)
        _syrk_lower(be, v, a)
        be.add_diag_(a, 1.0)
        p = be.gemv(v, r * s[:, None])
        a_fac = be.copy(a)
        if config.epsilon:
            be.add_diag_(a_fac, config.epsilon)
        chol_a = Chol.factor_(a_fac)
        u = chol_a.solve(p)
        _, uu = be.colreduce(u, want_ss=True)
        elbo = -0.5 * (torch.log(2 * math.pi * d).sum() + chol_a.logdet() + (r[:, 0] ** 2 / d).sum() - uu[0]
                       + tau * (corr / d).sum())
        ctx.saved = dict(x=x, z=z, r=r, d=d, s=s, v=v, q=q, corr=corr, a=a, u=u, chol_z=chol_z, chol_a=chol_a,
                         terms=terms, tau=tau, nt=nt, fitc=fitc)
        ctx.param_meta = [(p_.device, p_.dtype) for p_ in params]
        ctx.values = ([float(v_) for v_ in variances], [float(s_) for s_ in scales])
        ctx.kinds = kinds
        return elbo

    @staticmethod
    def backward(ctx, grad_out):
        be = ops.get_backend()
        sv = ctx.saved
        x, z, r, d, s, v, q, corr = sv["x"], sv["z"], sv["r"], sv["d"], sv["s"], sv["v"], sv["q"], sv["corr"]
        terms, tau, nt = sv["terms"], sv["tau"], sv["nt"]
        m = z.shape[0]
        eye = torch.eye(m, dtype=x.dtype, device=x.device)
        w_a = sv["chol_a"].inverse_lower()                                     # L_A^{-1}
        a_inv = be.symmetrize_(be.gemm(w_a, w_a, a_kmajor=False, b_kmajor=False, lower_only=True, tri_k=True))
        c = be.colreduce(w_a, sv["u"][:, 0], want_dot=True, want_ss=False)[0]  # A^{-1} p
        vtc = be.colreduce(v, c, want_dot=True, want_ss=False)[0] / s          # V^T c
        beta = r[:, 0] - vtc
        b = beta / d
        w_z = sv["chol_z"].inverse_lower()                                     # L^{-1}
        h = be.gemm(w_z, tau * eye - a_inv, a_kmajor=False, b_kmajor=True)     # L^{-T} (tau I - A^{-1})
        w = be.colreduce(w_z, c, want_dot=True, want_ss=False)[0]              # L^{-T} c
        g_k = be.gemm(h, v, a_kmajor=True, b_kmajor=False)                     # M x N, the one big GEMM
        need_z, need_x = ctx.needs_input_grad[1], ctx.needs_input_grad[0]
        fitc = sv["fitc"]
        s_k, colsum, gz_k = be.kmat_vjp_dense(terms, z, x, g_k, colscale=s, w=w, b=b, want_colsum=True,
                                              want_gradx=need_z and not fitc)
        vav = tau * q - d * colsum + vtc * beta                                # (V^T A^{-1} V)_jj
        g_d = -0.5 * (1.0 / d - (vav + beta * beta + tau * corr) / (d * d))
        a_full = be.symmetrize_(be.copy(sv["a"]))
        mid = tau * a_full - (tau + 1.0) * eye + a_inv + c[:, None] * c[None, :]
        if fitc:
            # the effective noise depends on V through q_jj = |V_j|^2:  dELBO/dV -= 2 V diag(g_d), i.e. the
            # cotangent of K_zx gets  -2 (L^{-T} V) diag(g_d)  and the K_z part  -2 V diag(g_d) V^T
            rr = be.gemm(w_z, v, a_kmajor=False, b_kmajor=False)               # L^{-T} V D^{-1/2}   (M x N)
            g_k.addcmul_(rr, (-2.0 * g_d / (s * s))[None, :])
            s_k, _, gz_k = be.kmat_vjp_dense(terms, z, x, g_k, colscale=s, w=w, b=b, want_gradx=need_z)
            rr.copy_(v)
            be.scale_cols_(rr, g_d * d)
            vgv = be.symmetrize_(be.gemm(rr, v, a_kmajor=True, b_kmajor=True, lower_only=True))
            mid = mid - 2.0 * vgv
            del rr
        gx_k = None
        if need_x:
            # d/dx through K_zx: the same reduction with the roles of the arguments swapped, on the explicit N x M
            # transpose of the effective cotangent  g_k diag(s) + w b^T  (one extra N x M buffer, backward only)
            be.scale_cols_(g_k, s)
            g_t = g_k.t().contiguous()
            del g_k
            g_t.addcmul_(b[:, None], w[None, :])
            _, _, gx_k = be.kmat_vjp_dense(terms, x, z, g_t, want_gradx=True)
            del g_t
        else:
            del g_k
        t1 = be.gemm(mid, w_z, a_kmajor=True, b_kmajor=False)
        g_kz = be.gemm(w_z, t1, a_kmajor=False, b_kmajor=False, alpha=-0.5)
        s_kz, _, gz_kz = be.kmat_vjp_dense(terms, z, z, g_kz, want_gradx=need_z)
        S = s_k + s_kz
        # through k(x_j, x_j) (VFE: trace term; FITC: the effective noise): stationary terms are constant there
        g_kd = g_d if fitc else -0.5 * tau / d
        variances, scales = ctx.values
        go = grad_out.to(x.dtype)
        grads_v, grads_s = [], []
        for t in range(nt):
            gv, gs = S[t, 0], -2.0 * variances[t] / scales[t] * S[t, 1]
            if tau or fitc:
                if ctx.kinds[t] == "linear":
                    xx = ((x * x).sum(-1) * g_kd).sum() / scales[t] ** 2
                    gv, gs = gv + xx, gs - 2.0 * variances[t] / scales[t] * xx
                    if need_x:      # k(x_j, x_j) = v |x_j|^2 / l^2 moves with x_j (stationary terms are constant there)
                        gx_k = gx_k + (2.0 * variances[t] / scales[t] ** 2) * g_kd[:, None] * x
                else:
                    gv = gv + g_kd.sum()
            grads_v.append(gv * go)
            grads_s.append(gs * go)
        grads = [g_.to(device=dev, dtype=dt) for g_, (dev, dt) in zip(grads_v + grads_s, ctx.param_meta)]
        grad_z = (gz_k + 2.0 * gz_kz) * go if need_z else None
        grad_x = gx_k * go if need_x else None
        grad_r = (-b * go)[:, None] if ctx.needs_input_grad[2] else None
        grad_noise = g_d * go if ctx.needs_input_grad[3] else None
        return (grad_x, grad_z, grad_r, grad_noise, None, None, *grads)


def elbo_needs_grad(tensor_terms, noise_vec, z, r, x=None):
    if not torch.is_grad_enabled():
        return False
    if x is not None and torch.is_tensor(x) and x.requires_grad:
        return True
    for _, v, s in tensor_terms:
        if (torch.is_tensor(v) and v.requires_grad) or (torch.is_tensor(s) and s.requires_grad):
            return True
    return bool(noise_vec.requires_grad or z.requires_grad or r.requires_grad)


def sparse_elbo(kernel, x, z, noise_vec, r, method):
    """Differentiable VFE / FITC / DTC bound for ``r = y - m(x)`` with inducing inputs ``z``."""
    if method not in ("vfe", "fitc", "dtc"):
        raise ValueError(f'Invalid approximation method "{method}".')
    tt = kernel.tensor_terms()
    kinds = tuple(k for k, _, _ in tt)
    as_t = lambda v: v if torch.is_tensor(v) else torch.tensor(float(v), dtype=torch.float64)  # noqa: E731
    params = [as_t(v) for _, v, _ in tt] + [as_t(s) for _, _, s in tt]
    return _SparseELBO.apply(x, z, r, noise_vec, {"vfe": 1.0, "dtc": 0.0, "fitc": -1.0}[method], kinds, *params)
