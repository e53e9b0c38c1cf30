"""Structured matrices for the GP path: ``Dense`` (with a cached Cholesky), ``Diagonal``,
``Zero`` -- the slice of the reference's ``matrix`` dependency (PyPI ``backends-matrix``)
that ``stheno/random.py`` and ``stheno/model/*.py`` touch -- plus ``KernelDense``, a
kernel matrix that can be factorised without ever being materialised next to its factor.

All heavy lifting goes through :mod:`stheno_amd.ops` (HIP kernels).
"""
import math
import threading

import torch

from . import ops

__all__ = ["AbstractMatrix", "Dense", "Diagonal", "Zero", "KernelDense", "FactoredDense", "Chol", "ChainChol", "config", "any_missing", "forget_scan", "deferred_checks"]


class _Config:
    """Global knobs.  ``epsilon`` mirrors ``lab``'s ``B.epsilon`` (README.md:820-831):
    the diagonal jitter added before every Cholesky."""

    epsilon = 1e-12
    #: raise ``torch.linalg.LinAlgError`` for non-positive-definite matrices (costs one
    #: 4-byte device->host read per factorisation)
    check_info = True
    #: scan `y` for NaN (missing observations, ``random.py:262-264`` / ``observations.py:73-74``) before a log-density / conditioning;
    #: the scan is a device reduction plus ONE host read of its flag -- set False when `y` is known to be complete and the host should
    #: not wait for the device there
    check_nan = True
    #: ``Normal.mean`` as the reference returns it (``README.md:58-68``: a ``Dense`` matrix that ``B.dense`` strips) instead of a plain
    #: ``torch.Tensor``.  Off by default: every caller on this path wants the tensor, and ``B.dense`` accepts both.
    mean_as_matrix = False
    #: outer block of the blocked Cholesky (0 = library default)
    potrf_nbo = 0
    #: single matrices of at least this order take the look-ahead factorisation (``gpk_potrf_la``); 0 disables it.  Below, the plain
    #: path (one pipelined launch per 1024-column panel, the rest of each trailing update riding along in the next panel's launch) is
    #: as fast or faster -- measured on MI355X at the end of round 3 (`profiles/r03_native_perf_plain_vs_lookahead.log`), plain / best
    #: look-ahead: fp64 N = 8192 5.21 / 5.48 ms, 10240 8.71 / 8.79, 12288 13.79 / 13.32, 16384 29.5 / 26.6; fp32 N = 8192 3.27 / 3.49,
    #: 12288 7.82 / 7.76, 16384 16.1 / 15.0 (15.5 with the 512-blocks fp32 uses)
    potrf_lookahead_from = 11264
    #: its outer block = the order of the explicitly inverted diagonal blocks: the per-dtype value from `potrf_lookahead_wide_from` on,
    #: 512 below (no such order is left with the defaults: at N = 12288 fp64, 1024-blocks 13.32 ms, 512-blocks 13.47)
    #: fp32 keeps 512-wide EXPLICIT inverses at every order: the rows below a diagonal block are multiplied by its explicit inverse, and
    #: in fp32 the posterior mean pays for the width of that inverse -- cfg3 at its full N = 32768 against fp64 on the device: mean error
    #: 1.00e-3 with 1024-wide inverses, 7.3e-4 with 512 (6.2e-4 with 256-block solves on top), 1.6e-3 with 2048
    #: (scripts/dev_fp32_fullsize_accuracy.py; north_star's fp32 bar is 1e-3).  The OUTER blocks are 1024 columns for both types
    #: (N = 32768 fp32: 97.6 ms with 1024-blocks, 102.5 with 512): with `potrf_lookahead_inv` narrower than the block, the rows below it
    #: are solved by block substitution over its column blocks (``gpk_potrf_la_split``).
    potrf_lookahead_nb = {torch.float64: 1024, torch.float32: 1024}
    potrf_lookahead_inv = {torch.float64: 1024, torch.float32: 512}
    potrf_lookahead_wide_from = 11264
    #: Posterior at `ns` new points when the observations' kernel matrix has not been factorised yet: from this order on (a multiple of
    #: 128, one unbatched matrix, a kernel that is a sum of primitives) K(x*, x) is written UNDER the kernel matrix in one buffer and
    #: carried through the factorisation (``gpk_potrf_rows``): it comes out as K(x*, x) L^{-T}, the transposed whitened cross-covariance,
    #: and the separate many-column triangular solve disappears.  0 disables it.
    posterior_rows_from = 2048
    posterior_rows_min_points = 64
    #: ... and the observations ``y - m(x)`` ride along as ONE more row, a right-hand side (``gpk_potrf_rows_rhs``): ``L^{-1} (y - m(x))``,
    #: which the posterior mean and the log-density need, comes out of the factorisation -- its ~30 dependent matrix-vector launches
    #: (0.7 ms at N = 16384) run beside the trailing updates instead of behind the factorisation.
    posterior_rows_rhs = True
    #: A log-density that is what factorises ``k(x) + noise`` hands its residual ``y - m(x)`` to the factorisation: a batch through
    #: ``gpk_potrf_rhs`` (fp32 batches of >= 64 matrices: solved inside the mixed-phase launches -- cfg4 16.59 -> 16.23 ms; every other
    #: batch factorises, then sweeps, inside the one native call), ONE matrix of at least ``posterior_rows_from`` rows as the only strip
    #: under the matrix (``gpk_potrf_rows_rhs``, the merged inverses kept for the posterior's solve that usually follows).
    logpdf_rhs = True
    #: Pseudo-point bounds (VFE / DTC) with many more observations than inducing points: build the cross-covariance transposed and padded
    #: to whole 128-tiles (``observations.py``), so that the M x N product runs in the GEMM kernel without bounds checks on two k-contiguous operands.
    pseudo_padded_transposed = True
    #: ... and that cross-covariance is built on a side stream BESIDE the factorisation and inversion of the pseudo-points' ``K_z``
    #: (chain-bound, 1.6 ms at M = 4096; the build is an HBM-bound write that depends on neither).
    pseudo_overlap_build = True
    #: fp32 kernel matrices that are regularised by the jitter alone (the pseudo-points' ``K_z``) are evaluated in fp64 and rounded
    #: once up to this order (``observations._kernel_matrix``); 0 disables it
    fp64_build_max_order = 4096
    #: Solves against an ILL-CONDITIONED factor get one step of iterative refinement against the factor
    #: (``x += L^{-1} (b - L x)``, :meth:`Chol._refined`): the blocked solves multiply by EXPLICIT inverses of 512- ... 2048-wide diagonal
    #: blocks, which is as accurate as substitution while those blocks are well conditioned and loses with their condition number
    #: beyond that -- measured against an 80-bit reference (``tests/golden/illcond_n1536.json``): posterior variance 500x, mean 20x
    #: the error of LAPACK's substitution at kappa ~ 5e8; one refinement step brings both back to what 128-wide blocks give.
    #: "auto": when the a-priori bound ``n * sum(variances) / (noise + epsilon)`` on the condition number
    #: (:meth:`KernelDense.cond_bound`; host-side, no device read) reaches ``refine_kappa`` of the dtype -- i.e. noise-free or
    #: nearly noise-free fp64 models (the README's regime, ``README.md:43-86``); never for the noisy benchmark configs.
    #: True / False: always / never (factors of kernel matrices; a ``Dense`` handed in by the user is never refined).
    refine_solves = "auto"
    refine_kappa = {torch.float64: 1e9}


config = _Config()

def _version_of(t):
    """The autograd version counter of ``t``, or ``None`` when nothing can vouch for its contents between two calls: inference
    tensors have no counter, and host memory can be aliased by NumPy / DLPack and written behind torch's back."""
    if not t.is_cuda or t.is_inference():
        return None
    return t._version


def any_missing(y):
    """Whether column 0 of the (N, 1) tensor ``y`` holds a NaN (a missing observation) -- a device reduction plus ONE host read
    (``random.py:262-264``, ``observations.py:73-74``).  The log-density and the conditioning on the same observations both ask, so
    the answer is remembered ON THE TENSOR OBJECT (an attribute: no global table, nothing outlives ``y``) together with the version
    counter it was computed at; any tracked in-place write invalidates it.  Not remembered at all for host tensors and inference
    tensors (``_version_of``).  Writes that bypass the counter on a device tensor (``y.data[...] = ...``, raw-pointer kernels) are
    the caller's to announce with :func:`forget_scan`; ``config.check_nan = False`` skips the scan altogether."""
    ver = _version_of(y)
    if ver is not None:
        seen = getattr(y, "_gpk_nan_scan", None)
        if seen is not None and seen[0] == ver:
            return seen[1]
    ans = bool(torch.isnan(y[:, 0]).any())
    if ver is not None:
        try:
            y._gpk_nan_scan = (ver, ans)
        except (AttributeError, RuntimeError):      # (a tensor subclass without a __dict__)
            pass
    return ans


def forget_scan(y):
    """Drop what :func:`any_missing` remembers about ``y`` (after a write torch did not see; ``bench.py`` calls it every step so that
    each step pays for its scan as a first call does)."""
    if getattr(y, "_gpk_nan_scan", None) is not None:
        y._gpk_nan_scan = None


class _DeferredState(threading.local):
    pending = None      # factors waiting for their check inside a `deferred_checks()` block of THIS thread


_deferred_state = _DeferredState()


class deferred_checks:
    """``with deferred_checks(): ...`` -- factorisations inside the block do not wait for their ``info`` word; all of them are checked
    when the block ends.  Reading ``info`` is a host read behind the factorisation: done right away, the device runs dry while the
    host comes back and enqueues what follows (0.3-0.6 ms of idle device per cfg2 eval, from the kernel trace); done at the end of the
    block, the work that depends on the factor (log-determinant, solves) is already queued behind it.  A failed factorisation still
    raises before anything computed from it is handed out, and the failed factor stays failed: whoever holds it in a cache
    (``Dense.chol()``) raises again on the next use instead of handing it out (``Chol.vetted``).  Nested blocks check at the end of
    the outermost one; a block left through an exception checks nothing, its factors are checked by their next user.  Per thread."""

    def __enter__(self):
        self._outer = _deferred_state.pending
        if self._outer is None:
            _deferred_state.pending = []
        return self

    def __exit__(self, exc_type, exc, tb):
        pending, _deferred_state.pending = _deferred_state.pending, self._outer
        if self._outer is None and exc_type is None:
            first = None
            for c in pending:                 # every one of them is looked at, the first failure is the one reported
                try:
                    c.check()
                except (RuntimeError, torch.linalg.LinAlgError) as e:
                    first = first or e
            if first is not None:
                raise first
        return False


def _solve_block(n, nrhs, fp64=True):
    """Size of the merged (explicitly inverted) diagonal blocks used by the triangular solves --
    measured on MI355X (profiles/r01_experiments.md): the few-right-hand-side GEMV sweep is
    launch-latency-bound (512-blocks); the many-right-hand-side sweep runs on the MFMA GEMM, where
    bigger blocks mean fewer, better-filled launches (fp64: 1024 from n = 8192, 2048 from n = 32768;
    in fp32 the posterior mean loses accuracy with the block size -- 3e-4 / 6e-4 / 1.1e-3 / 2.4e-3 relative
    at 128 / 512 / 1024 / 2048 for cfg3's kernel at N = 8192, and at its full N = 32768 7.3e-4 / 6.2e-4 with 512 / 256 -- so fp32
    takes 256 for the many-column solve); with far more
    right-hand sides than unknowns (pseudo-point path: M x N with N >> M; accuracy measured insensitive
    to the block size there) the whole factor is inverted once (M^3/3 flops) and the solve is ONE
    triangular GEMM."""
    if nrhs > 8 and nrhs >= 4 * n and n >= 1024:
        return min(4096, 1 << (n - 1).bit_length())
    if fp64 and n >= 32768:          # the GEMV sweep takes the same blocks: one merge serves logpdf and posterior
        return 2048
    if fp64 and n >= 8192:
        return 1024
    if n >= 2048 and (fp64 or nrhs <= 8):
        return 512       # (fp32: the single-column sweep of logpdf keeps the 512-blocks the look-ahead leaves behind)
    if n >= 512:
        return 256       # fp32 with many right-hand sides at every order: accuracy of the posterior mean (config.potrf_lookahead_nb;
                         # the recursive solve makes small blocks cheap)
    return 128


#: rows of the right-hand side's strip under the matrix (``gpk.h``: GPK_ROWS_RHS_STRIP)
RHS_STRIP = 64


class Chol:
    """Lower Cholesky factor ``L`` (possibly batched) with the diagonal-block inverses
    the HIP solve kernels use.  ``L``'s strict upper triangle is unspecified until
    :meth:`lower` is called."""

    def __init__(self, l, dinv, info):
        self.l = l
        self.dinv = dinv
        self.info = info
        self._checked = False
        self._error = None            # what `check` raised: a failed factor stays failed
        self._dinv_sb = {128: dinv}
        self._clean = False
        self.lookahead_nb = 0
        self.lookahead_sb = 0
        self.rows_under = 0
        self.rhs_rode = False         # the observations rode through the factorisation as a right-hand side (gpk_potrf_rows_rhs)
        self.refine = False           # one refinement step behind every solve (config.refine_solves; set by the matrix that owns the factor)
        self.refined = 0              # (how many solves took it; the tests ask)
        self._residuals = {}

    @classmethod
    def factor_(cls, a, rhs=None):
        """Factorise ``a`` (..., n, n; lower triangle read) IN PLACE.  ``rhs``: a contiguous (B, n) tensor, one right-hand side per
        matrix, overwritten by ``L^{-1} rhs`` along the way where the path can (``gpk_potrf_rhs``: the plain / batched path of a backend
        that says ``supports_potrf_rhs``); whether it did: ``chol.rhs_rode``."""
        be = ops.get_backend()
        n = a.shape[-1]
        nb = 0
        if a.dim() == 2 and config.potrf_lookahead_from and n >= config.potrf_lookahead_from and getattr(be, "name", "") == "hip":
            nb = config.potrf_lookahead_nb.get(a.dtype, 0)
            if nb > 512 and n < config.potrf_lookahead_wide_from:
                nb = 512
        if nb:
            sb = min(nb, config.potrf_lookahead_inv.get(a.dtype, nb))
            dinv, info, dnb = be.potrf_(a, config.potrf_nbo, lookahead_nb=nb, lookahead_sb=sb)
            c = cls(a, dinv, info)
            c.lookahead_nb = nb           # (which path ran; the tests ask)
            c.lookahead_sb = sb
            if sb == _solve_block(n, 1, a.dtype == torch.float64):
                c._dinv_sb[sb] = dnb      # the merged inverses the solves want come for free
            # (otherwise nobody would ever read them: n * sb elements are released here)
        elif rhs is not None and getattr(be, "supports_potrf_rhs", False):
            dinv, info = be.potrf_(a, config.potrf_nbo, rhs=rhs)
            c = cls(a, dinv, info)
            c.rhs_rode = True
        else:
            dinv, info = be.potrf_(a, config.potrf_nbo)
            c = cls(a, dinv, info)
        if config.check_info:
            if _deferred_state.pending is not None:
                _deferred_state.pending.append(c)       # checked when the enclosing `deferred_checks()` block ends
            else:
                c.check()
        return c

    @classmethod
    def factor_rows_(cls, buf, n, n_true=None, rhs_row=False, tail_inverses=False):
        """Factorise the leading ``n x n`` of ``buf`` (rows, n) IN PLACE, the rows under it riding along (``gpk_potrf_rows``): returns
        ``(chol, zt)`` with ``zt = buf[n:]`` holding ``buf[n:] L^{-T}`` afterwards.  The factor is a view of ``buf``.
        ``n_true < n``: the matrix is ``diag(A, I)`` with ``A`` of order ``n_true`` (``KernelDense.chol_with_rows`` pads to whole
        128-blocks): the factor and the rows handed back are the leading-``n_true`` views.
        ``rhs_row``: the last ``RHS_STRIP`` (64) rows of ``buf`` are a strip whose first row is one right-hand side ``b``, the others zero
        (``gpk_potrf_rows_rhs``); returns ``(chol, zt, w)`` with ``zt = buf[n:-64]`` and ``w = L^{-1} b`` as an (n_true, 1) view of that row.  The single-column solve was the only reader of the
        tail's merged inverses in the posterior-first flow, so they are not computed then unless ``tail_inverses`` asks (a later solve
        merges on demand, ``Chol._blocks``)."""
        be = ops.get_backend()
        n_true = n if n_true is None else n_true
        nb = sb = 0
        if config.potrf_lookahead_from and n >= config.potrf_lookahead_from:
            nb = config.potrf_lookahead_nb.get(buf.dtype, 0)
            if nb > 512 and n < config.potrf_lookahead_wide_from:
                nb = 512
            sb = min(nb, config.potrf_lookahead_inv.get(buf.dtype, nb)) if nb else 0
        if rhs_row:
            dinv, info, dnb = be.potrf_rows_(buf, lookahead_nb=nb, lookahead_sb=sb, rhs_row=True, tail_inverses=bool(tail_inverses))
        else:
            dinv, info, dnb = be.potrf_rows_(buf, lookahead_nb=nb, lookahead_sb=sb)
        c = cls(buf[:n_true, :n_true], dinv, info)
        if nb:
            c.lookahead_nb, c.lookahead_sb = nb, sb
            # (a padded order: the solves merge their own inverses -- the blocks the look-ahead leaves are those of the padded matrix)
            if dnb is not None and n_true == n and sb == _solve_block(n, 1, buf.dtype == torch.float64):
                c._dinv_sb[sb] = dnb
        c.rows_under = buf.shape[0] - n - (RHS_STRIP if rhs_row else 0)       # (which path ran; the tests ask)
        c.rhs_rode = bool(rhs_row)
        if config.check_info:
            if _deferred_state.pending is not None:
                _deferred_state.pending.append(c)
            else:
                c.check()
        if rhs_row:
            return c, buf[n:-RHS_STRIP, :n_true], buf[-RHS_STRIP, :n_true].unsqueeze(-1)
        return c, buf[n:, :n_true]

    def remember_residual(self, source, w):
        """File ``w = L^{-1} (y - 0)`` under the data tensor ``y`` it came from (see :meth:`solve_residual`): the log-density of the same
        observations then finds it."""
        ver = _version_of(source) if torch.is_tensor(source) else None
        if ver is not None:
            if len(self._residuals) >= 2:
                self._residuals.clear()
            self._residuals[id(source)] = (source, ver, w)

    def check(self):
        if self._error is not None:
            raise self._error
        if not self._checked:
            bad = 0
            if self.info.numel() == 1:
                bad = int(self.info.item())       # (one matrix: the word itself -- no reduction, no stacking: two launches less in front of the read)
            elif self.info.numel():
                lo, hi = torch.stack(tuple(torch.aminmax(self.info))).tolist()       # (one host read)
                bad = lo if lo < 0 else hi
            self._checked = True
            if bad < 0:
                self._error = RuntimeError("cholesky: the workgroups of a multi-workgroup factorisation kernel stopped waiting for each other "
                                           "(gpk.h: info = -1) or ran on another XCD than their matrix is pinned to (info = -2); info = %d, the "
                                           "factor is unusable -- please report this" % bad)
            elif bad != 0:
                self._error = torch.linalg.LinAlgError(
                    f"cholesky: the leading minor of order {bad} is not positive-definite "
                    "(increase B.epsilon or the noise)"
                )
            if self._error is not None:
                raise self._error
        return self

    def vetted(self):
        """This factor, for a holder that hands it out of a cache: raises what its check raised; checks it now if nobody has yet and
        no ``deferred_checks()`` block of this thread is going to (the block it was made in was left through an exception)."""
        if self._error is not None:
            raise self._error
        if not self._checked and config.check_info:
            pending = _deferred_state.pending
            if pending is None or not any(c is self for c in pending):
                self.check()
        return self

    @property
    def n(self):
        return self.l.shape[-1]

    def logdet(self):
        return ops.get_backend().logdet_chol(self.l)

    def _blocks(self, nrhs):
        n = self.n
        # Batched factors keep the 128-blocks (the merge loops over the batch on the host and the
        # batch already fills the GPU); a single large factor always uses the merged blocks, also
        # for the few-rhs GEMV sweep: 4x fewer (launch-latency-bound) steps for a 0.2 ms merge.
        sb = 128 if self.l.dim() > 2 else _solve_block(n, nrhs, self.l.dtype == torch.float64)
        if sb not in self._dinv_sb:
            self._dinv_sb[sb] = ops.get_backend().trtri_merge(self.l, self.dinv, sb)
        return sb, self._dinv_sb[sb]

    def _apply_full_inverse(self, b):
        """``L^{-1} b`` as ONE triangular GEMM into a fresh buffer when the whole factor has been inverted
        (far more right-hand sides than unknowns, see ``_solve_block``); ``None`` if that does not apply."""
        if self.l.dim() != 2 or b.dim() != 2 or b.shape[-1] <= 8:
            return None
        sb, dsb = self._blocks(b.shape[-1])
        if dsb is None or sb < self.n:
            return None
        return ops.get_backend().gemm(dsb[0, 0][: self.n, : self.n], b, a_kmajor=True, b_kmajor=False, tri_k_lower=True)

    def solves_by_full_inverse(self, nrhs):
        """Whether a solve against ``nrhs`` right-hand sides is ONE triangular GEMM with the explicitly inverted factor (``_solve_block``:
        far more right-hand sides than unknowns) AND the backend can fold column statistics into it (:meth:`solve_scaled`)."""
        if self.l.dim() != 2 or nrhs <= 8 or not hasattr(ops.get_backend(), "gemm_colscale"):
            return False
        sb, dsb = self._blocks(nrhs)          # (the merged inverse is computed here at the latest: the solve needs it anyway)
        return dsb is not None and sb >= self.n

    def solve_scaled(self, b, colscale, want_colss, b_kmajor=False):
        """``(L^{-1} b) diag(colscale)`` and (``want_colss``) the column sums of squares of ``L^{-1} b``, from ONE triangular GEMM with
        both folded into its store -- when the whole factor has been inverted (``_solve_block``: far more right-hand sides than
        unknowns, the pseudo-point path).  ``None`` when that does not apply (the caller takes the separate passes).
        ``b_kmajor``: ``b`` is handed over TRANSPOSED, (nrhs, n) -- both operands of the product k-contiguous."""
        nrhs = b.shape[-2] if b_kmajor else b.shape[-1]
        if b.dim() != 2 or not self.solves_by_full_inverse(nrhs):
            return None
        sb, dsb = self._blocks(nrhs)
        if dsb is None or sb < self.n:
            return None
        return ops.get_backend().gemm_colscale(dsb[0, 0][: self.n, : self.n], b, colscale, want_colss=want_colss, a_kmajor=True,
                                               b_kmajor=b_kmajor, tri_k_lower=True)

    def _solve_once_(self, b):
        out = self._apply_full_inverse(b)
        if out is not None:
            return out
        sb, dsb = self._blocks(b.shape[-1])
        return ops.get_backend().tri_solve_(self.l, dsb, sb, b)

    def _refines(self, b):
        return self.refine and self.l.dim() == 2 and b.dim() == 2 and b.shape[-1] > 0

    def _refined(self, b0, x):
        """``x + L^{-1} (b0 - L x)``: one step of iterative refinement of ``x ~ L^{-1} b0`` against the factor (``config.refine_solves``).
        ``b0`` (a private copy of the right-hand side) is used up, ``x`` is updated in place."""
        be = ops.get_backend()
        l = self.lower()                  # (the residual product reads whole tiles / rows of L: zeros above the diagonal)
        if x.shape[-1] <= 8:
            r = be.gemv(l, x, alpha=-1.0, beta=1.0, out=b0)
        else:
            r = be.gemm(l, x, a_kmajor=True, b_kmajor=False, alpha=-1.0, beta=1.0, out=b0, tri_k_lower=True)
        self.refined += 1
        return x.add_(self._solve_once_(r))

    def refine_rows_(self, zt, kxs):
        """The rows that rode through the factorisation, ``zt ~ kxs L^{-T}`` (ns, n), refined in place by one step against the factor:
        ``zt += ((kxs - zt L^T) L^{-T})``; ``kxs`` (a fresh evaluation of the cross-covariance, (ns, n)) is used up."""
        be = ops.get_backend()
        r = be.gemm(zt, self.lower(), a_kmajor=True, b_kmajor=True, alpha=-1.0, beta=1.0, out=kxs)
        dz = self._solve_once_(r.transpose(-1, -2).contiguous())
        self.refined += 1
        zt.add_(dz.transpose(-1, -2))
        return zt

    def solve_(self, b):
        """``L^{-1} b``, overwriting ``b`` where possible (``b``: (..., n, nrhs), unit inner stride).
        Use the RETURN value: the single-GEMM case writes a fresh buffer and leaves ``b`` as it was."""
        if self._refines(b):
            b0 = b.clone()
            return self._refined(b0, self._solve_once_(b))
        return self._solve_once_(b)

    def solve(self, b):
        """``L^{-1} b`` as a new tensor."""
        if b.shape[-2] != self.n:
            raise ValueError(f"right-hand side has {b.shape[-2]} rows, the factor has order {self.n}")
        lb, bb = tuple(self.l.shape[:-2]), tuple(b.shape[:-2])
        if lb and bb and lb != bb:
            raise ValueError(f"batch shapes {lb} and {bb} do not match")
        out = self._apply_full_inverse(b)
        if out is None:
            out = b.expand((lb or bb) + tuple(b.shape[-2:])).clone(memory_format=torch.contiguous_format)
            out = self._solve_once_(out)
        if self._refines(b):
            out = self._refined(b.clone(memory_format=torch.contiguous_format), out)
        return out

    def iqf_diag(self, b, source=None):
        """Column-wise ``|L^{-1} b|^2``: (..., nrhs).  ``source``: see :meth:`solve_residual`."""
        v = self.solve_residual(b, source)
        _, ss = ops.get_backend().colreduce(v, want_ss=True)
        return ss

    def solve_residual(self, r, source=None):
        """``L^{-1} r`` for a few columns, remembered when the caller can name where ``r`` came from: ``source`` = the data
        tensor ``y`` when ``r`` is ``y`` minus a ZERO mean (``None`` otherwise).  The log-density (``random.py:276``) and the posterior
        mean (``observations.py:161-168``) of the same observations both need ``L^{-1} y`` -- the reference solves twice; here the
        second asker gets the first one's result (0.46 ms of a cfg2 eval).  The key is the tensor OBJECT ``y`` (the entry holds it, so
        its id cannot be recycled) at its version counter; no memory for tensors nothing vouches for (``_version_of``: host memory,
        inference tensors)."""
        for key in (source, r):            # (what rode through the factorisation is filed under `r` itself too: any mean, any batch)
            if torch.is_tensor(key) and self._residuals:
                hit = self._residuals.get(id(key))
                if hit is not None and hit[0] is key and hit[1] == _version_of(key) and tuple(hit[2].shape) == tuple(r.shape) \
                        and not (torch.is_grad_enabled() and key.requires_grad):
                    return hit[2]
        if source is None or not torch.is_tensor(source) or r.dim() != 2 or r.shape[-1] > 8 or torch.is_grad_enabled() and source.requires_grad:
            return self.solve(r)
        ver = _version_of(source)
        if ver is None:
            return self.solve(r)
        hit = self._residuals.get(id(source))
        if hit is not None and hit[0] is source and hit[1] == ver:
            return hit[2]
        out = self.solve(r)
        if len(self._residuals) >= 2:
            self._residuals.clear()
        self._residuals[id(source)] = (source, ver, out)
        return out

    def inverse_lower(self):
        """``W = L^{-1}`` as a full lower-triangular (n, n) matrix (unbatched; N^3/3 flops)."""
        if self.l.dim() != 2:
            raise NotImplementedError("inverse_lower is implemented for unbatched factors")
        sb, dsb = self._blocks(self.n)
        return ops.get_backend().trtri(self.l, dsb, sb)

    def lower(self):
        """The clean lower-triangular factor (zeros above the diagonal)."""
        if not self._clean:
            ops.get_backend().tril_(self.l)
            self._clean = True
        return self.l


class ChainChol:
    """Cholesky factor kept in product form ``L = L_1 L_2 ...`` (every ``L_i`` lower
    triangular, so the product is the Cholesky factor of ``L L^T``).  Used for the
    pseudo-point matrix ``L_z A L_z^T = (L_z L_A)(L_z L_A)^T`` (``observations.py:323``):
    re-factorising that product numerically squares the condition number for nothing."""

    def __init__(self, *chols):
        self.chols = chols

    @property
    def n(self):
        return self.chols[0].n

    def check(self):
        for c in self.chols:
            c.check()
        return self

    def vetted(self):
        for c in self.chols:
            c.vetted()
        return self

    def logdet(self):
        out = self.chols[0].logdet()
        for c in self.chols[1:]:
            out = out + c.logdet()
        return out

    def solve(self, b):
        out = self.chols[0].solve(b)
        for c in self.chols[1:]:
            out = c.solve_(out)
        return out

    def iqf_diag(self, b):
        _, ss = ops.get_backend().colreduce(self.solve(b), want_ss=True)
        return ss


class AbstractMatrix:
    """Base class of the structured matrices."""

    @property
    def dtype(self):
        raise NotImplementedError

    @property
    def device(self):
        raise NotImplementedError

    @property
    def shape(self):
        raise NotImplementedError

    def __radd__(self, other):
        return self.__add__(other)


class Zero(AbstractMatrix):
    """``n x m`` zero matrix (``fdd.py:26``: no noise)."""

    def __init__(self, dtype, rows, cols, device=None, batch=()):
        self._dtype, self.rows, self.cols, self._device, self.batch = dtype, rows, cols, device, tuple(batch)

    dtype = property(lambda self: self._dtype)
    device = property(lambda self: self._device)
    shape = property(lambda self: self.batch + (self.rows, self.cols))

    def dense(self):
        return torch.zeros(self.shape, dtype=self._dtype, device=self._device)

    def diag(self):
        return torch.zeros(self.batch + (min(self.rows, self.cols),), dtype=self._dtype, device=self._device)

    def __add__(self, other):
        if isinstance(other, (int, float)) and other == 0:
            return self
        return other

    def __eq__(self, other):
        return isinstance(other, Zero) and self.shape == other.shape

    def __hash__(self):
        return id(self)

    def __repr__(self):
        return f"<zero matrix: shape={self.rows}x{self.cols}, dtype={self._dtype}>"


class Diagonal(AbstractMatrix):
    """Diagonal matrix given by its diagonal (..., n)."""

    def __init__(self, diag):
        self._diag = diag

    dtype = property(lambda self: self._diag.dtype)
    device = property(lambda self: self._diag.device)
    shape = property(lambda self: tuple(self._diag.shape) + (self._diag.shape[-1],))

    def diag(self):
        return self._diag

    def dense(self):
        return torch.diag_embed(self._diag)

    def logdet(self):
        return torch.log(self._diag).sum(-1)

    def iqf_diag(self, b):
        return (b * b / self._diag[..., :, None]).sum(-2)

    def __add__(self, other):
        if isinstance(other, Zero):
            return self
        if isinstance(other, Diagonal):
            return Diagonal(self._diag + other._diag)
        if isinstance(other, (int, float)):
            if other == 0:
                return self
            raise TypeError("adding a non-zero scalar to a Diagonal densifies it; add to .dense() instead")
        if isinstance(other, Dense):
            return other + self
        return NotImplemented

    def __sub__(self, other):
        if isinstance(other, Diagonal):
            return Diagonal(self._diag - other._diag)
        return NotImplemented

    def __mul__(self, s):
        return Diagonal(self._diag * s)

    __rmul__ = __mul__

    def __repr__(self):
        return f"<diagonal matrix: shape={self.shape}, dtype={self.dtype}>"


class Dense(AbstractMatrix):
    """Dense (symmetric when used as a variance) matrix with a cached Cholesky factor --
    the behaviour of ``matrix.Dense`` the reference relies on: ``B.logdet`` and
    ``B.iqf_diag`` in ``random.py:274-276`` share one factorisation."""

    def __init__(self, mat):
        self._mat = mat
        self._chol = None

    # materialisation ---------------------------------------------------------
    @property
    def mat(self):
        return self._mat

    dtype = property(lambda self: self.mat.dtype)
    device = property(lambda self: self.mat.device)
    shape = property(lambda self: tuple(self.mat.shape))

    def dense(self):
        return self.mat

    def diag(self):
        return torch.diagonal(self.mat, dim1=-2, dim2=-1)

    # Cholesky ----------------------------------------------------------------
    def chol(self):
        """``chol(self + epsilon * I)``, computed once (a copy is factorised)."""
        if self._chol is None:
            be = ops.get_backend()
            a = be.copy(self.mat)
            if config.epsilon:
                be.add_diag_(a, config.epsilon)
            self._chol = Chol.factor_(a)
            return self._chol
        return self._chol.vetted()

    def logdet(self):
        return self.chol().logdet()

    def iqf_diag(self, b, source=None):
        return self.chol().iqf_diag(b, source) if source is not None else self.chol().iqf_diag(b)

    # algebra -----------------------------------------------------------------
    def __add__(self, other):
        be = ops.get_backend()
        if isinstance(other, Zero) or (isinstance(other, (int, float)) and other == 0):
            return self
        if isinstance(other, Diagonal):
            return Dense(be.add_diag_(be.copy(self.mat), 0.0, other.diag()))
        if isinstance(other, Dense):
            return Dense(self.mat + other.mat)
        if torch.is_tensor(other):
            return Dense(self.mat + other)
        return NotImplemented

    def __repr__(self):
        return f"<dense matrix: shape={self.shape}, dtype={self.dtype}>"


class KernelDense(Dense):
    """``k(x) + noise`` as a lazily materialised ``Dense`` (``fdd.py:79``,
    ``observations.py:139,286``).

    If only the Cholesky factor is needed (logpdf, conditioning), the kernel matrix is
    built lower-triangle-only straight into the buffer that is then factorised in
    place: ``K`` and ``L`` are never both resident (N = 32768 fp32 is 4.3 GB per copy).
    """

    def __init__(self, kernel, x, noise):
        super().__init__(None)
        self.kernel, self.x, self.noise = kernel, x, noise

    def _noise_parts(self):
        noise = self.noise
        if noise is None or isinstance(noise, Zero):
            return 0.0, None, None
        if isinstance(noise, Diagonal):
            return 0.0, noise.diag(), None
        if isinstance(noise, Dense):
            return 0.0, None, noise.mat
        raise TypeError(f"unsupported noise type {type(noise).__name__}")

    def differentiable_noise(self):
        """The noise as a vector (n,) -- (B, n) for batched inputs -- or ``None`` if it is diagonal, else ``NotImplemented``."""
        noise = self.noise
        if noise is None or isinstance(noise, Zero):
            return None
        if isinstance(noise, Diagonal) and noise.diag().dim() in (1, 2):
            return noise.diag()
        return NotImplemented

    def _build(self, lower, jitter):
        _, dvec, dense_noise = self._noise_parts()
        out = self.kernel.pairwise(self.x, None, lower=lower and dense_noise is None, diag_add=jitter, diag_vec=dvec)
        if dense_noise is not None:
            out = out + dense_noise
        return out

    @property
    def mat(self):
        if self._mat is None:
            self._mat = self._build(lower=False, jitter=0.0)
        return self._mat

    @property
    def dtype(self):
        return self.x.dtype

    @property
    def device(self):
        return self.x.device

    @property
    def shape(self):
        n = self.kernel.num_outputs(self.x)
        return tuple(self.x.shape[:-2]) + (n, n)

    def diag(self):
        if self._mat is not None:
            return torch.diagonal(self._mat, dim1=-2, dim2=-1)
        d = self.kernel.elwise(self.x)[..., 0]
        _, dvec, dense_noise = self._noise_parts()
        if dvec is not None:
            d = d + dvec
        if dense_noise is not None:
            d = d + torch.diagonal(dense_noise, dim1=-2, dim2=-1)
        return d

    def chol(self):
        if self._chol is None:
            if self._mat is not None:
                return super().chol()
            a = self._build(lower=True, jitter=config.epsilon)
            self._chol = Chol.factor_(a)
            self._chol.refine = self.wants_refinement()
            return self._chol
        return self._chol.vetted()

    def chol_with_rhs(self, r, source=None):
        """:meth:`chol`, with the residual ``r`` (..., n, 1) of a log-density solved along if THIS call is what factorises the matrix
        (``config.logpdf_rhs``, batched matrices): ``L^{-1} r`` is filed with the factor under ``r`` (and under the data tensor
        ``source`` it equals for a zero mean), where :meth:`Chol.solve_residual` finds it."""
        if (self._chol is not None or self._mat is not None or not config.logpdf_rhs or not torch.is_tensor(r) or r.dim() not in (2, 3) or r.shape[-1] != 1
                or (torch.is_grad_enabled() and r.requires_grad) or self.wants_refinement()):
            return self.chol()
        if r.dim() == 2:
            # ONE matrix of an order the rows-under-the-matrix path takes: the residual is the only "row" (its strip), the solve that
            # usually follows a log-density (a posterior) finds its merged inverses where the look-ahead leaves them
            be = ops.get_backend()
            n = self.x.shape[-2] if self.x.dim() == 2 else -1
            npad = -(-n // 128) * 128
            if (n < max(config.posterior_rows_from, 1) or not config.posterior_rows_from or not hasattr(be, "potrf_rows_") or self.x.requires_grad
                    or n != self.kernel.num_outputs(self.x) or tuple(r.shape) != (n, 1) or npad > 64 * 512 or self._noise_parts()[2] is not None
                    or not rows_panels_fit(npad, RHS_STRIP, self.x.element_size(), bool(config.potrf_lookahead_from) and npad >= config.potrf_lookahead_from)):
                return self.chol()
            got = self.chol_with_rows(None, None, rhs=r, tail_inverses=True)
            c = got[0]
            if len(got) == 3:
                c.remember_residual(r, got[2])
                if source is not None and source is not r:
                    c.remember_residual(source, got[2])
            return c
        a = self._build(lower=True, jitter=config.epsilon)
        if a.dim() != 3 or tuple(r.shape[:-1]) != tuple(a.shape[:-1]) or r.dtype != a.dtype or r.device != a.device:
            self._chol = Chol.factor_(a)
            return self._chol
        rhs = r[..., 0].clone(memory_format=torch.contiguous_format)
        self._chol = c = Chol.factor_(a, rhs=rhs)
        if c.rhs_rode:
            w = rhs.unsqueeze(-1)
            c.remember_residual(r, w)
            if source is not None and source is not r:
                c.remember_residual(source, w)
        return c

    def cond_bound(self):
        """An a-priori upper bound on the condition number of ``k(x) + noise + epsilon I`` from what the HOST knows (no device read):
        ``n * sum of the stationary terms' variances / (noise + epsilon)`` -- ``lambda_max <= trace``, ``lambda_min >=`` the smallest
        diagonal addition.  ``None`` when the host does not know the pieces: a kernel that is no sum of primitives, noise given as a
        tensor (its smallest entry lives on the device), dense noise.  Non-stationary terms (``Linear``) are left out of the trace:
        they add at most ``D`` large eigenvalues, the bound is a trigger for `config.refine_solves`, not a guarantee."""
        terms = self.kernel.terms() if hasattr(self.kernel, "terms") else None
        if not terms or self.x.dim() != 2:
            return None
        noise = self.noise
        if noise is None or isinstance(noise, Zero):
            floor = 0.0
        elif isinstance(noise, Diagonal) and getattr(noise, "host_min", None) is not None:
            floor = float(noise.host_min)
        else:
            return None
        floor += float(config.epsilon)
        if not floor > 0.0:
            return math.inf
        trace = sum(float(var) for kind, var, _ in terms if kind != "linear")
        return self.x.shape[-2] * max(trace, 0.0) / floor

    def wants_refinement(self):
        mode = config.refine_solves
        if mode is True or mode is False:
            return mode
        limit = config.refine_kappa.get(self.dtype)
        if limit is None or self.x.dim() != 2:
            return False
        bound = self.cond_bound()
        return bound is not None and bound >= limit

    def can_factor_with_rows(self, ns):
        """Whether :meth:`chol_with_rows` applies: nothing factorised or materialised yet, ONE matrix of an order the native path
        takes (``config.posterior_rows_from``, a multiple of 128), diagonal noise, the HIP backend."""
        if self._chol is not None or self._mat is not None or not config.posterior_rows_from:
            return False
        be = ops.get_backend()
        if not hasattr(be, "potrf_rows_") or self.x.dim() != 2 or self.x.requires_grad:
            return False
        n = self.kernel.num_outputs(self.x)
        # ns <= n: the cached factor is a view of the (n + ns, n) buffer, so the whitened rows live as long as the factor does --
        # at most twice the factor's memory (with ns up to 4 n it was five times)
        if n != self.x.shape[-2] or n < config.posterior_rows_from or ns < config.posterior_rows_min_points or ns > n:
            return False
        npad = -(-n // 128) * 128      # (round 6: any order -- the native path wants whole 128-blocks, `chol_with_rows` pads with the identity)
        if npad > 64 * 512:         # (the look-ahead's column groups are a 64-bit mask: at most 64 outer blocks of >= 512 columns)
            return False
        if not rows_panels_fit(npad, ns + RHS_STRIP, self.x.element_size(), bool(config.potrf_lookahead_from) and npad >= config.potrf_lookahead_from):
            return False
        return self._noise_parts()[2] is None

    def chol_with_rows(self, k_cross, xs, rhs=None, tail_inverses=False):
        """The factor AND ``k_cross(xs, x) L^{-T}`` (ns, n) from one factorisation: the kernel matrix is built in the first ``n`` rows
        of an (n + ns, n) buffer, the cross-covariance under it, and ``gpk_potrf_rows`` carries those rows through its panel solves
        and trailing updates.  Replaces ``cholesky`` + ``solve(L, K_zx)`` of mlkernels' PosteriorKernel (observations.py:148-168).

        An order that is no multiple of 128 (round 6) is PADDED to one: ``diag(K, I)`` in an (npad + ns, npad) buffer, zero columns
        under the identity -- its factor is ``diag(L, I)``, the rows come out as ``[K* L^{-T}, 0]``; the factor and the whitened rows
        handed on are the leading-``n`` views of that buffer (every consumer takes a leading dimension).

        ``rhs`` (n, 1): one right-hand side that rides along as the last row (``config.posterior_rows_rhs``); returns
        ``(chol, zt, w)`` with ``w = L^{-1} rhs`` then.  ``k_cross = None``: no rows but the right-hand side's (a log-density that
        factorises: :meth:`chol_with_rhs`); ``tail_inverses``: with a right-hand side, compute the merged inverses of the look-ahead's
        tail all the same (a many-column solve is expected to follow)."""
        n, ns = self.x.shape[-2], (xs.shape[-2] if k_cross is not None else 0)
        npad = -(-n // 128) * 128
        refine = self.wants_refinement()
        if rhs is not None and (refine or tuple(rhs.shape) != (n, 1) or rhs.dtype != self.x.dtype or rhs.device != self.x.device):
            rhs = None           # (a refined factor solves for it separately: Chol.solve refines, the row would not be)
        nr = RHS_STRIP if rhs is not None else 0
        buf = torch.empty((npad + ns + nr, npad), dtype=self.x.dtype, device=self.x.device)
        _, dvec, _ = self._noise_parts()
        top = self.kernel.pairwise(self.x, None, lower=True, diag_add=config.epsilon, diag_vec=dvec, out=buf[:n, :n])
        low = k_cross.pairwise(xs, self.x, out=buf[npad:npad + ns, :n]) if k_cross is not None else None
        # (a kernel that ignores `out=` would leave the buffer uninitialised and the factorisation would whiten garbage)
        if top.data_ptr() != buf.data_ptr() or (low is not None and low.data_ptr() != buf[npad:].data_ptr()):
            raise RuntimeError(f"{type(self.kernel).__name__} / {type(k_cross).__name__}.pairwise did not write into `out`")
        if nr:
            buf[npad + ns, :n].copy_(rhs[:, 0])
            buf[npad + ns + 1:].zero_()
        if npad > n:
            buf[n:npad].zero_()
            buf[n:npad, n:npad].fill_diagonal_(1.0)
            buf[npad:, n:].zero_()
        if nr:
            self._chol, zt, w = Chol.factor_rows_(buf, npad, n, rhs_row=True, tail_inverses=tail_inverses)
            return self._chol, zt, w
        if k_cross is None:          # (nothing to carry: the plain factorisation)
            self._chol = Chol.factor_(buf[:n, :n] if npad == n else buf[:n, :n].contiguous())
            self._chol.refine = refine
            return self._chol, buf[npad:, :n]
        self._chol, zt = Chol.factor_rows_(buf, npad, n)
        self._chol.refine = refine
        if refine:
            self._chol.refine_rows_(zt, k_cross.pairwise(xs, self.x))
        return self._chol, zt


def rows_panels_fit(n, ns, itemsize, lookahead):
    """Whether every pipelined panel ``gpk_potrf_rows`` would launch for an order ``n`` with ``ns`` rows under the matrix finds room
    for its control words -- the native limit (``gpk_potrf.hip:potrf_panel_pipe``, ``gpk_potrf_pipe.hpp:pipe_ctrl_words``), mirrored
    so that the host never picks a shape the library refuses (``GPK_ERR_ARG(2)`` where the separate solve would have worked): a panel
    of ``npb`` 128-column blocks over ``m`` rows keeps ``96 + 2 ceil(m / 64) npb`` 4-byte words in ONE 128 x 128 slot of ``dinv``.
    Orders up to 4096 are one panel over all ``n + ns`` rows; above, 1024-column panels, the tallest one over ``n + ns`` rows
    (plain path) or over the look-ahead's plain tail of at most 6144 columns plus the ``ns`` rows."""
    budget = 128 * 128 * itemsize // 4
    if n <= 4096:
        m, npb = n + ns, -(-n // 128)
    else:
        m, npb = (min(n, 6144) if lookahead else n) + ns, 8
    return 96 + 2 * (-(-m // 64)) * npb <= budget


class FactoredDense(Dense):
    """A ``Dense`` whose Cholesky factor is known in advance and whose entries are only
    computed (by ``build``) if somebody asks for them."""

    def __init__(self, build, chol, shape, dtype, device):
        super().__init__(None)
        self._build, self._chol = build, chol
        self._shape, self._dtype, self._device = tuple(shape), dtype, device

    @property
    def mat(self):
        if self._mat is None:
            self._mat = self._build()
        return self._mat

    dtype = property(lambda self: self._dtype)
    device = property(lambda self: self._device)
    shape = property(lambda self: self._shape)


def to_matrix(a):
    """``convert(a, AbstractMatrix)`` (random.py:110)."""
    if isinstance(a, AbstractMatrix):
        return a
    if torch.is_tensor(a):
        if a.dim() < 2:
            raise ValueError("a variance must be a matrix")
        return Dense(a)
    raise TypeError(f"cannot interpret {type(a).__name__} as a matrix")


LOG_2_PI = math.log(2 * math.pi)
