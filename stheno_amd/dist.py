"""Multi-GPU for the one part of the path that shards: batched computation.

Independent GPs (stheno's batched computation, ``README.md:744-766``,
``tests/model/test_cases.py:134-155``) are partitioned in contiguous blocks over the
ranks of a ``torch.distributed`` process group (one process per GPU; backend ``nccl`` is
RCCL over xGMI on ROCm).  Every rank builds, factorises and solves only its own GPs --
there is no data-path collective.  The only exchange is the result: an all-gather of
``B / G`` log-densities per rank (256 bytes per rank at B = 512, G = 8: latency-bound),
or an all-reduce of one scalar when only the sum is needed.

A single large dense GP does not shard (one coupled N x N factorisation): replicas only.
"""
import torch
import torch.distributed as dist

__all__ = ["shard_bounds", "sharded_logpdf", "sharded_logpdf_sum", "sharded_elbo"]


def shard_bounds(total, world_size, rank):
    """Contiguous block ``[lo, hi)`` of ``total`` items owned by ``rank`` (sizes differ by
    at most one)."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _world(group):
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _collective(group):
    """The exchange step runs whenever a process group exists -- also a one-rank group (RCCL then copies in place): the
    single-GPU run of a sharded job exercises the same calls as the 8-GPU one.  Without a process group there is nothing to call."""
    return dist.is_available() and dist.is_initialized()


def sharded_logpdf(process, x_local, noise, y_local, total, group=None):
    """Log-densities of ``total`` independent GPs, of which this rank holds the shard
    ``x_local`` (b_local, N, D) / ``y_local`` (b_local, N, 1) given by :func:`shard_bounds`.
    Returns the full ``(total,)`` vector on every rank."""
    world, rank = _world(group)
    lo, hi = shard_bounds(total, world, rank)
    if x_local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} should hold {hi - lo} GPs, got {x_local.shape[0]}")
    local = process(x_local, noise).logpdf(y_local).reshape(-1)
    if not _collective(group):
        return local
    sizes = [h - l for l, h in (shard_bounds(total, world, r) for r in range(world))]
    width = max(sizes)
    if all(s == width for s in sizes):
        out = torch.empty((total,), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged split (sizes differ by one): pad every shard to the widest, gather once, trim
    padded = torch.zeros((width,), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * width,), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * width : r * width + s] for r, s in enumerate(sizes)])


def sharded_logpdf_sum(process, x_local, noise, y_local, group=None):
    """Sum of the log-densities over all ranks' GPs (one-scalar all-reduce)."""
    s = process(x_local, noise).logpdf(y_local).sum().reshape(1)
    if _collective(group):
        dist.all_reduce(s, op=dist.ReduceOp.SUM, group=group)
    return s[0]


def sharded_elbo(obs, measure, group=None):
    """Pseudo-point ELBO with the N observations sharded over the ranks.

    ``obs`` is a ``PseudoObs`` / ``PseudoObsFITC`` / ``PseudoObsDTC`` built on every rank from
    the SAME inducing points ``u`` and that rank's shard of the observations
    ``(f(x_local, noise_local), y_local)``.  Each rank builds ``K_z`` and its Cholesky factor
    redundantly (M x M, cheap), computes ``V_g = L_z^{-1} K_{z, x_g}`` for its own columns and
    the sums over its observations; ONE all-reduce (sum) of an ``M x (M + 2)`` buffer -- the
    only exchange step of the path: 67 MB in fp32 at M = 4096, ring time about
    2 * (7/8) * 67 MB / 153 GB/s = 0.8 ms, per-link bound on xGMI -- then every rank finishes the
    M x M solve and holds the same ELBO (``stheno/model/observations.py:279-336`` with the
    sums over observations distributed)."""
    def reduce(stats):
        if _collective(group):
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)

    if torch.is_grad_enabled() and obs._differentiable_requested(measure):
        raise NotImplementedError(
            "gradients of the observation-sharded bound are not implemented (every rank would differentiate its own "
            "shard's bound): take them of `obs.elbo(measure)` on one rank, or wrap this call in torch.no_grad()"
        )
    obs._compute(measure, reduce=reduce)
    return obs._elbo[measure]        # the all-reduced bound (NOT obs.elbo(): that may re-enter the local autograd path)
