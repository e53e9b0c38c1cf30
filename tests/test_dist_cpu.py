"""The N > 1 path on the CPU box: two ``gloo`` ranks shard a batch of independent GPs
(contiguous blocks, no data-path collective) and all-gather the log-densities."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from .conftest import ROOT, golden


def _worker(rank, world, port, total, result_file):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import stheno_amd as st
        from stheno_amd import ops
        from stheno_amd.dist import shard_bounds, sharded_logpdf, sharded_logpdf_sum
        from tests.conftest import OracleBackend

        ops.set_backend(OracleBackend())       # test-only CPU backend (no GPU here)
        g = golden("batched_eq_b16_n100_d3.npz")
        x, y = torch.as_tensor(g["x"][:total]), torch.as_tensor(g["y"][:total])
        lo, hi = shard_bounds(total, world, rank)
        p = st.GP(2 * st.EQ().stretch(0.5))
        full = sharded_logpdf(p, x[lo:hi], 0.1, y[lo:hi], total)
        s = sharded_logpdf_sum(p, x[lo:hi], 0.1, y[lo:hi])
        if rank == 0:
            np.savez(result_file, full=full.numpy(), s=float(s))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [16, 15])   # even split -> all_gather_into_tensor; ragged -> all_gather
def test_two_rank_sharded_logpdf(tmp_path, total):
    port = 29500 + (os.getpid() % 2000) + total
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(2, port, total, out), nprocs=2, join=True)
    res = np.load(out)
    g = golden("batched_eq_b16_n100_d3.npz")
    np.testing.assert_allclose(res["full"], g["logpdf"][:total], rtol=1e-10)
    np.testing.assert_allclose(res["s"], g["logpdf"][:total].sum(), rtol=1e-10)


def _elbo_worker(rank, world, port, result_file):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import stheno_amd as st
        from stheno_amd import B, ops
        from stheno_amd.dist import shard_bounds, sharded_elbo
        from tests.conftest import OracleBackend

        ops.set_backend(OracleBackend())
        g = golden("sparse_eq_n400_m50_d2.npz")
        B.epsilon = float(g["epsilon"])
        x, y, z = (torch.as_tensor(g[k]) for k in ("x", "y", "z"))
        lo, hi = shard_bounds(x.shape[0], world, rank)
        out = {}
        for cls, tag in [(st.PseudoObs, "vfe"), (st.PseudoObsFITC, "fitc"), (st.PseudoObsDTC, "dtc")]:
            m = st.Measure()
            f = st.GP(st.EQ(), measure=m)
            obs = cls(f(z), f(x[lo:hi], float(g["noise"])), y[lo:hi])
            out[tag] = float(sharded_elbo(obs, m))
        if rank == 0:
            np.savez(result_file, **out)
    finally:
        dist.destroy_process_group()


def test_two_rank_observation_sharded_elbo(tmp_path):
    """The one real exchange step of the path: all-reduce of the M x (M + 2) statistics."""
    port = 31500 + (os.getpid() % 2000)
    out = str(tmp_path / "elbo.npz")
    mp.spawn(_elbo_worker, args=(2, port, out), nprocs=2, join=True)
    res = np.load(out)
    g = golden("sparse_eq_n400_m50_d2.npz")
    for tag in ("vfe", "fitc", "dtc"):
        np.testing.assert_allclose(res[tag], g[f"elbo_{tag}"][0], rtol=1e-9)


def test_shard_bounds_partition():
    from stheno_amd.dist import shard_bounds

    for total in (0, 1, 7, 512):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
