"""RCCL on ONE MI355X: the exchange steps of the sharded paths (``stheno_amd/dist.py``) through a one-rank ``nccl``
process group -- the same ``torch.distributed`` calls the 8-GPU run makes (on ROCm the ``nccl`` backend IS RCCL).

    NCCL_DEBUG=INFO python scripts/rccl_1rank.py            (tests/test_round3_evidence.py runs exactly this)

Prints RCCL's own init banner (through NCCL_DEBUG) and ONE JSON line: the sharded log-densities / their sum / the
observation-sharded VFE bound against the committed golden fixtures, and the latency of the collectives the batched
configuration issues per step (all-gather of B/G log-densities; one-scalar all-reduce) plus the M x (M + 2) all-reduce
of the sharded ELBO at M = 4096.  Reference semantics of the batched computation: tests/model/test_cases.py:134-155."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}
    try:
        import stheno_amd as st
        from stheno_amd import B
        from stheno_amd.dist import sharded_elbo, sharded_logpdf, sharded_logpdf_sum

        gdir = os.path.join(ROOT, "tests", "golden")
        g = np.load(os.path.join(gdir, "batched_eq_b16_n100_d3.npz"))
        x, y = torch.as_tensor(g["x"], device=dev), torch.as_tensor(g["y"], device=dev)
        p = st.GP(2 * st.EQ().stretch(0.5))
        full = sharded_logpdf(p, x, 0.1, y, x.shape[0])
        s = sharded_logpdf_sum(p, x, 0.1, y)
        out["logpdf_max_rel_err"] = float(np.max(np.abs(full.cpu().numpy() - g["logpdf"]) / np.abs(g["logpdf"])))
        out["logpdf_sum_rel_err"] = float(abs(float(s) - g["logpdf"].sum()) / abs(g["logpdf"].sum()))

        gs = np.load(os.path.join(gdir, "sparse_eq_n400_m50_d2.npz"))
        B.epsilon = float(gs["epsilon"])
        xs_, ys_, zs_ = (torch.as_tensor(gs[k], device=dev) for k in ("x", "y", "z"))
        for cls, tag in [(st.PseudoObs, "vfe"), (st.PseudoObsFITC, "fitc"), (st.PseudoObsDTC, "dtc")]:
            prior = st.Measure()
            f = st.GP(st.EQ(), measure=prior)
            obs = cls(f(zs_), f(xs_, float(gs["noise"])), ys_)
            elbo = float(sharded_elbo(obs, prior))
            ref = float(gs["elbo_" + tag][0])
            out["elbo_%s_rel_err" % tag] = abs(elbo - ref) / abs(ref)
        B.epsilon = 1e-12

        def lat(fn, reps=200):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e6

        src = torch.zeros(512, dtype=torch.float32, device=dev)
        dst = torch.empty(512, dtype=torch.float32, device=dev)
        one = torch.zeros(1, dtype=torch.float32, device=dev)
        stats = torch.zeros(4096, 4098, dtype=torch.float32, device=dev)
        out["allgather_512_logpdfs_us"] = lat(lambda: dist.all_gather_into_tensor(dst, src))
        out["allreduce_scalar_us"] = lat(lambda: dist.all_reduce(one))
        out["allreduce_elbo_stats_67MB_us"] = lat(lambda: dist.all_reduce(stats), reps=50)
    finally:
        dist.barrier()
        dist.destroy_process_group()
    print("RCCL1RANK " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
