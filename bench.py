"""Benchmark of the dense GP inference hot path on MI355X.

Metric (BASELINE.json): GP logpdf+posterior evals/sec at N=16384 D=8 fp64, with the
Cholesky trailing-update GEMM priced against the fp64 MFMA peak.

One STEP = one eval of ``configs[1]``:  from ``x`` (N x D, resident in HBM) build
``K + sigma^2 I`` (lower triangle, jitter fused), Cholesky-factorise it in place,
``f(x, noise).logpdf(y)``, condition ``f | (f(x, noise), y)`` (re-using the factor), and
the posterior mean + marginal variance at N* = 2048 test points -- through the public
``stheno_amd`` API, i.e. through libgpk.so.  Nothing is cached across steps.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload dense_f64|sum_f32|batched_f32|sparse_f32]

For N > 1 it is launched by ``python -m torch.distributed.run --nproc-per-node N ...``:
the dense workload does not shard (one coupled factorisation), so every rank runs an
independent replica ("replicas only"); ``batched_f32`` shards its 512 GPs over the ranks
and all-gathers the log-densities (stheno_amd/dist.py).

Rank 0 prints ONE JSON line.  ``roofline`` is the dominant kernel (the MFMA GEMM that
performs the Cholesky trailing update, ``gemm_kernel<double, 128, true, true, false>``):
algorithmic flops of its launches in one step / their summed duration, measured with HIP
events on the launch stream (gpk_prof_* hooks) in extra, untimed steps right after the
timed region.  ``cpu_baseline`` is the NumPy/SciPy oracle (a restatement of Stheno's
NumPy path) timed on the host cores on a bounded sample -- a reported baseline, not
the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import stheno_amd as st  # noqa: E402
from stheno_amd import _native  # noqa: E402

PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}     # dense MFMA peaks, MI355X_MICROARCH.md / SURVEY 8(d)
NOISE = 0.1

WORKLOADS = {
    # name: (description, dtype, N, D, N*, extra)
    "dense_f64": dict(desc="EQ() kernel, N=16384 D=8 fp64, logpdf + condition, posterior mean/var at N*=2048",
                      dtype="f64", n=16384, d=8, ns=2048),
    "sum_f32": dict(desc="EQ()+Linear(), N=32768 D=4 fp32, logpdf + posterior at 2048 test points",
                    dtype="f32", n=32768, d=4, ns=2048),
    "batched_f32": dict(desc="512 independent GPs x N=2048 D=3 EQ fp32, logpdf, sharded over ranks",
                        dtype="f32", n=2048, d=3, b=512),
    "sparse_f32": dict(desc="PseudoObs VFE ELBO, N=200000 D=8, M=4096 inducing, EQ fp32",
                       dtype="f32", n=200000, d=8, m=4096),
}
TORCH_DTYPE = {"f64": torch.float64, "f32": torch.float32}


def make_inputs(name, device, rank=0, world=1, n_override=None):
    """Synthetic inputs of the named config (seeded; ``x ~ N(0,1)``, ``y ~ N(0,1)``)."""
    w = dict(WORKLOADS[name])
    if n_override:
        w["n"] = n_override
    dt = TORCH_DTYPE[w["dtype"]]
    g = torch.Generator(device="cpu").manual_seed(0)
    if name in ("dense_f64", "sum_f32"):
        x = torch.randn(w["n"], w["d"], generator=g, dtype=torch.float64).to(dt).to(device)
        y = torch.randn(w["n"], 1, generator=g, dtype=torch.float64).to(dt).to(device)
        xs = torch.randn(w["ns"], w["d"], generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dt).to(device)
        return w, dict(x=x, y=y, xs=xs)
    if name == "batched_f32":
        from stheno_amd.dist import shard_bounds

        lo, hi = shard_bounds(w["b"], world, rank)
        x = torch.randn(w["b"], w["n"], w["d"], generator=g, dtype=torch.float32)[lo:hi].to(device)
        y = torch.randn(w["b"], w["n"], 1, generator=g, dtype=torch.float32)[lo:hi].to(device)
        return w, dict(x=x, y=y)
    if name == "sparse_f32":
        x = torch.randn(w["n"], w["d"], generator=g, dtype=torch.float32).to(device)
        y = torch.randn(w["n"], 1, generator=g, dtype=torch.float32).to(device)
        z = torch.randn(w["m"], w["d"], generator=torch.Generator().manual_seed(2), dtype=torch.float32).to(device)
        return w, dict(x=x, y=y, z=z)
    raise ValueError(name)


def make_step(name, w, t):
    """Returns a zero-argument callable running one eval; its return value is kept alive
    until the next call (so allocations are part of the step, like for a user)."""
    if name == "dense_f64":
        kernel = st.EQ()
    elif name == "sum_f32":
        kernel = st.EQ() + st.Linear()
    else:
        kernel = st.EQ()

    if name in ("dense_f64", "sum_f32"):
        def step():
            f = st.GP(kernel)
            fdd = f(t["x"], NOISE)
            lp = fdd.logpdf(t["y"])
            post = f | (fdd, t["y"])
            mean, var = post(t["xs"]).marginals()
            return lp, mean, var
    elif name == "batched_f32":
        from stheno_amd.dist import sharded_logpdf

        def step():
            f = st.GP(kernel)
            return sharded_logpdf(f, t["x"], NOISE, t["y"], w["b"])
    else:
        def step():
            prior = st.Measure()
            f = st.GP(kernel, measure=prior)
            return st.PseudoObs(f(t["z"]), f(t["x"], NOISE), t["y"]).elbo(prior)
    return step


def pmc_traffic(name, w):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (``rocprofv3 --pmc FETCH_SIZE`` and ``--pmc WRITE_SIZE`` run separately on this same
    command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  PMC
    collection cannot run inside the timed process, so the figure is read from
    ``profiles/``; ``None`` if no pass exists for this workload."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_%s.json" % name)
    if not os.path.exists(path) or w["n"] != WORKLOADS[name]["n"]:
        return None
    with open(path) as f:
        d = json.load(f)
    k = d["kernels"].get("gemm_kernel<%s, 128, true, true, false, 1>" % ("double" if w["dtype"] == "f64" else "float"))
    return None if k is None else k["hbm_bytes_per_launch"]


def cpu_baseline(name):
    """The oracle (NumPy/SciPy restatement of Stheno's NumPy path) on the host cores, on a
    bounded sample of the same workload."""
    from oracle import gp_oracle as O

    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    if name != "dense_f64":
        return None
    w = WORKLOADS[name]
    n_s = 8192
    rng = np.random.default_rng(0)
    x, y = rng.standard_normal((n_s, w["d"])), rng.standard_normal((n_s, 1))
    xs = rng.standard_normal((w["ns"], w["d"]))
    terms = [("eq", 1.0, 1.0)]
    t0 = time.perf_counter()
    # one eval with ONE shared factorisation (the same op sequence as the GPU step)
    k = O.kernel_matrix(terms, x) + NOISE * np.eye(n_s)
    chol = O.cholesky(k, 1e-12)
    lp = -(O.logdet_chol(chol) + n_s * O.LOG_2_PI + O.iqf_diag(chol, y)) / 2
    v = O.solve_lower(chol, O.kernel_matrix(terms, x, xs))
    mean = v.T @ O.solve_lower(chol, y)
    var = O.kernel_diag(terms, xs) - np.sum(v * v, axis=0)
    dt = time.perf_counter() - t0
    assert np.isfinite(lp).all() and np.isfinite(mean).all() and np.isfinite(var).all()
    scale = (w["n"] / n_s) ** 3
    return {
        "value": 1.0 / (dt * scale),
        "unit": "evals/s",
        "cores": int(threads),
        "kind": "port",
        "sample": f"oracle/gp_oracle.py (NumPy/SciPy, fp64) at N={n_s}, D={w['d']}, N*={w['ns']}: "
                  f"{dt:.2f} s per eval measured; value extrapolated to N={w['n']} by (N/{n_s})^3 = x{scale:.0f}",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="dense_f64", choices=sorted(WORKLOADS))
    ap.add_argument("--n", type=int, default=0, help="override N (development only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path in stheno_amd)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("GPK_BENCH_FORCE_DIST") == "1"   # (the env switch exercises the RCCL path on one GPU)
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    name = args.workload
    w, tensors = make_inputs(name, device, rank, world, args.n or None)
    if w["dtype"] == "f32":
        st.B.epsilon = 1e-6          # the reference's own fp32 setting (README.md:887-888)
    step = make_step(name, w, tensors)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    keep = None
    for _ in range(args.warmup):
        keep = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        keep = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])

    # -- live roofline of the dominant kernel: extra untimed steps under HIP-event hooks --
    roofline = None
    lib = _native.load()
    import ctypes

    variant = (8 if w["dtype"] == "f64" else 0) + 4 + 2     # NT, non-edge: the potrf trailing/panel GEMM
    prof_steps = 2
    lib.gpk_prof_start()
    for _ in range(prof_steps):
        keep = step()
    torch.cuda.synchronize()
    ms, nl, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
    lib.gpk_prof_stop(variant, ctypes.byref(ms), ctypes.byref(nl), ctypes.byref(fl))
    if nl.value > 0 and ms.value > 0:
        achieved = fl.value / (ms.value * 1e-3) / 1e12
        peak = PEAK_TFLOPS[w["dtype"]]
        roofline = {
            "kernel": f"gemm_kernel<{'double' if w['dtype'] == 'f64' else 'float'}, 128, true, true, false, 1>",
            "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": pmc_traffic(name, w),
            "launches_per_step": nl.value // prof_steps,
            "avg_launch_us": ms.value * 1e3 / nl.value,
            "algorithmic_flops_per_step": fl.value / prof_steps,
            "kernel_ms_per_step": ms.value / prof_steps,
        }

    if rank == 0:
        units = args.steps * (world if name != "batched_f32" else 1)
        if name == "batched_f32":
            metric, unit, value = "batched GP logpdfs/sec (512 x N=2048 D=3 fp32)", "GPs/s", w["b"] * args.steps / elapsed
            scaling = "strong"
        elif name == "sparse_f32":
            metric, unit, value, scaling = "VFE ELBO evals/sec at N=200000 M=4096 fp32", "evals/s", units / elapsed, "weak"
        else:
            metric = "GP logpdf+posterior evals/sec at N=%d D=%d %s" % (w["n"], w["d"], "fp64" if w["dtype"] == "f64" else "fp32")
            unit, value, scaling = "evals/s", units / elapsed, "weak"
        out = {
            "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": w["dtype"], "data": "synthetic",
            "config": {"workload": w["desc"], "noise_variance": NOISE, "epsilon": st.B.epsilon,
                       "parallelism": ("replicas only (%d independent evals in flight)" % world) if name != "batched_f32"
                       else "GPs sharded over %d ranks, all-gather of log-densities" % world},
            "roofline": roofline,
            "cpu_baseline": None if (args.no_cpu_baseline or world > 1 or args.n) else cpu_baseline(name),
        }
        line = json.dumps(out)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(line, flush=True)          # the ONE JSON line, last thing on stdout (after RCCL's own init / teardown chatter)


if __name__ == "__main__":
    main()
