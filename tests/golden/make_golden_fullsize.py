"""Full-size golden values for BASELINE.json's configurations 2, 3 and 5 (configs[1], [2], [4]):

  cfg2  EQ() kernel, N = 16384, D = 8, fp64, noise 0.1, epsilon 1e-12          -> cfg2_n16384.json
  cfg3  EQ() + Linear(), N = 32768, D = 4, the fp32 inputs of the bench cast to fp64, epsilon 1e-6 (the
        reference's own fp32 setting, README.md:887-888), N* = 16 of the 2048 test points -> cfg3_n32768.json
  cfg5  PseudoObs (VFE), N = 200000, D = 8, M = 4096, same convention          -> cfg5_n200000_m4096.json

cfg3 / cfg5 are the fp64 oracle on the fp32-ROUNDED inputs: what the fp32 HIP path must reproduce to 1e-3 and
the fp64 HIP path (fed the same rounded numbers, same epsilon) to 1e-6.

north_star: "N=16384, D=8 ... with logpdf matching CPU reference to 1e-6 rel".  This script
regenerates EXACTLY the seeded inputs ``bench.make_inputs("dense_f64", ...)`` produces (CPU
``torch.Generator`` streams, so they are the same on the build container and on the GPU box),
runs ``oracle/gp_oracle.py`` -- the NumPy/SciPy restatement of Stheno's NumPy path
(``stheno/random.py:248-280`` for the log-density, ``stheno/model/observations.py:148-168``
+ mlkernels ``PosteriorMean/PosteriorKernel`` for the posterior) -- at the full size on the host,
and writes ``tests/golden/cfg2_n16384.json``: the log-density, posterior mean / marginal variance at
the first 16 test points, sums / extreme values / every 64th value of the posterior at ALL 2048 test points (round 5), and checksums of the inputs (so that the GPU-side test can prove it fed the
same numbers).  Takes a few minutes and ~6 GB on 8 cores.  Re-run:

    python tests/golden/make_golden_fullsize.py [cfg2] [cfg3] [cfg5]

cfg3: ~10 GB (the kernel matrix is built in row blocks by the oracle's own ``kernel_matrix`` so that its N x N
temporaries stay small, then factorised in place block by block, see ``cholesky_blocked``), cfg5: ~35 GB peak
(``oracle.pseudo_obs`` as is).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from bench import NOISE, make_inputs  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402

N_TEST = 16


def checksum(a):
    """sha256 of the little-endian fp64 bytes + plain sums (the latter are what a device-side check can recompute cheaply)."""
    a = np.ascontiguousarray(a, dtype="<f8")
    return {"sha256": hashlib.sha256(a.tobytes()).hexdigest(), "sum": float(a.sum()), "sum_abs": float(np.abs(a).sum()),
            "shape": list(a.shape)}


def all_points(mean, var):
    """Round 5: the posterior at ALL the test points -- plain sums (what a device-side check recomputes cheaply), the extreme values
    and every 64th point -- next to the first N_TEST values the round-3/4 files held."""
    def one(a):
        a = np.asarray(a, dtype=np.float64).reshape(-1)
        return {"n": int(a.size), "sum": float(a.sum()), "sum_abs": float(np.abs(a).sum()), "max_abs": float(np.abs(a).max()),
                "sum_sq": float((a * a).sum()), "every_64th": [float(v) for v in a[::64]]}
    return {"posterior_mean_all": one(mean), "posterior_var_all": one(var)}


def kernel_matrix_blocked(terms, x, rows=2048):
    """``O.kernel_matrix(terms, x)`` evaluated row block by row block (element-wise identical arithmetic: every entry
    comes out of the same ``pw_dists2`` / ``_kappa`` expressions) -- keeps the temporaries at rows x N."""
    n = x.shape[0]
    k = np.empty((n, n), dtype=x.dtype)
    for i in range(0, n, rows):
        k[i:i + rows] = O.kernel_matrix(terms, x[i:i + rows], x)
    return k


def cholesky_blocked(k, nb=4096):
    """Lower Cholesky factor of ``k`` IN PLACE, as LAPACK's own blocked right-looking algorithm spelt out on ``nb``-blocks with the
    oracle's calls: ``np.linalg.cholesky`` (potrf) on the diagonal block, ``scipy.linalg.solve_triangular`` (trsm) for the rows
    below, ``@`` (gemm) for the trailing update.  Why not ``np.linalg.cholesky(k)`` as ``oracle.cholesky`` does: at N = 32768 the
    OpenBLAS bundled with this NumPy segfaults inside its parallel potrf on the 8-core build container (twice, same address --
    ``dmesg``: libscipy_openblas64); at the orders it survives, the two agree to round-off (checked below at N = 6000)."""
    import scipy.linalg as sla

    n = k.shape[0]
    for j in range(0, n, nb):
        je = min(j + nb, n)
        k[j:je, j:je] = np.linalg.cholesky(k[j:je, j:je])
        if je < n:
            k[je:, j:je] = sla.solve_triangular(k[j:je, j:je], k[je:, j:je].T, lower=True, check_finite=False).T
            for i in range(je, n, nb):          # lower block triangle of the trailing matrix, one block row at a time
                ie = min(i + nb, n)
                k[i:ie, je:ie] -= k[i:ie, j:je] @ k[je:ie, j:je].T
    for j in range(0, n, nb):                   # zeros above the diagonal (np.linalg.cholesky's convention)
        je = min(j + nb, n)
        k[j:je, je:] = 0.0
    return k


def _check_blocked_cholesky():
    rng = np.random.default_rng(7)
    x = rng.standard_normal((6000, 4))
    a = O.kernel_matrix([("eq", 1.0, 1.0), ("linear", 1.0, 1.0)], x)
    a[np.diag_indices_from(a)] += NOISE
    ref = np.linalg.cholesky(a)
    got = cholesky_blocked(a.copy(), nb=1024)
    err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
    assert err < 1e-13, err
    return err


def cfg3():
    print("blocked Cholesky vs np.linalg.cholesky at N = 6000: %.2e" % _check_blocked_cholesky(), flush=True)
    w, t = make_inputs("sum_f32", torch.device("cpu"))
    assert t["x"].dtype == torch.float32 and t["x"].shape == (32768, 4)
    x, y, xs = (t[k].double().numpy() for k in ("x", "y", "xs"))      # the fp32-rounded numbers, in fp64
    terms = [("eq", 1.0, 1.0), ("linear", 1.0, 1.0)]
    eps = 1e-6
    t0 = time.perf_counter()
    k = kernel_matrix_blocked(terms, x)
    d = np.diag_indices_from(k)
    k[d] += NOISE                     # fdd.py:79
    k[d] += eps                       # B.reg (same order of additions as oracle.reg)
    chol = cholesky_blocked(k)        # (= O.cholesky: see cholesky_blocked for why not in one LAPACK call)
    del k
    print("factorised after %.0f s" % (time.perf_counter() - t0), flush=True)
    logdet = O.logdet_chol(chol)
    quad = float(O.iqf_diag(chol, y)[0])
    lp = -(logdet + x.shape[0] * O.LOG_2_PI + quad) / 2
    ks = O.kernel_matrix(terms, x, xs)
    v = O.solve_lower(chol, ks)
    mean_all = (v.T @ O.solve_lower(chol, y))[:, 0]
    var_all = O.kernel_diag(terms, xs).reshape(-1) - np.sum(v * v, axis=0)
    mean, var = mean_all[:N_TEST], var_all[:N_TEST]
    dt = time.perf_counter() - t0
    out = {
        "config": "BASELINE.json configs[2]: EQ()+Linear(), N=32768, D=4, fp32 inputs (cast to fp64 for the oracle), noise 0.1, epsilon 1e-6",
        "generator": "tests/golden/make_golden_fullsize.py cfg3 (oracle/gp_oracle.py on bench.make_inputs('sum_f32'))",
        "noise": NOISE, "epsilon": eps, "n_test": N_TEST,
        "inputs": {"x": checksum(x), "y": checksum(y), "xs_first": checksum(xs[:N_TEST])},
        "logpdf": float(lp), "logdet": float(logdet), "quadratic_form": quad,
        "posterior_mean": [float(a) for a in mean], "posterior_var": [float(a) for a in var],
        "oracle_seconds": round(dt, 1),
    }
    out.update(all_points(mean_all, var_all))
    with open(os.path.join(HERE, "cfg3_n32768.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:400], "...", flush=True)


def cfg5():
    w, t = make_inputs("sparse_f32", torch.device("cpu"))
    assert t["x"].dtype == torch.float32 and t["x"].shape == (200000, 8) and t["z"].shape == (4096, 8)
    x, y, z = (t[k].double().numpy() for k in ("x", "y", "z"))
    terms = [("eq", 1.0, 1.0)]
    eps = 1e-6
    t0 = time.perf_counter()
    r = O.pseudo_obs(terms, x, NOISE, y, z, method="vfe", eps=eps)
    dt = time.perf_counter() - t0
    out = {
        "config": "BASELINE.json configs[4]: PseudoObs VFE, EQ(), N=200000, D=8, M=4096, fp32 inputs (cast to fp64 for the oracle), noise 0.1, epsilon 1e-6",
        "generator": "tests/golden/make_golden_fullsize.py cfg5 (oracle/gp_oracle.py pseudo_obs on bench.make_inputs('sparse_f32'))",
        "noise": NOISE, "epsilon": eps, "n_mu": N_TEST,
        "inputs": {"x": checksum(x), "y": checksum(y), "z": checksum(z)},
        "elbo": float(r["elbo"]), "mu_first": [float(a) for a in r["mu"][:N_TEST, 0]],
        "mu_checksum": {"sum": float(r["mu"].sum()), "sum_abs": float(np.abs(r["mu"]).sum()), "max_abs": float(np.abs(r["mu"]).max())},
        "oracle_seconds": round(dt, 1),
    }
    with open(os.path.join(HERE, "cfg5_n200000_m4096.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:400], "...", flush=True)


def cfg2():
    w, t = make_inputs("dense_f64", torch.device("cpu"))
    x, y, xs = (t[k].numpy() for k in ("x", "y", "xs"))
    assert x.shape == (16384, 8) and x.dtype == np.float64
    terms = [("eq", 1.0, 1.0)]
    eps = 1e-12
    t0 = time.perf_counter()
    k = O.kernel_matrix(terms, x)
    k[np.diag_indices_from(k)] += NOISE
    chol = O.cholesky(k, eps)
    del k
    logdet = O.logdet_chol(chol)
    quad = float(O.iqf_diag(chol, y)[0])
    lp = -(logdet + x.shape[0] * O.LOG_2_PI + quad) / 2
    ks = O.kernel_matrix(terms, x, xs)
    v = O.solve_lower(chol, ks)
    mean_all = (v.T @ O.solve_lower(chol, y))[:, 0]
    var_all = O.kernel_diag(terms, xs).reshape(-1) - np.sum(v * v, axis=0)
    mean, var = mean_all[:N_TEST], var_all[:N_TEST]
    dt = time.perf_counter() - t0
    out = {
        "config": "BASELINE.json configs[1]: EQ(), N=16384, D=8, fp64, noise 0.1, epsilon 1e-12",
        "generator": "tests/golden/make_golden_fullsize.py (oracle/gp_oracle.py on bench.make_inputs('dense_f64'))",
        "noise": NOISE, "epsilon": eps, "n_test": N_TEST,
        "inputs": {"x": checksum(x), "y": checksum(y), "xs_first": checksum(xs[:N_TEST])},
        "logpdf": float(lp), "logdet": float(logdet), "quadratic_form": quad,
        "posterior_mean": [float(a) for a in mean], "posterior_var": [float(a) for a in var],
        "oracle_seconds": round(dt, 1),
    }
    out.update(all_points(mean_all, var_all))
    with open(os.path.join(HERE, "cfg2_n16384.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:400], "...")


if __name__ == "__main__":
    for which in (sys.argv[1:] or ["cfg2"]):
        {"cfg2": cfg2, "cfg3": cfg3, "cfg5": cfg5}[which]()
