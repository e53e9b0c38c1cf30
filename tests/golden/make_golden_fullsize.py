"""Full-size golden values for BASELINE.json's headline configuration (configs[1]):
EQ() kernel, N = 16384, D = 8, fp64, noise 0.1, epsilon 1e-12.

north_star: "N=16384, D=8 ... with logpdf matching CPU reference to 1e-6 rel".  This script
regenerates EXACTLY the seeded inputs ``bench.make_inputs("dense_f64", ...)`` produces (CPU
``torch.Generator`` streams, so they are the same on the build container and on the GPU box),
runs ``oracle/gp_oracle.py`` -- the NumPy/SciPy restatement of Stheno's NumPy path
(``stheno/random.py:248-280`` for the log-density, ``stheno/model/observations.py:148-168``
+ mlkernels ``PosteriorMean/PosteriorKernel`` for the posterior) -- at the full size on the host,
and writes ``tests/golden/cfg2_n16384.json``: the log-density, posterior mean / marginal variance at
the first 16 test points, and checksums of the inputs (so that the GPU-side test can prove it fed the
same numbers).  Takes a few minutes and ~6 GB on 8 cores.  Re-run:

    python tests/golden/make_golden_fullsize.py
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


def main():
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
    ks = O.kernel_matrix(terms, x, xs[:N_TEST])
    v = O.solve_lower(chol, ks)
    mean = (v.T @ O.solve_lower(chol, y))[:, 0]
    var = O.kernel_diag(terms, xs[:N_TEST]).reshape(-1) - np.sum(v * v, axis=0)
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
    with open(os.path.join(HERE, "cfg2_n16384.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:400], "...")


if __name__ == "__main__":
    main()
