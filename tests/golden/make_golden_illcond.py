"""An 80-bit reference for the nearly noise-free regime (round 6; VERDICT r5 "what's weak" 2).

Two fp64 paths through a kernel matrix of condition number ~1e9 ... 1e10 differ from each other by more than either differs from
the truth, so the conditioning sweep of ``tests/test_round5_evidence.py`` (HIP path against the fp64 oracle) cannot say which of the
two is off.  This script evaluates the posterior of

    f ~ GP(EQ()),  y = f(x) + noise,  x sorted uniform on [0, n / 200] (D = 1, ~200 points per length scale),  noise 1e-6

in ``numpy.longdouble`` (x87 extended precision, 64-bit mantissa: ~2000 times finer than fp64) by the textbook route -- unblocked
Cholesky, forward substitution (``stheno/random.py:272-279``, ``stheno/model/observations.py:148-168`` with exact arithmetic in
mind) -- on exactly the fp64 inputs the tests regenerate from the seed, and writes ``tests/golden/illcond_n<N>.json``: the posterior
mean and marginal variance at 64 points, the errors of the fp64 oracle (LAPACK) against them, and checksums of the inputs.

    python tests/golden/make_golden_illcond.py 1536 8192        # n = 8192 takes ~10 minutes on one core

The GPU-side test holds the HIP path to a small multiple of the fp64 oracle's own error against this reference.
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import gp_oracle as O  # noqa: E402

LD = np.longdouble
NOISE = 1e-6


def inputs(n):
    """The sweep's inputs (``tests/test_round5_evidence.py``: same generator calls, same seed rule)."""
    rng = np.random.default_rng(n + int(-np.log10(NOISE)))
    x = np.sort(rng.uniform(0.0, n / 204.8, size=(n, 1)), axis=0)
    y = np.sin(x) + 0.1 * rng.standard_normal((n, 1))
    xs = rng.uniform(0.0, n / 204.8, size=(64, 1))
    return x, y, xs


def eq(a, b):
    d = a.astype(LD) - b.astype(LD).T
    return np.exp(-0.5 * d * d)


def cholesky_ld(a):
    """Unblocked left-looking Cholesky in extended precision, in place (lower triangle)."""
    n = a.shape[0]
    for j in range(n):
        v = a[j:, j] - a[j:, :j] @ a[j, :j]
        a[j:, j] = v / np.sqrt(v[0])
    return a


def forward_ld(l, b):
    b = b.copy()
    for j in range(l.shape[0]):
        b[j] = (b[j] - l[j, :j] @ b[:j]) / l[j, j]
    return b


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def main(n):
    assert np.finfo(LD).nmant >= 63, "numpy.longdouble is not extended precision on this machine"
    x, y, xs = inputs(n)
    t0 = time.time()
    k = eq(x, x)
    k[np.diag_indices(n)] += LD(NOISE) + LD(1e-12)          # (B.epsilon = 1e-12, as the library adds it)
    l = cholesky_ld(k)
    w = forward_ld(l, y.astype(LD))
    v = forward_ld(l, eq(x, xs))
    mean = (v.T @ w).reshape(-1)
    var = (LD(1.0) - np.sum(v * v, axis=0)).reshape(-1)
    logdet = 2 * np.sum(np.log(np.diag(l)))
    logpdf = -(logdet + np.sum(w * w) + n * np.log(2 * LD(np.pi))) / 2
    t_ld = time.time() - t0
    # the fp64 oracle on the same inputs: how far LAPACK's path is from the truth
    terms = [("eq", 1.0, 1.0)]
    o_mean, _, o_var = O.gp_posterior(terms, x, NOISE, y, xs, full_cov=False)
    o_lp = O.gp_logpdf(terms, x, NOISE, y)

    def rel(a, b):
        return float(np.max(np.abs(np.asarray(a, dtype=LD).reshape(-1) - b)) / np.max(np.abs(b)))

    out = {
        "n": n, "noise": NOISE, "epsilon": 1e-12, "precision": "numpy.longdouble (64-bit mantissa)",
        "x_sha256": digest(x), "y_sha256": digest(y), "xs_sha256": digest(xs),
        "mean": [float(m) for m in mean], "var": [float(s) for s in var], "logpdf": float(logpdf),
        "oracle_fp64_error": {"mean": rel(o_mean, mean), "var": rel(o_var, var), "logpdf": float(abs(LD(o_lp) - logpdf) / abs(logpdf))},
        "seconds": round(t_ld, 1),
    }
    path = os.path.join(HERE, f"illcond_n{n}.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(path, out["oracle_fp64_error"], f"{t_ld:.0f} s", flush=True)


if __name__ == "__main__":
    for arg in sys.argv[1:] or ["1536"]:
        main(int(arg))
