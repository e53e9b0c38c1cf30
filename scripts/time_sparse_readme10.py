"""Harness counterpart of the reference's sparse example at ITS size (``readme_example10_sparse.py:8-29``): 50 000 observations
on [0, 7], 20 inducing points on [0, 10], noise variance 0.5, prediction at 100 points -- pseudo-point ELBO, approximate posterior,
``marginal_credible_bounds`` -- timed on the MI355X and compared with ``oracle/gp_oracle.py`` at the SAME size (N x M = 1e6 entries:
the CPU oracle runs it in well under a second).  The example's ``EQ().periodic(2 pi)`` kernel is outside the accelerated path
(out of scope): plain ``EQ()`` here, everything else as in the example.

    python scripts/time_sparse_readme10.py          (on the GPU box; tests/test_round3_evidence.py checks the same numbers)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stheno_amd as st  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def inputs(device):
    rng = np.random.default_rng(10)
    x = np.linspace(0, 10, 100)
    x_obs = np.linspace(0, 7, 50_000)
    x_ind = np.linspace(0, 10, 20)
    y_obs = np.sin(x_obs) + np.sqrt(0.5) * rng.standard_normal(x_obs.shape)
    return {k: torch.as_tensor(v, dtype=torch.float64, device=device) for k, v in dict(x=x, x_obs=x_obs, x_ind=x_ind, y_obs=y_obs).items()}


def run(t):
    prior = st.Measure()
    f = st.GP(st.EQ(), measure=prior)
    obs = st.PseudoObs(f(t["x_ind"]), (f(t["x_obs"], 0.5), t["y_obs"]))
    elbo = obs.elbo(prior)
    f_post = f | obs
    mean, lower, upper = f_post(t["x"]).marginal_credible_bounds()
    return elbo, mean, lower, upper


def main():
    dev = torch.device("cuda")
    t = inputs(dev)
    for _ in range(3):
        out = run(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        out = run(t)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    elbo, mean, lower, upper = out
    h = {k: v.cpu().numpy() for k, v in t.items()}
    terms = [("eq", 1.0, 1.0)]
    t1 = time.perf_counter()
    ref_elbo = float(O.pseudo_obs(terms, h["x_obs"], 0.5, h["y_obs"][:, None], h["x_ind"])["elbo"])
    ref_m, _, ref_v = O.pseudo_posterior(terms, h["x_obs"], 0.5, h["y_obs"][:, None], h["x_ind"], h["x"], full_cov=False)
    cpu_ms = (time.perf_counter() - t1) * 1e3
    ref_lo, ref_hi = ref_m - 1.96 * np.sqrt(np.maximum(ref_v, 0)), ref_m + 1.96 * np.sqrt(np.maximum(ref_v, 0))
    rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
    print(f"SPARSE10 N=50000 M=20 fp64: ELBO + approximate posterior + credible bounds at 100 points  {ms:.3f} ms per run on the GPU "
          f"(oracle on the host: {cpu_ms:.0f} ms);  ELBO {float(elbo):.6f} vs oracle {ref_elbo:.6f} (rel {abs(float(elbo) - ref_elbo) / abs(ref_elbo):.1e}), "
          f"mean rel {rel(mean.cpu().numpy(), ref_m):.1e}, lower rel {rel(lower.cpu().numpy(), ref_lo):.1e}, upper rel {rel(upper.cpu().numpy(), ref_hi):.1e}")


if __name__ == "__main__":
    main()
