"""Generate the golden fixtures in this directory.

The reference package cannot be imported in the build container (its dependencies
``lab``/``matrix``/``mlkernels``/``plum`` are absent and there is no network), so the
vectors come from (a) the known answers printed in the reference's README, copied here
as data with their ``file:line``, and (b) ``oracle/gp_oracle.py`` -- the NumPy/SciPy
restatement, itself pinned against (a), SciPy's multivariate normal and the exact-vs-sparse
identities in ``tests/test_oracle_pins.py``.  Re-run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import gp_oracle as O  # noqa: E402


def readme_kats():
    """Known answers printed in the reference README (data, not code)."""
    return {
        "source": "wesselb/stheno README.md",
        "eq_matrix_x012": {  # README.md:475-479
            "x": [0.0, 1.0, 2.0],
            "k": [[1.0, 0.607, 0.135], [0.607, 1.0, 0.607], [0.135, 0.607, 1.0]],
            "decimals": 3,
        },
        "logpdf_y1": {  # README.md:481-489
            "x": [0.0, 1.0, 2.0],
            "y": [-0.45172746, 0.46581948, 0.78929767],
            "logpdf": -2.811609567720761,
        },
        "logpdf_y2": {  # README.md:491-497
            "x": [0.0, 1.0, 2.0],
            "y": [[-0.43771276, -2.36741858], [0.86080043, -1.22503079], [2.15779126, -0.75319405]],
            "logpdf": [-4.82949038, -5.40084225],
        },
        "posterior_20s": {  # README.md:43-86: x = linspace(0, 2, 10), y = x**2, GP(EQ()), no noise
            "x_linspace": [0.0, 2.0, 10],
            "x_new": [1.0, 2.0, 3.0],
            "mean": [1.00000068, 3.99999999, 8.4825932],
            "var": [
                [8.03246358e-13, 7.77156117e-16, -4.57690943e-09],
                [7.77156117e-16, 9.99866856e-13, 2.77333267e-10],
                [-4.57690943e-09, 2.77333267e-10, 3.31283378e-03],
            ],
            "epsilon": 1e-12,
        },
        "elbo_gap": {  # README.md:686-720: N = 2000 on [0, 10], M = 100, noise 1
            "value": -3.537934389896691e-10,
            "n": 2000, "m": 100, "noise": 1.0,
        },
    }


def dense_case(name, terms, n, d, ns, noise, seed, c=1):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d))
    xs = rng.standard_normal((ns, d))
    k = O.kernel_matrix(terms, x) + noise * np.eye(n)
    y = np.linalg.cholesky(k) @ rng.standard_normal((n, c))
    logpdf = O.gp_logpdf(terms, x, noise, y)
    mean, var, var_diag = O.gp_posterior(terms, x, noise, y[:, :1], xs)
    np.savez(
        os.path.join(HERE, name + ".npz"),
        kinds=np.array([t[0] for t in terms]), variances=np.array([t[1] for t in terms]),
        scales=np.array([t[2] for t in terms]), x=x, xs=xs, y=y, noise=np.array(noise), epsilon=np.array(1e-12),
        logpdf=np.atleast_1d(logpdf), post_mean=mean, post_var=var, post_var_diag=var_diag,
        kdiag=O.kernel_diag(terms, x), k_corner=O.kernel_matrix(terms, x[:8], xs[:8]),
    )


def batched_case(name, terms, b, n, d, noise, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((b, n, d))
    y = np.stack([np.linalg.cholesky(O.kernel_matrix(terms, x[i]) + noise * np.eye(n)) @ rng.standard_normal((n, 1)) for i in range(b)])
    np.savez(
        os.path.join(HERE, name + ".npz"),
        kinds=np.array([t[0] for t in terms]), variances=np.array([t[1] for t in terms]),
        scales=np.array([t[2] for t in terms]), x=x, y=y, noise=np.array(noise), epsilon=np.array(1e-12),
        logpdf=O.gp_logpdf_batched(terms, x, noise, y),
    )


def sparse_case(name, terms, n, m, d, ns, noise, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 4, (n, d))
    z = rng.uniform(0, 4, (m, d))
    xs = rng.uniform(0, 4, (ns, d))
    y = np.linalg.cholesky(O.kernel_matrix(terms, x) + noise * np.eye(n)) @ rng.standard_normal((n, 1))
    out = dict(
        kinds=np.array([t[0] for t in terms]), variances=np.array([t[1] for t in terms]),
        scales=np.array([t[2] for t in terms]), x=x, z=z, xs=xs, y=y, noise=np.array(noise), epsilon=np.array(1e-10),
        exact_logpdf=np.atleast_1d(O.gp_logpdf(terms, x, noise, y, eps=1e-10)),
    )
    for method in ("vfe", "fitc", "dtc"):
        r = O.pseudo_obs(terms, x, noise, y, z, method=method, eps=1e-10)
        mean, _, vd = O.pseudo_posterior(terms, x, noise, y, z, xs, method=method, eps=1e-10, full_cov=False)
        out[f"elbo_{method}"] = np.atleast_1d(r["elbo"])
        out[f"mu_{method}"] = r["mu"]
        out[f"A_{method}"] = r["A"]
        out[f"post_mean_{method}"] = mean
        out[f"post_var_diag_{method}"] = vd
    np.savez(os.path.join(HERE, name + ".npz"), **out)


def main():
    with open(os.path.join(HERE, "readme_kats.json"), "w") as f:
        json.dump(readme_kats(), f, indent=1)
    dense_case("dense_eq_n256_d8", [("eq", 1.0, 1.0)], 256, 8, 64, 0.1, 0)
    dense_case("dense_eq_n300_d1_c3", [("eq", 1.3, 0.7)], 300, 1, 40, 0.05, 1, c=3)
    dense_case("dense_matern12_n200_d3", [("matern12", 0.8, 1.5)], 200, 3, 50, 0.1, 2)
    dense_case("dense_matern32_n200_d3", [("matern32", 1.0, 1.2)], 200, 3, 50, 0.1, 3)
    dense_case("dense_matern52_n333_d5", [("matern52", 2.0, 2.0)], 333, 5, 33, 0.2, 4)
    dense_case("dense_eq_linear_n512_d4", [("eq", 1.0, 1.0), ("linear", 1.0, 1.0)], 512, 4, 96, 0.1, 5)
    batched_case("batched_eq_b16_n100_d3", [("eq", 2.0, 0.5)], 16, 100, 3, 0.1, 6)
    sparse_case("sparse_eq_n400_m50_d2", [("eq", 1.0, 1.0)], 400, 50, 2, 30, 0.1, 7)
    sparse_case("sparse_matern32_linear_n300_m40_d3", [("matern32", 1.2, 0.9), ("linear", 0.4, 2.0)], 300, 40, 3, 25, 0.2, 8)
    sparse_case("sparse_matern52_n350_m45_d2", [("matern52", 0.8, 1.1)], 350, 45, 2, 25, 0.15, 9)


if __name__ == "__main__":
    main()
