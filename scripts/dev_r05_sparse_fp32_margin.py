"""Round 5: the margin of the fp32 pseudo-point goldens (tests/test_gpu_parity.py::test_sparse_golden) -- the errors against the fp64
oracle at the same jitter, for whatever libgpk.so is loaded (GPK_DEV=1 -> csrc/dev/libgpk.so, so that two builds can be compared on one box)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stheno_amd as st
from stheno_amd import B
from oracle import gp_oracle as O
from tests.test_gpu_parity import golden, kernel_from, rel, dev, eps

for name in ["sparse_eq_n400_m50_d2", "sparse_matern32_linear_n300_m40_d3", "sparse_matern52_n350_m45_d2"]:
    g = golden(name + ".npz")
    terms = list(zip(g["kinds"], g["variances"], g["scales"]))
    e = 1e-6
    with eps(e):
        m = st.Measure()
        f = st.GP(kernel_from(g), measure=m)
        x, z, xs, y = (dev(g[k], torch.float32) for k in ("x", "z", "xs", "y"))
        for cls, tag in [(st.PseudoObs, "vfe"), (st.PseudoObsFITC, "fitc"), (st.PseudoObsDTC, "dtc")]:
            obs = cls(f(z), f(x, float(g["noise"])), y)
            ref_elbo = np.atleast_1d(O.pseudo_obs(terms, g["x"], float(g["noise"]), g["y"], g["z"], method=tag, eps=e)["elbo"])
            ref_mean, _, ref_vd = O.pseudo_posterior(terms, g["x"], float(g["noise"]), g["y"], g["z"], g["xs"], method=tag, eps=e, full_cov=False)
            mean, vd = (m | obs)(f)(xs).marginals()
            print(f"{name:38s} {tag:5s} elbo {rel(obs.elbo(m).reshape(1), ref_elbo):.3e}  mean {rel(mean, ref_mean):.3e}  var {rel(vd, np.maximum(ref_vd, 0)):.3e}")
