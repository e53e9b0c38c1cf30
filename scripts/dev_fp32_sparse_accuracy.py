"""fp32 accuracy of the pseudo-point path on the golden sparse case (N = 400, M = 50 clustered inducing points,
kappa(K_z) ~ 1e8) at the reference's fp32 jitter 1e-6 and at larger ones: HIP fp32 vs the fp64 oracle at the
SAME epsilon, next to the oracle itself evaluated in float32 (LAPACK fp32 on the host)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

import stheno_amd as st
from oracle import gp_oracle as O
from stheno_amd import B

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sparse_eq_n400_m50_d2.npz"))
terms = list(zip(g["kinds"], g["variances"], g["scales"]))


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def dev(a, dtype):
    return torch.as_tensor(np.asarray(a), dtype=dtype, device="cuda")


for e in (1e-6, 1e-5, 1e-4):
    for tag, cls in (("vfe", st.PseudoObs), ("fitc", st.PseudoObsFITC), ("dtc", st.PseudoObsDTC)):
        ref = O.pseudo_obs(terms, g["x"], float(g["noise"]), g["y"], g["z"], method=tag, eps=e)["elbo"]
        rm, _, rv = O.pseudo_posterior(terms, g["x"], float(g["noise"]), g["y"], g["z"], g["xs"], method=tag, eps=e, full_cov=False)
        out = {}
        for name, dtype in (("hip fp32", torch.float32), ("hip fp64", torch.float64)):
            B.epsilon = e
            m = st.Measure()
            f = st.GP(st.EQ(), measure=m)
            x, z, xs, y = (dev(g[k], dtype) for k in ("x", "z", "xs", "y"))
            obs = cls(f(z), f(x, float(g["noise"])), y)
            el = float(obs.elbo(m))
            mean, vd = (m | obs)(f)(xs).marginals()
            out[name] = (abs(el - ref) / abs(ref), rel(mean.cpu().numpy(), rm), rel(vd.cpu().numpy(), np.maximum(rv, 0)))
        try:
            f32 = lambda a: np.asarray(a, dtype=np.float32)  # noqa: E731
            el32 = O.pseudo_obs(terms, f32(g["x"]), float(g["noise"]), f32(g["y"]), f32(g["z"]), method=tag, eps=e)["elbo"]
            m32, _, v32 = O.pseudo_posterior(terms, f32(g["x"]), float(g["noise"]), f32(g["y"]), f32(g["z"]), f32(g["xs"]), method=tag, eps=e, full_cov=False)
            out["numpy fp32"] = (abs(float(el32) - ref) / abs(ref), rel(m32, rm), rel(np.maximum(v32, 0), np.maximum(rv, 0)))
        except Exception as ex:  # noqa: BLE001
            out["numpy fp32"] = (float("nan"),) * 3
            print("numpy fp32 failed:", type(ex).__name__, ex)
        for name, (a, b, c) in out.items():
            print(f"eps={e:.0e} {tag:4s} {name:10s}: elbo {a:.2e}  mean {b:.2e}  var {c:.2e}")
B.epsilon = 1e-12
