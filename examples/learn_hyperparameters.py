"""Hyper-parameter learning with the MI355X path -- the counterpart of the reference's
``readme_example13_optimisation_torch.py`` (exact GP, ``logpdf`` in an Adam loop) and of
``readme_example10_sparse.py`` (pseudo-point bound), written against ``stheno_amd.torch``.

    python examples/learn_hyperparameters.py [exact|sparse] [N]

Every objective evaluation builds the kernel matrix, factorises it in place and back-propagates
through the custom autograd functions of ``stheno_amd/autograd.py`` -- all on the GPU; the only
host work is Adam on a handful of scalars.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a source checkout

from stheno_amd.torch import EQ, GP, B, PseudoObs

mode = sys.argv[1] if len(sys.argv) > 1 else "exact"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (4096 if mode == "exact" else 50000)
dev = torch.device("cuda")
# exact: fp32 with the jitter the reference advises for it (README.md:887-888).  sparse: fp64 -- on 1-D inputs
# the inducing-point Gram matrix is ill-conditioned (cond ~ 1/epsilon) and A = I + V K_n^-1 V^T inherits it;
# the reference's own sparse example (readme_example10_sparse.py) runs in float64 as well.
dt = torch.float32 if mode == "exact" else torch.float64
B.epsilon = 1e-6 if mode == "exact" else 1e-8

g = torch.Generator().manual_seed(0)
x = (torch.rand(n, 1, generator=g, dtype=dt) * 10).to(dev)
y = (torch.sin(x) + 0.3 * torch.cos(3 * x) + 0.2 * torch.randn(n, 1, generator=g, dtype=dt).to(dev))

log_var = torch.zeros((), requires_grad=True)
log_scale = torch.zeros((), requires_grad=True)
log_noise = torch.tensor(-1.0, requires_grad=True)
params = [log_var, log_scale, log_noise]
if mode == "sparse":
    z = torch.linspace(0, 10, 20, device=dev, dtype=dt)[:, None].clone().requires_grad_(True)   # inducing inputs are learned too
    params.append(z)
opt = torch.optim.Adam(params, lr=5e-2)


def objective():
    f = GP(log_var.exp() * EQ().stretch(log_scale.exp()))
    noise = log_noise.exp().to(device=dev, dtype=dt)
    if mode == "exact":
        return -f(x, noise).logpdf(y) / n
    return -PseudoObs(f(z), f(x, noise), y).elbo(f.measure) / n


t0 = time.perf_counter()
for it in range(101):
    opt.zero_grad()
    loss = objective()
    loss.backward()
    opt.step()
    if it % 20 == 0:
        print(f"iter {it:3d}  objective/N {float(loss.detach()):+.5f}  variance {float(log_var.detach().exp()):.3f}  "
              f"scale {float(log_scale.detach().exp()):.3f}  noise {float(log_noise.detach().exp()):.4f}", flush=True)
torch.cuda.synchronize()
print(f"{mode}: N={n}, 101 iterations in {time.perf_counter() - t0:.2f} s")
