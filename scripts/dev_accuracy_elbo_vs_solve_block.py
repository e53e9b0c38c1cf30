"""Development aid: fp32 ELBO / pseudo-point posterior accuracy (vs fp64) as a function of the merged solve-block size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stheno_amd as st
from stheno_amd import B, matrix

dev = torch.device("cuda")
n, m, d = 50000, 2048, 8
g = torch.Generator().manual_seed(0)
x = torch.randn(n, d, generator=g).to(dev); y = torch.randn(n, 1, generator=g).to(dev)
z = torch.randn(m, d, generator=torch.Generator().manual_seed(2)).to(dev)
xs = torch.randn(512, d, generator=torch.Generator().manual_seed(1)).to(dev)
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
orig = matrix._solve_block

def run(dt, eps):
    B.epsilon = eps
    f = st.GP(st.EQ())
    obs = st.PseudoObs(f(z.to(dt)), f(x.to(dt), 0.1), y.to(dt))
    e = obs.elbo(f.measure)
    mean, var = (f | obs)(xs.to(dt)).marginals()
    return float(e), mean, var

e64, m64, v64 = run(torch.float64, 1e-6)      # same jitter as the fp32 run: isolates the arithmetic
print("fp64 elbo", e64)
for sb in (128, 512, 2048):
    matrix._solve_block = lambda n_, r_, f_=True, sb=sb: sb if r_ > 8 else orig(n_, r_, f_)
    e32, m32, v32 = run(torch.float32, 1e-6)
    print(f"sb={sb}: elbo rel {abs(e32 - e64) / abs(e64):.2e} mean rel {rel(m32, m64):.2e} var rel {rel(v32, v64):.2e}")
