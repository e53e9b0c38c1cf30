"""Development aid: forward / backward wall time of the differentiable pseudo-point bound
(cfg5 shape by default: N=200000, M=4096, D=8, fp32, VFE)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stheno_amd as st

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dt = torch.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else torch.float32
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
x = torch.randn(n, 8, generator=g, dtype=dt).to(dev)
y = torch.randn(n, 1, generator=g, dtype=dt).to(dev)
z = torch.randn(m, 8, generator=torch.Generator().manual_seed(2), dtype=dt).to(dev).requires_grad_(True)
v = torch.tensor(1.0, dtype=torch.float64, requires_grad=True)
s = torch.tensor(1.0, dtype=torch.float64, requires_grad=True)
nz = torch.tensor(0.1, dtype=dt, device=dev, requires_grad=True)
st.B.epsilon = 1e-6 if dt == torch.float32 else 1e-10
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f = st.GP(v * st.EQ().stretch(s))
    elbo = st.PseudoObs(f(z), f(x, nz), y).elbo(f.measure)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    elbo.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"n={n} m={m} rep{rep}: elbo {float(elbo):.6e} forward {1e3*(t1-t0):.1f} ms  backward {1e3*(t2-t1):.1f} ms  "
          f"grads v {float(v.grad):.4e} s {float(s.grad):.4e} noise {float(nz.grad):.4e} |dz| {float(z.grad.norm()):.4e} "
          f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
    v.grad = s.grad = nz.grad = z.grad = None
    del elbo, f
