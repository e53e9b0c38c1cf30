"""Development aid: forward / backward wall time of the differentiable logpdf."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stheno_amd as st

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
x = torch.randn(n, 8, generator=g, dtype=torch.float64).to(dev)
y = torch.randn(n, 1, generator=g, dtype=torch.float64).to(dev)
v = torch.tensor(1.0, dtype=torch.float64, requires_grad=True)
s = torch.tensor(1.0, dtype=torch.float64, requires_grad=True)
nz = torch.tensor(0.1, dtype=torch.float64, requires_grad=True)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lp = st.GP(v * st.EQ().stretch(s))(x, nz.to(dev)).logpdf(y)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    lp.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"n={n} rep{rep}: forward {1e3*(t1-t0):.1f} ms  backward {1e3*(t2-t1):.1f} ms  grads {float(v.grad):.4f} {float(s.grad):.4f} {float(nz.grad):.4f}")
    v.grad = s.grad = nz.grad = None
