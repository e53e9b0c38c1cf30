"""Development aid: one logpdf + posterior at a size far beyond the benchmark configs (HBM is 288 GB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stheno_amd as st

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
x = torch.randn(n, 8, generator=g, dtype=torch.float64).to(dev)
y = torch.randn(n, 1, generator=g, dtype=torch.float64).to(dev)
xs = torch.randn(1024, 8, generator=g, dtype=torch.float64).to(dev)
res = {}
for dt, eps in (((torch.float32, 1e-6),) if (len(sys.argv) > 2 and sys.argv[2] == "f32") else ((torch.float32, 1e-6), (torch.float64, 1e-12))):
    st.B.epsilon = eps
    f = st.GP(st.EQ())
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); t0 = time.perf_counter()
    fdd = f(x.to(dt), 0.1)
    lp = fdd.logpdf(y.to(dt))
    mean, var = (f | (fdd, y.to(dt)))(xs.to(dt)).marginals()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    res[dt] = (float(lp), mean.double(), var.double())
    print(f"N={n} {dt}: logpdf {float(lp):.6f}  {t1 - t0:.2f} s  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB  "
          f"POTRF-equivalent {n**3 / 3 / (t1 - t0) / 1e12:.1f} TFLOP/s incl. everything", flush=True)
    del fdd, lp, mean, var, f
if torch.float64 not in res: sys.exit(0)
a, b = res[torch.float32], res[torch.float64]
rel = lambda u, v: float((u - v).abs().max() / v.abs().max())
print(f"fp32 vs fp64: logpdf rel {abs(a[0] - b[0]) / abs(b[0]):.2e}  mean rel {rel(a[1], b[1]):.2e}  var rel {rel(a[2], b[2]):.2e}")
