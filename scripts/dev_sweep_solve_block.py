"""Development aid: time the many-right-hand-side triangular solve for several merged
diagonal-block sizes (and report the deviation from the 128-block result)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stheno_amd as st
from stheno_amd import matrix, ops

n = int(sys.argv[1]); nrhs = int(sys.argv[2]); dt = torch.float64 if sys.argv[3] == "f64" else torch.float32
sbs = [int(s) for s in sys.argv[4:]]
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)
x = torch.randn(n, 8, generator=g, dtype=dt).to(dev)
be = ops.get_backend()
k = be.kmat(ops.KTerms([("eq", 1.0, 1.0)]), x, None, lower=True, diag_add=0.1 if dt == torch.float64 else 1e-3)
chol = matrix.Chol.factor_(k).check()
b0 = torch.randn(n, nrhs, generator=g, dtype=dt).to(dev)
ref = None
for sb in sbs:
    matrix._solve_block = lambda n_, r_, f_=True, sb=sb: sb
    chol._dinv_sb = {128: chol.dinv}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    chol._blocks(nrhs)
    torch.cuda.synchronize(); tm = time.perf_counter() - t0
    best = 1e9
    for rep in range(3):
        b = b0.clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        chol.solve_(b)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    if ref is None:
        ref = b
    err = float((b - ref).abs().max() / ref.abs().max())
    print(f"n={n} nrhs={nrhs} {sys.argv[3]} sb={sb}: merge {1e3*tm:.2f} ms solve {1e3*best:.2f} ms  "
          f"{n*n*nrhs/best/1e12:.1f} TFLOP/s  dev vs first {err:.2e}")
