"""cfg3 at its full size (EQ + Linear, N = 32768, D = 4, fp32, N* = 2048): error of logpdf / posterior mean / variance against fp64 on the
device, by look-ahead block width and solve block (round 3: the 1e-3 bar of north_star at full N)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stheno_amd as st
from stheno_amd import B, matrix
from bench import NOISE, make_inputs

dev = torch.device("cuda")
w, t = make_inputs("sum_f32", dev)


def run(dtype):
    k = st.EQ() + st.Linear()
    f = st.GP(k)
    x, y, xs = (t[n].to(dtype) for n in ("x", "y", "xs"))
    fdd = f(x, NOISE)
    lp = fdd.logpdf(y)
    m, v = (f | (fdd, y))(xs).marginals()
    return float(lp), m.double(), v.double()


B.epsilon = 1e-12
lp64, m64, v64 = run(torch.float64)
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
B.epsilon = 1e-6
# (look-ahead from, outer block, width of the explicit inverses, largest solve block)
CASES = [(7168, 1024, 512, 256), (7168, 1024, 1024, 256), (7168, 512, 512, 256), (7168, 1024, 256, 256), (0, 0, 0, 256)]
if len(sys.argv) > 1 and sys.argv[1] == "all":
    CASES += [(7168, 1024, 1024, 512), (7168, 512, 512, 512), (7168, 1024, 512, 128), (7168, 2048, 2048, 512)]
for la_from, nb, inv, sbmax in CASES:
    matrix.config.potrf_lookahead_from = la_from
    matrix.config.potrf_lookahead_nb[torch.float32] = nb
    matrix.config.potrf_lookahead_inv[torch.float32] = inv
    orig = matrix._solve_block
    matrix._solve_block = lambda n, nrhs, fp64=True, _o=orig, _m=sbmax: min(_o(n, nrhs, fp64), _m) if not fp64 else _o(n, nrhs, fp64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lp32, m32, v32 = run(torch.float32)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    matrix._solve_block = orig
    print(f"FP32ACC look-ahead from {la_from} outer block {nb} explicit inverses {inv} solve block <= {sbmax}: logpdf {abs(lp32 - lp64) / abs(lp64):.2e}  mean {rel(m32, m64):.3e}  var {rel(v32, v64):.2e}   {dt * 1e3:.0f} ms")
