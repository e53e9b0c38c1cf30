"""Development aid: fp32 posterior accuracy (vs fp64) as a function of the merged solve-block size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stheno_amd as st
from stheno_amd import B, matrix
from bench import make_inputs, NOISE

dev = torch.device("cuda")
w, t = make_inputs("sum_f32", dev)
n0 = 8192
f = st.GP(st.EQ() + st.Linear())
x32, y32, xs32 = t["x"][:n0], t["y"][:n0], t["xs"]
rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
orig = matrix._solve_block
B.epsilon = 1e-12
fd64 = f(x32.double(), NOISE)
lp64 = fd64.logpdf(y32.double())
m64, v64 = (f | (fd64, y32.double()))(xs32.double()).marginals()
print("var range", float(v64.min()), float(v64.max()))
for sb in (128, 256, 512, 1024, 2048):
    matrix._solve_block = lambda n_, r_, f_=True, sb=sb: sb if r_ > 8 else orig(n_, r_, f_)
    B.epsilon = 1e-6
    fd32 = f(x32, NOISE)
    lp32 = fd32.logpdf(y32)
    m32, v32 = (f | (fd32, y32))(xs32).marginals()
    print(f"sb={sb}: lp rel {abs(float(lp32)-float(lp64))/abs(float(lp64)):.2e} mean rel {rel(m32, m64):.2e} var rel {rel(v32, v64):.2e}")
