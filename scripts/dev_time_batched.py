"""Development aid: wall-clock breakdown of one batched logpdf step."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stheno_amd as st
from bench import make_inputs, NOISE
from stheno_amd import ops
from stheno_amd.matrix import Chol

dev = torch.device("cuda")
st.B.epsilon = 1e-6
w, t = make_inputs("batched_f32", dev)
be = ops.get_backend()
terms = ops.KTerms([("eq", 1.0, 1.0)])

def sync():
    torch.cuda.synchronize()
    return time.perf_counter()

for rep in range(3):
    t0 = sync()
    k = be.kmat(terms, t["x"], None, lower=True, diag_add=NOISE + 1e-6)
    t1 = sync()
    c = Chol.factor_(k)
    t2 = sync()
    ld = c.logdet()
    t3 = sync()
    v = c.solve(t["y"])
    t4 = sync()
    _, ss = be.colreduce(v, want_ss=True)
    t5 = sync()
    print(f"rep{rep}: kmat {1e3*(t1-t0):.2f}  potrf {1e3*(t2-t1):.2f}  logdet {1e3*(t3-t2):.2f}  trsv {1e3*(t4-t3):.2f}  colreduce {1e3*(t5-t4):.2f} ms;"
          f" device allocs {torch.cuda.memory_stats()['num_device_alloc']}")
    del k, c, v
f = st.GP(st.EQ())
for rep in range(3):
    t0 = sync()
    lp = f(t["x"], NOISE).logpdf(t["y"])
    t1 = sync()
    print(f"api rep{rep}: {1e3*(t1-t0):.2f} ms; device allocs {torch.cuda.memory_stats()['num_device_alloc']}")
