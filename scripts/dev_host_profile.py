"""Host-side cost of one cfg2 eval on the GPU box (development): where Python spends its time between the device launches.

    python scripts/dev_host_profile.py [posterior-first|logpdf-first]

Prints (a) wall time per phase with a device synchronisation in front of each phase -- what the HOST needs to enqueue it --, (b) a
cProfile of ten steps."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stheno_amd as st  # noqa: E402
from bench import NOISE, make_inputs  # noqa: E402
from stheno_amd.matrix import forget_scan  # noqa: E402

order = sys.argv[1] if len(sys.argv) > 1 else "posterior-first"
dev = torch.device("cuda")
w, t = make_inputs("dense_f64", dev)
kernel = st.EQ()


def sync():
    torch.cuda.synchronize()


def step(timing=None):
    def mark(name, t0):
        if timing is not None:
            timing.setdefault(name, []).append(time.perf_counter() - t0)
    forget_scan(t["y"])
    t0 = time.perf_counter()
    f = st.GP(kernel)
    fdd = f(t["x"], NOISE)
    mark("construct", t0)
    if order == "posterior-first":
        t0 = time.perf_counter(); post = f | (fdd, t["y"]); mark("condition (host)", t0)
        t0 = time.perf_counter(); mean, var = post(t["xs"]).marginals(); mark("marginals incl. info read (host blocks on the device)", t0)
        if timing is not None:
            sync()
        t0 = time.perf_counter(); lp = fdd.logpdf(t["y"]); mark("logpdf enqueue (device idle in front: pure host cost)", t0)
    else:
        t0 = time.perf_counter(); lp = fdd.logpdf(t["y"]); mark("logpdf incl. info read", t0)
        if timing is not None:
            sync()
        t0 = time.perf_counter(); post = f | (fdd, t["y"]); mean, var = post(t["xs"]).marginals(); mark("condition + marginals enqueue (pure host cost)", t0)
    return lp, mean, var


for _ in range(3):
    step()
sync()
timing = {}
for _ in range(10):
    step(timing)
    sync()
for k, v in timing.items():
    print(f"{k:90s} {1e3 * sum(v) / len(v):8.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step()
sync()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
