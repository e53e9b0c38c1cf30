"""A/B on ONE box: a bench step with the observations riding through the factorisation as a right-hand side
(matrix.config.posterior_rows_rhs -> gpk_potrf_rows_rhs for one matrix, posterior first; matrix.config.logpdf_rhs -> gpk_potrf_rhs
for batches) and with the separate single-column sweep.  Interleaved repetitions.

usage: python scripts/dev_ab_rows_rhs.py [workload (dense_f64)] [steps (20)] [reps (3)] [order (posterior-first | logpdf-first)]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bench import make_inputs, make_step  # noqa: E402
from stheno_amd import matrix  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "dense_f64"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if len(sys.argv) > 4:
    bench.ORDER = sys.argv[4]
w, t = make_inputs(wl, "cuda")
step = make_step(wl, w, t)
out = {True: [], False: []}
vals = {}
for on in (True, False):
    matrix.config.posterior_rows_rhs = matrix.config.logpdf_rhs = on
    for _ in range(3):
        r = step()
    vals[on] = [float(torch.as_tensor(v).double().sum()) for v in (r if isinstance(r, (tuple, list)) else [r])]
for rep in range(reps):
    for on in (True, False):
        matrix.config.posterior_rows_rhs = matrix.config.logpdf_rhs = on
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        out[on].append((time.perf_counter() - t0) * 1e3 / steps)
print(json.dumps({"workload": wl, "order": bench.ORDER, "steps": steps, "rhs_under_the_matrix_ms": out[True], "separate_sweep_ms": out[False],
                  "checks_rhs": vals[True], "checks_separate": vals[False]}))
