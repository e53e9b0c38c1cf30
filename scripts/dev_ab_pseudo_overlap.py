"""A/B on ONE box: cfg5's step with the cross-covariance built on a side stream beside the factorisation of K_z
(matrix.config.pseudo_overlap_build) and in stream order.  usage: python scripts/dev_ab_pseudo_overlap.py [steps (10)] [reps (3)]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_inputs, make_step  # noqa: E402
from stheno_amd import matrix  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w, t = make_inputs("sparse_f32", "cuda")
step = make_step("sparse_f32", w, t)
out = {True: [], False: []}
vals = {}
for on in (True, False):
    matrix.config.pseudo_overlap_build = on
    for _ in range(3):
        r = step()
    vals[on] = [float(torch.as_tensor(v).double().sum()) for v in (r if isinstance(r, (tuple, list)) else [r])]
for rep in range(reps):
    for on in (True, False):
        matrix.config.pseudo_overlap_build = on
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        out[on].append((time.perf_counter() - t0) * 1e3 / steps)
print(json.dumps({"workload": "sparse_f32", "steps": steps, "side_stream_build_ms": out[True], "stream_order_ms": out[False],
                  "checks_side": vals[True], "checks_order": vals[False]}))
