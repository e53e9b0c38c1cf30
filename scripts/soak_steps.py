"""Soak: every bench workload's step repeated, every output compared BIT FOR BIT with the first repetition's -- the right-hand sides that
ride through the factorisations run on side streams (one matrix) or inside the batched launches, and the pseudo-point cross-covariance is
built beside a factorisation: nothing about a result may depend on how those overlaps happen to interleave.

usage: python scripts/soak_steps.py [reps for dense_f64 (200)] -> one JSON line"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bench import make_inputs, make_step  # noqa: E402

reps0 = int(sys.argv[1]) if len(sys.argv) > 1 else 200
plan = [("dense_f64", "posterior-first", reps0), ("dense_f64", "logpdf-first", reps0 // 2), ("batched_f32", "posterior-first", reps0 * 2),
        ("sum_f32", "posterior-first", max(reps0 // 5, 10)), ("sparse_f32", "posterior-first", reps0 // 2)]
out = []
for wl, order, reps in plan:
    bench.ORDER = order
    w, t = make_inputs(wl, "cuda")
    step = make_step(wl, w, t)
    r = step()
    ref = [torch.as_tensor(v).clone() for v in (r if isinstance(r, (tuple, list)) else [r])]
    torch.cuda.synchronize()
    bad = 0
    t0 = time.perf_counter()
    for _ in range(reps):
        r = step()
        got = [torch.as_tensor(v) for v in (r if isinstance(r, (tuple, list)) else [r])]
        bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(got, ref))
    torch.cuda.synchronize()
    out.append({"workload": wl, "order": order, "repetitions": reps, "outputs_that_differed": bad,
                "seconds": round(time.perf_counter() - t0, 2)})
    del w, t, step, r, ref
    torch.cuda.empty_cache()
print(json.dumps({"soak": out, "all_bit_identical": all(o["outputs_that_differed"] == 0 for o in out)}))
