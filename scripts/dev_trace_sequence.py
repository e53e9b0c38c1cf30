"""List the kernels of the LAST repetition in a rocprofv3 --kernel-trace CSV, in start order:
start offset, duration, gap to the previous kernel's end, grid size, name.

usage: python scripts/dev_trace_sequence.py <kernel_trace.csv> [marker-kernel-substring (default: kmat)] [k: start at the k-th last marker kernel (default 1)]
"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "kmat"
keys = rows[0].keys()
ks = next(k for k in keys if k.lower().startswith("start"))
ke = next(k for k in keys if k.lower().startswith("end"))
kn = next(k for k in keys if "kernel_name" in k.lower() or k.lower() == "name")
kg = [k for k in keys if k.lower().startswith("grid_size")]
kw = [k for k in keys if k.lower().startswith("workgroup_size")]
ev = []
for r in rows:
    name = re.sub(r"\(.*", "", r[kn].replace("(anonymous namespace)::", "").replace("void ", ""))
    grid = 1
    for g, w in zip(sorted(kg), sorted(kw)):
        grid *= max(1, int(r[g]) // max(1, int(r[w])))
    ev.append((int(r[ks]), int(r[ke]), grid, name[:60]))
ev.sort()
kth = int(sys.argv[3]) if len(sys.argv) > 3 else 1
marks = [i for i, e in enumerate(ev) if marker in e[3]]
last = marks[-kth] if len(marks) >= kth else 0
ev = ev[last:]
t0 = ev[0][0]
prev = t0
tot = {}
for s, e, g, n in ev:
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:6.1f}  wgs {g:7d}  {n}")
    prev = max(prev, e)
    tot[n] = tot.get(n, 0) + (e - s)
print(f"window {(prev - t0) / 1e6:.3f} ms")
for n, t in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"  {t / 1e6:8.3f} ms  {n}")
