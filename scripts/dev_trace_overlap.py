"""Read a rocprofv3 --kernel-trace CSV and report how kernels of different queues overlap in time.

usage: python scripts/dev_trace_overlap.py <kernel_trace.csv> [skip_first_fraction]
Prints, per queue: launches, busy time, span; the pairwise overlap of queue busy intervals; and
per-kernel-name average durations per queue (so the same kernel alone vs under load can be compared).
"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
rows = list(csv.DictReader(open(path)))
if not rows:
    sys.exit("empty trace")
keys = rows[0].keys()
ks = next(k for k in keys if k.lower().startswith("start"))
ke = next(k for k in keys if k.lower().startswith("end"))
kq = next((k for k in keys if "queue" in k.lower()), None)
kn = next(k for k in keys if "kernel_name" in k.lower() or k.lower() == "name")
ev = []
for r in rows:
    name = re.sub(r"\(.*", "", r[kn].replace("(anonymous namespace)::", "").replace("void ", ""))
    ev.append((int(r[ks]), int(r[ke]), r[kq] if kq else "0", name[:70]))
ev.sort()
t0 = ev[0][0]
# the LAST repetition only: find the last kmat kernel and start there
last = max((i for i, e in enumerate(ev) if "kmat" in e[3]), default=0)
ev = ev[last:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
print(f"window {(t1 - t0) / 1e6:.3f} ms, {len(ev)} kernels")
byq = defaultdict(list)
for s, e, q, n in ev:
    byq[q].append((s, e, n))
def union(iv):
    out, cur = 0, None
    for s, e in sorted(iv):
        if cur is None or s > cur[1]:
            if cur: out += cur[1] - cur[0]
            cur = [s, e]
        else:
            cur[1] = max(cur[1], e)
    if cur: out += cur[1] - cur[0]
    return out
for q, lst in byq.items():
    busy = union([(s, e) for s, e, _ in lst])
    print(f"queue {q}: {len(lst)} kernels, busy {busy / 1e6:.3f} ms, span {(max(e for _, e, _ in lst) - min(s for s, _, _ in lst)) / 1e6:.3f} ms")
qs = list(byq)
for i in range(len(qs)):
    for j in range(i + 1, len(qs)):
        a = [(s, e) for s, e, _ in byq[qs[i]]]
        b = [(s, e) for s, e, _ in byq[qs[j]]]
        ov = union(a) + union(b) - union(a + b)
        print(f"overlap of queues {qs[i]} and {qs[j]}: {ov / 1e6:.3f} ms")
agg = defaultdict(lambda: [0, 0])
for s, e, q, n in ev:
    agg[(q, n)][0] += 1
    agg[(q, n)][1] += e - s
for (q, n), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  q{q} {n:70s} x{c:4d} total {t / 1e6:8.3f} ms avg {t / c / 1e3:8.1f} us")
# timeline of the helper queue relative to the persistent kernels
pers = [(s, e, q) for s, e, q, n in ev if "persist" in n]
for k, (s, e, q) in enumerate(pers[:6]):
    inside = [(s2, e2, n2) for s2, e2, q2, n2 in ev if q2 != q and s2 < e and e2 > s]
    if inside:
        first = min(s2 for s2, _, _ in inside); lastk = max(e2 for _, e2, _ in inside)
        print(f"persistent #{k}: {(e - s) / 1e3:.0f} us; helper-queue kernels inside: {len(inside)}, from +{(first - s) / 1e3:.0f} us to +{(lastk - s) / 1e3:.0f} us")
    else:
        print(f"persistent #{k}: {(e - s) / 1e3:.0f} us; no helper-queue kernel inside")
