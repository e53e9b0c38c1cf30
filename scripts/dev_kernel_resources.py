"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stdin or file): one line per kernel.

usage: hipcc ... -c x.hip -Rpass-analysis=kernel-resource-usage 2>&1 | python scripts/dev_kernel_resources.py [filter]
"""
import re
import subprocess
import sys

txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur, rows = None, {}
keys = ["TotalSGPRs", "VGPRs", "AGPRs", r"ScratchSize \[bytes/lane\]", r"LDS Size \[bytes/block\]", "VGPRs Spill",
        r"Occupancy \[waves/SIMD\]"]
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
    for k in keys:
        m = re.search(r"\s" + k + r": (\d+)", line)
        if m and cur:
            rows[cur][k.replace("\\", "").replace(" [bytes/lane]", "").replace(" [bytes/block]", "").replace(" [waves/SIMD]", "").replace(" ", "")] = m.group(1)
names = list(rows)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n") if names else []
for mangled, name in zip(names, dem):
    if flt in name:
        name = name.replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*\)$", "", name)
        print(f"{name[:80]:80s}", " ".join(f"{k}={v}" for k, v in rows[mangled].items()))
