"""SQ counters of one native self-test command (run ON the GPU box, from /tmp):

    python $REPO/scripts/dev_sq_counters.py OUT.txt -- $REPO/stheno_amd/csrc/gpk_selftest --gemm f64 15360 15360 1024 1

One rocprofv3 pass with 8 SQ counters (+ --kernel-trace only); prints, per kernel name, the sums and the ratios
MFMA-busy / busy cycles, wave-parked / wave cycles, issue-stalled / wave cycles.
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES cycles
(/opt/skills/guides/MI355X_MICROARCH.md)."""
import csv, glob, os, subprocess, sys

out = sys.argv[1]
cmd = sys.argv[sys.argv.index("--") + 1:]
counters = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
            "SQ_WAIT_INST_LDS", "SQ_INSTS_VALU"]
d = "/tmp/sqc_%d" % os.getpid()
subprocess.run(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd,
               check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
    a = acc.setdefault(name, {})
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    a["_n"] = a.get("_n", 0) + (1 if r["Counter_Name"] == counters[0] else 0)
with open(out, "a") as fh:
    fh.write("command: " + " ".join(cmd[1:]) + "\n")
    for name, a in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:4]:
        wc = a.get("SQ_WAVE_CYCLES", 0) or 1
        line = (f"  {name[:60]:60s} launches {a['_n']:3d}  mfma_busy/busy {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (a.get('SQ_BUSY_CYCLES', 0) or 1):.3f}"
                f"  parked/wave {a.get('SQ_WAIT_ANY', 0) / wc:.3f}  issue-stall/wave {a.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}"
                f"  active/wave {a.get('SQ_ACTIVE_INST_ANY', 0) / wc:.3f}  lds-stall/wave {a.get('SQ_WAIT_INST_LDS', 0) / wc:.3f}"
                f"  | raw " + " ".join(f"{k}={v:.4g}" for k, v in a.items() if k != '_n'))
        fh.write(line + "\n")
        print(line)
