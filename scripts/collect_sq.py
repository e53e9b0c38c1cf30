"""SQ / GRBM counters of the kernels of one bench workload (run ON the GPU box, from /tmp, TMPDIR=/tmp):

    python $REPO/scripts/collect_sq.py dense_f64 $REPO/gpurun_out/r05/r05_sq_dense_f64.json

ONE rocprofv3 pass: `--pmc` with 8 SQ counters + GRBM_GUI_ACTIVE, `--kernel-trace` only (no other trace domain, as gpurun demands).
Per kernel name: launches, summed duration (kernel trace), the raw counter sums and three derived figures:

  clock_ghz   = GRBM_GUI_ACTIVE / (XCDs x duration)   -- the shader clock the kernel actually ran at (DVFS: the chip clocks to its power
                                                          budget, /opt/skills/guides/MI355X_MICROARCH.md "DVFS give-back")
  mfma_util   = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / XCDs)   -- fraction of the matrix pipes' cycles spent in MFMAs
                 at THAT clock (the counter sums busy cycles over every SIMD: 64 per v_mfma_f64_16x16x4, checked against the
                 instruction count in profiles/r02_sq_counters.txt)
  mfma_frac_of_nominal = mfma_util x clock_ghz / 2.4   -- the same against the 2.4 GHz the 78.6 / 157.3 TFLOP/s peaks are quoted at

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (guide); WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY
(issue stall) + ACTIVE_INST_ANY ~ WAVE_CYCLES.  stderr of the profiler is KEPT (next to the output file)."""
import csv, glob, json, os, subprocess, sys

workload, out_path = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = [sys.executable, os.path.join(repo, "bench.py"), "--workload", workload, "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
       "--no-batched-record"] + extra
mops = "SQ_INSTS_VALU_MFMA_MOPS_F64" if workload.endswith("f64") else "SQ_INSTS_VALU_MFMA_MOPS_F32"
counters = ["SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", mops, "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
            "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"]
d = "/tmp/sq_%s_%d" % (workload, os.getpid())
log = out_path + ".log"
with open(log, "w") as lf:
    subprocess.run(["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd,
                   check=False, stdout=lf, stderr=subprocess.STDOUT, timeout=900)
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not cc:
    print("no counter file -- see", log)
    sys.exit(1)


def short(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]


dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r.get("Dispatch_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = {}
seen = {}
for r in csv.DictReader(open(cc[0])):
    name = short(r["Kernel_Name"])
    a = acc.setdefault(name, {"launches": 0, "duration_ns": 0})
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    did = r.get("Dispatch_Id")
    if (name, did) not in seen:
        seen[(name, did)] = 1
        a["launches"] += 1
        a["duration_ns"] += dur.get(did, 0)
        if "Start_Timestamp" in r and "End_Timestamp" in r and did not in dur:
            a["duration_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = {}
for name, a in acc.items():
    gui, ns = a.get("GRBM_GUI_ACTIVE", 0.0), a.get("duration_ns", 0)
    xcds = 8 if (ns and gui / ns > 6.0) else 1          # (the counter is summed over the 8 XCDs' GRBMs when the ratio says 8 clocks)
    clock = gui / xcds / ns if ns else None
    util = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui / xcds) if gui else None
    # GRBM_GUI_ACTIVE also counts what the profiler does around a dispatch (counter set-up, read-back): for launches of a few hundred
    # microseconds the ratio came out at 3.6-4.2 "GHz" on a part whose clock tops out at 2.4.  The estimate is
    # only kept where it can be right: launches of at least 1 ms on average and a result inside the DVFS range; the independent
    # witnesses are scripts/clock_trace.py (amdsmi samples across a run) and gpk_mfma_peak (s_memtime over the 100 MHz wall clock).
    avg_ns = ns / a["launches"] if a.get("launches") else 0
    if clock is not None and (avg_ns < 1.0e6 or not (0.4 <= clock <= 2.45)):
        a["clock_estimate_rejected"] = {"gui_over_duration_ghz": clock, "avg_launch_us": avg_ns / 1e3}
        clock = None
        util = None
    wc = a.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    a.update({
        "xcds_assumed": xcds, "clock_ghz": clock, "mfma_util": util,
        "mfma_frac_of_nominal": (util * clock / 2.4) if (util is not None and clock) else None,
        "parked_per_wave_cycle": a.get("SQ_WAIT_ANY", 0.0) / wc, "issue_stall_per_wave_cycle": a.get("SQ_WAIT_INST_ANY", 0.0) / wc,
        "active_per_wave_cycle": a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, "lds_issue_stall_per_wave_cycle": a.get("SQ_WAIT_INST_LDS", 0.0) / wc,
    })
    rows[name] = a
top = dict(sorted(rows.items(), key=lambda kv: -kv[1].get("duration_ns", 0))[:12])
json.dump({"source": "scripts/collect_sq.py: rocprofv3 --pmc " + " ".join(counters) + " --kernel-trace -- " + " ".join(cmd[1:]),
           "note": "4 evals per pass (1 warm-up + 2 timed + the steps under the HIP-event hooks); profiled passes clock lower than un-profiled ones "
                   "(guide): compare ratios, not wall times", "kernels": top}, open(out_path, "w"), indent=1)
for name, a in list(top.items())[:6]:
    print(f"{name[:64]:64s} n={a['launches']:4d} {a['duration_ns'] / 1e6:9.3f} ms  clock {('%.3f GHz' % a['clock_ghz']) if a['clock_ghz'] else 'n/a (short launches)'}  mfma_util {('%.3f' % a['mfma_util']) if a['mfma_util'] is not None else 'n/a'}"
          f"  parked {a['parked_per_wave_cycle']:.3f} stall {a['issue_stall_per_wave_cycle']:.3f} active {a['active_per_wave_cycle']:.3f}")
