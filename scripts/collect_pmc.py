"""HBM traffic per kernel from rocprofv3 PMC passes over bench.py (run ON the GPU box):

    cd /tmp && export TMPDIR=/tmp
    python $REPO/scripts/collect_pmc.py dense_f64 $REPO/gpurun_out/r01_pmc_dense_f64.json

Two separate passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`: the TCC counters do not fit one pass), with
`--kernel-trace` only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; counter unit KB;
FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B).  The result is per kernel name:
launches, raw/corrected read bytes, write bytes, HBM bytes per launch."""
import csv, glob, json, os, subprocess, sys

workload, out_path = sys.argv[1], sys.argv[2]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = [sys.executable, os.path.join(repo, "bench.py"), "--workload", workload, "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-batched-record"]


def one_pass(counter):
    d = f"/tmp/pmc_{workload}_{counter}"
    # (no check=True: the counters are written before the wrapped process tears down)
    subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd,
                   check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
        a = acc.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return acc


fetch, write = one_pass("FETCH_SIZE"), one_pass("WRITE_SIZE")
kernels = {}
for name, (launches, kb) in fetch.items():
    wkb = write.get(name, [launches, 0.0])[1]
    kernels[name] = {
        "launches": launches, "fetch_bytes_raw": kb * 1024.0, "fetch_bytes_corrected": 2 * kb * 1024.0,
        "write_bytes": wkb * 1024.0, "hbm_bytes_per_launch": (2 * kb + wkb) * 1024.0 / launches,
    }
json.dump({
    "source": "scripts/collect_pmc.py: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- " + " ".join(cmd[1:]),
    "note": "counter unit KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); "
            "4 evals per pass (1 warm-up + 1 timed + 2 under the HIP-event hooks)",
    "evals": 4, "kernels": kernels}, open(out_path, "w"), indent=1)
top = sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:6]
for name, k in top:
    print(f"{name[:70]:70s} launches {k['launches']:5d}  HBM/launch {k['hbm_bytes_per_launch']/1e6:10.1f} MB")
