"""HBM-side traffic per kernel of ANY command (run ON the GPU box, from /tmp, TMPDIR=/tmp): the two rocprofv3 PMC passes of
scripts/collect_pmc.py (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--kernel-trace` only; FETCH_SIZE doubled for gfx950) around a native
binary -- A/B runs of the self-test's timing modes with a knob set.

    python $REPO/scripts/pmc_native.py OUT.json -- ./gpk_selftest --set 58 1 --perf-rows f64 16384 2048 1024 0 2
"""
import csv, glob, json, os, subprocess, sys

out_path = sys.argv[1]
cmd = sys.argv[sys.argv.index("--") + 1:]


def one_pass(counter):
    d = "/tmp/pmcn_%d_%s" % (os.getpid(), counter)
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
    kernels[name] = {"launches": launches, "fetch_bytes_corrected": 2 * kb * 1024.0, "write_bytes": wkb * 1024.0,
                     "hbm_bytes_per_launch": (2 * kb + wkb) * 1024.0 / launches}
json.dump({"source": "scripts/pmc_native.py: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- " + " ".join(cmd),
           "note": "counter unit KB; FETCH_SIZE doubled per MI355X_MICROARCH.md", "kernels": kernels}, open(out_path, "w"), indent=1)
for name, k in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:5]:
    print(f"{name[:70]:70s} launches {k['launches']:5d}  HBM-side bytes / launch {k['hbm_bytes_per_launch'] / 1e6:10.1f} MB")
