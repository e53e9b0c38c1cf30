"""Samples the GPU's shader clock and socket power while a command runs (an independent
witness for "the gap to the nominal peak is the clock the chip grants a dense MFMA stream").

    python scripts/clock_trace.py OUT.json -- python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-batched-record

Sources, first one that works: the amdsmi Python binding (gfx clock + socket power), else sysfs (pp_dpm_sclk's starred level /
hwmon power1_average|power1_input), else `rocm-smi --showclocks --showpower --json` (slow: ~0.2 s per sample).  Ordinary-user
reads only; nothing is set.  Output: the samples (t, sclk MHz, W), their summary while the device was busy (power above the idle
floor), and the wrapped command's last JSON line if it printed one."""
import glob
import json
import os
import re
import subprocess
import sys
import threading
import time


def _amdsmi_source():
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[0]

        def read():
            sclk = watts = None
            try:
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                sclk = m.get("current_gfxclk") or m.get("average_gfxclk_frequency")
                if isinstance(m.get("current_gfxclks"), (list, tuple)):
                    v = [x for x in m["current_gfxclks"] if isinstance(x, (int, float)) and 0 < x < 10000]
                    if v:
                        sclk = sum(v) / len(v)
                watts = m.get("current_socket_power") or m.get("average_socket_power")
            except Exception:
                pass
            if sclk is None:
                c = amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)
                sclk = c.get("clk") or c.get("cur_clk")
            if watts is None:
                p = amdsmi.amdsmi_get_power_info(h)
                watts = p.get("current_socket_power") or p.get("average_socket_power")
            return sclk, watts
        read()
        return "amdsmi", read
    except Exception:
        return None


def _sysfs_source():
    cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
    if not cards:
        return None
    dev = os.path.dirname(cards[0])
    pw = sorted(glob.glob(dev + "/hwmon/hwmon*/power1_average") + glob.glob(dev + "/hwmon/hwmon*/power1_input"))

    def read():
        sclk = watts = None
        try:
            for line in open(dev + "/pp_dpm_sclk"):
                if "*" in line:
                    sclk = float(re.search(r"(\d+)\s*Mhz", line, re.I).group(1))
        except Exception:
            pass
        try:
            if pw:
                watts = float(open(pw[0]).read()) / 1e6
        except Exception:
            pass
        return sclk, watts
    if read() == (None, None):
        return None
    return "sysfs " + dev, read


def _cli_source():
    def read():
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        sclk = watts = None
        for k, v in card.items():
            if "sclk" in k.lower() and sclk is None:
                m = re.search(r"(\d+)", str(v))
                sclk = float(m.group(1)) if m else None
            if "power" in k.lower() and "socket" in k.lower() or "Average Graphics Package Power" in k:
                try:
                    watts = float(v)
                except Exception:
                    pass
        return sclk, watts
    try:
        read()
        return "rocm-smi", read
    except Exception:
        return None


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    src = _amdsmi_source() or _sysfs_source() or _cli_source()
    samples, stop = [], threading.Event()

    def loop():
        while not stop.is_set():
            t = time.time()
            try:
                s, w = src[1]()
            except Exception:
                s = w = None
            samples.append((t, s, w))
            time.sleep(0.02)

    th = None
    if src is not None:
        th = threading.Thread(target=loop, daemon=True)
        th.start()
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True)
    t1 = time.time()
    stop.set()
    if th is not None:
        th.join(timeout=5)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{"):
            line = ln
    rec = {"command": " ".join(cmd), "source": src[0] if src else None, "returncode": p.returncode, "seconds": t1 - t0,
           "n_samples": len(samples)}
    ws = [w for _, _, w in samples if w]
    if ws:
        floor = min(ws)
        busy = [(s, w) for _, s, w in samples if w and s and w > floor + 0.5 * (max(ws) - floor)]
        rec["idle_floor_w"], rec["max_w"] = floor, max(ws)
        if busy:
            ss = sorted(s for s, _ in busy)
            rec["busy"] = {"n": len(busy), "sclk_mhz_mean": sum(ss) / len(ss), "sclk_mhz_min": ss[0], "sclk_mhz_p10": ss[len(ss) // 10],
                           "sclk_mhz_median": ss[len(ss) // 2], "sclk_mhz_max": ss[-1], "watts_mean": sum(w for _, w in busy) / len(busy)}
    rec["samples_every_10th"] = [(round(t - t0, 3), s, w) for t, s, w in samples[::10]]
    if line:
        try:
            rec["bench_line"] = json.loads(line)
        except Exception:
            rec["bench_line_raw"] = line
    if p.returncode != 0:
        rec["stderr_tail"] = p.stderr[-2000:]
    with open(out_path, "w") as f:
        json.dump(rec, f)
    print(json.dumps({k: v for k, v in rec.items() if k not in ("samples_every_10th", "bench_line")}))


if __name__ == "__main__":
    main()
