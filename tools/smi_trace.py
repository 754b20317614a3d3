#!/usr/bin/env python
"""Sample the GPU's shader clock and power while another command runs (back — or drop — the claim
that the tile GEMM runs at a power-limited 1.4-1.65 GHz).

    python tools/smi_trace.py out.json -- python bench.py --no-cpu-baseline

Sources, whichever exist on the box: the amdgpu hwmon / sysfs files of card 0 (freq1_input = sclk in Hz, power1_average or
power1_input in uW, pp_dpm_sclk) every 50 ms, and `amd-smi metric --clock --power --json` about once a second as a cross
check. Writes {"samples": [...], "summary": {...}}; the summary is over the window in which the command was running."""
import glob
import json
import os
import subprocess
import sys
import threading
import time


def _read(p):
    try:
        with open(p) as f:
            return f.read().strip()
    except OSError:
        return None


def sysfs_paths():
    out = {}
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        hw = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
        if not hw:
            continue
        for name in ("freq1_input", "power1_average", "power1_input", "temp1_input"):
            p = os.path.join(hw[0], name)
            if os.path.exists(p):
                out[name] = p
        p = os.path.join(dev, "pp_dpm_sclk")
        if os.path.exists(p):
            out["pp_dpm_sclk"] = p
        break
    return out


def main():
    out_path, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]
    paths = sysfs_paths()
    samples, smi = [], []
    stop = threading.Event()

    def fast():
        while not stop.is_set():
            s = {"t": time.time()}
            for k, p in paths.items():
                v = _read(p)
                if v is None:
                    continue
                if k == "pp_dpm_sclk":
                    cur = [ln for ln in v.splitlines() if "*" in ln]
                    s[k] = cur[0] if cur else v
                else:
                    s[k] = int(v) if v.lstrip("-").isdigit() else v
            samples.append(s)
            time.sleep(0.05)

    def slow():
        while not stop.is_set():
            try:
                r = subprocess.run(["amd-smi", "metric", "-g", "0", "--clock", "--power", "--json"], capture_output=True,
                                   text=True, timeout=10)
                smi.append({"t": time.time(), "out": r.stdout[-4000:]})
            except Exception as e:          # noqa: BLE001
                smi.append({"t": time.time(), "err": repr(e)})
            time.sleep(1.0)

    th = [threading.Thread(target=fast, daemon=True), threading.Thread(target=slow, daemon=True)]
    for t in th:
        t.start()
    time.sleep(0.5)
    t0 = time.time()
    rc = subprocess.call(cmd)
    t1 = time.time()
    time.sleep(0.3)
    stop.set()
    win = [s for s in samples if t0 <= s["t"] <= t1]

    def stat(key, scale):
        v = sorted(s[key] * scale for s in win if isinstance(s.get(key), int))
        if not v:
            return None
        return {"n": len(v), "min": v[0], "p10": v[len(v) // 10], "median": v[len(v) // 2], "p90": v[9 * len(v) // 10],
                "max": v[-1], "mean": sum(v) / len(v)}
    summary = {"command": cmd, "rc": rc, "seconds": t1 - t0, "sysfs": paths,
               "sclk_MHz": stat("freq1_input", 1e-6),
               "power_W": stat("power1_average", 1e-6) or stat("power1_input", 1e-6)}
    with open(out_path, "w") as f:
        json.dump({"summary": summary, "samples": samples, "amd_smi": smi}, f)
    print(json.dumps(summary))
    sys.exit(rc)


if __name__ == "__main__":
    main()
