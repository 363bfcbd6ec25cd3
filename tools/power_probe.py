#!/usr/bin/env python3
"""Socket power / shader clock telemetry while a command runs (development aid, round 4; VERDICT r03 "next" #2).

    python tools/power_probe.py [--period 0.1] [--series] [--csv out.csv] -- <command ...>

Reads the amdgpu hwmon files of the first card that has them: power1_input (or power1_average; microwatts), freq1_input
(sclk, Hz), power1_cap, temp*_input, plus gpu_busy_percent.  `amd-smi metric -p -c` prints the same sources.  The module
part (`HwmonSampler`) is what bench.py imports for its `power` object.
"""
import glob
import os
import subprocess
import sys
import threading
import time


def _read_int(path):
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def device_bdf(index=0):
    """PCI address of HIP device `index` ("0000:d9:00.0").  A box shows the hwmon directories of every GPU of the host; only
    this one is the GPU the process computes on."""
    env = os.environ.get("SDRHIP_POWER_BDF")
    if env:
        return env.lower()
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        return "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    except Exception:
        return None


def find_hwmon(bdf=None):
    pattern = f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*" if bdf else "/sys/class/drm/card*/device/hwmon/hwmon*"
    for d in sorted(glob.glob(pattern)):
        for name in ("power1_input", "power1_average"):
            if _read_int(os.path.join(d, name)) is not None:
                return d, os.path.join(d, name)
    return None, None


class HwmonSampler:
    """Background thread: (t, watts, sclk_mhz, busy_percent) every `period` seconds."""

    def __init__(self, period=0.05, bdf=None):
        self.bdf = bdf
        self.dir, self.power_path = find_hwmon(bdf)
        self.period = period
        self.samples = []
        self._stop = threading.Event()
        self._th = None

    @property
    def available(self):
        return self.power_path is not None

    def cap_watts(self):
        v = _read_int(os.path.join(self.dir, "power1_cap")) if self.dir else None
        return None if v is None else v * 1e-6

    def start(self):
        self.samples = []
        self._stop.clear()
        if not self.available:
            return self
        t0 = time.perf_counter()
        busy_path = os.path.join(self.dir, "..", "..", "gpu_busy_percent")
        self.t0 = t0

        def run():
            while not self._stop.is_set():
                p = _read_int(self.power_path)
                f = _read_int(os.path.join(self.dir, "freq1_input"))
                b = _read_int(busy_path)
                self.samples.append((time.perf_counter() - t0, None if p is None else p * 1e-6, None if f is None else f * 1e-6, b))
                self._stop.wait(self.period)

        self._th = threading.Thread(target=run, daemon=True)
        self._th.start()
        return self

    def stop(self):
        self._stop.set()
        if self._th is not None:
            self._th.join()
        return self.samples

    @staticmethod
    def summarize(samples, t_lo=None, t_hi=None):
        sel = [s for s in samples if (t_lo is None or s[0] >= t_lo) and (t_hi is None or s[0] <= t_hi)]
        w = [s[1] for s in sel if s[1] is not None]
        f = [s[2] for s in sel if s[2] is not None]
        if not w:
            return None
        return {"samples": len(w), "mean_w": sum(w) / len(w), "min_w": min(w), "max_w": max(w),
                "mean_sclk_mhz": (sum(f) / len(f)) if f else None, "min_sclk_mhz": min(f) if f else None,
                "max_sclk_mhz": max(f) if f else None}


def main():
    args = sys.argv[1:]
    period, series, csv = 0.1, False, None
    while args and args[0] != "--":
        a = args.pop(0)
        if a == "--period":
            period = float(args.pop(0))
        elif a == "--series":
            series = True
        elif a == "--csv":
            csv = args.pop(0)
    cmd = args[1:]
    s = HwmonSampler(period, device_bdf())
    print(f"hwmon: {s.dir} power file: {s.power_path} cap: {s.cap_watts()} W", flush=True)
    idle = []
    s.start()
    time.sleep(1.0)
    idle = HwmonSampler.summarize(s.stop())
    print("idle (1 s before the command):", idle, flush=True)
    s.start()
    t0 = time.perf_counter()
    r = subprocess.run(cmd)
    dt = time.perf_counter() - t0
    smp = s.stop()
    print(f"command ran {dt:.2f} s, rc {r.returncode}; {len(smp)} samples")
    print("whole run:", HwmonSampler.summarize(smp))
    print("last 60 %:", HwmonSampler.summarize(smp, t_lo=0.4 * dt))
    if series:
        for t, w, f, b in smp:
            print(f"  {t:7.2f} s  {w if w is None else round(w, 1)} W  sclk {f if f is None else round(f)} MHz  busy {b}%")
    if csv:
        with open(csv, "w") as fh:
            fh.write("t_s,socket_power_w,sclk_mhz,gpu_busy_percent\n")
            for t, w, f, b in smp:
                fh.write(f"{t:.3f},{w},{f},{b}\n")
    return r.returncode


if __name__ == "__main__":
    sys.exit(main())
