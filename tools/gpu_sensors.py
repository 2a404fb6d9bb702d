"""Shader clock, package power and temperatures of one GPU, read in-process from sysfs (amdgpu hwmon) -- no subprocess, no
GPU call, microseconds per sample, so bench.py can sample between event records without disturbing the timed region.

The headline kernel runs at the package power cap: its time follows the clock the power manager settles at,
which differs from box to box.  A bench line that carries the clock and the power says whether a slower number is a slower
box or a slower kernel.

    s = Sensors.for_device(0)      # HIP device index -> PCI address -> sysfs
    s.read() -> {"sclk_mhz": 1690, "power_w": 1400.0, "temp_c": {...}} (keys absent when the box does not expose them)
"""
from __future__ import annotations

import glob
import os
import re


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


class Sensors:
    def __init__(self, pci_dir: str | None):
        self.pci_dir = pci_dir
        self.hwmon = None
        if pci_dir:
            h = sorted(glob.glob(os.path.join(pci_dir, "hwmon", "hwmon*")))
            self.hwmon = h[0] if h else None
        self.power_file = None
        if self.hwmon:
            for name in ("power1_input", "power1_average"):      # microwatts
                if _read(os.path.join(self.hwmon, name)) not in (None, ""):
                    self.power_file = os.path.join(self.hwmon, name)
                    break
        self.temp_files = {}
        if self.hwmon:
            for f in sorted(glob.glob(os.path.join(self.hwmon, "temp*_input"))):
                label = _read(f.replace("_input", "_label")) or os.path.basename(f).split("_")[0]
                self.temp_files[label] = f

    @staticmethod
    def for_device(index: int = 0) -> "Sensors":
        """HIP device index -> its PCI directory under /sys/bus/pci/devices (through torch's device properties; the only
        amdgpu card in sysfs when that is not available)."""
        pci = None
        try:
            import torch
            p = torch.cuda.get_device_properties(index)
            addr = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            cand = os.path.join("/sys/bus/pci/devices", addr)
            if os.path.isdir(cand):
                pci = cand
        except Exception:  # noqa: BLE001
            pass
        if pci is None:
            cards = [d for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")) if os.path.exists(os.path.join(d, "pp_dpm_sclk"))]
            if len(cards) == 1:
                pci = os.path.realpath(cards[0])
        return Sensors(pci)

    @property
    def available(self) -> bool:
        return bool(self.pci_dir and (self.hwmon or os.path.exists(os.path.join(self.pci_dir, "pp_dpm_sclk"))))

    def sclk_mhz(self):
        if self.hwmon:
            v = _read(os.path.join(self.hwmon, "freq1_input"))      # Hz
            if v and v.isdigit() and int(v) > 0:
                return int(v) / 1e6
        v = _read(os.path.join(self.pci_dir, "pp_dpm_sclk")) if self.pci_dir else None
        if v:
            for line in v.splitlines():                               # "1: 1690Mhz *"
                if line.rstrip().endswith("*"):
                    m = re.search(r"(\d+)\s*Mhz", line, re.I)
                    if m:
                        return float(m.group(1))
        return None

    def power_w(self):
        v = _read(self.power_file) if self.power_file else None
        return int(v) / 1e6 if v and v.isdigit() else None

    def temps_c(self):
        out = {}
        for label, f in self.temp_files.items():
            v = _read(f)
            if v and v.lstrip("-").isdigit():
                out[label] = int(v) / 1e3
        return out

    def power_cap_w(self):
        v = _read(os.path.join(self.hwmon, "power1_cap")) if self.hwmon else None
        return int(v) / 1e6 if v and v.isdigit() else None

    def read(self) -> dict:
        d = {}
        if not self.pci_dir:
            return d
        s, p, t = self.sclk_mhz(), self.power_w(), self.temps_c()
        if s is not None:
            d["sclk_mhz"] = s
        if p is not None:
            d["power_w"] = p
        if t:
            d["temp_c"] = t
        return d


def summarize(samples: list[dict]) -> dict:
    """min / median / max over the samples of a region, per sensor."""
    import statistics
    out = {"samples": len(samples)}
    for key in ("sclk_mhz", "power_w"):
        v = [s[key] for s in samples if key in s]
        if v:
            out[key] = {"min": min(v), "median": statistics.median(v), "max": max(v)}
    temps = {}
    for s in samples:
        for k, v in s.get("temp_c", {}).items():
            temps.setdefault(k, []).append(v)
    if temps:
        out["temp_c"] = {k: max(v) for k, v in temps.items()}
    return out


if __name__ == "__main__":
    import json
    import sys
    s = Sensors.for_device(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    print(json.dumps({"pci_dir": s.pci_dir, "hwmon": s.hwmon, "power_file": s.power_file, "cap_w": s.power_cap_w(), "now": s.read()}))
