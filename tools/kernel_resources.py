#!/usr/bin/env python3
"""Per-kernel register / spill / LDS figures of one source file, from hipcc's own -Rpass-analysis=kernel-resource-usage
(the code-object view; rocprofv3's dispatch columns are not the place to read them).

    python tools/kernel_resources.py dsp_amd/csrc/mfcc_kernels.hip [substring]
"""
import re
import subprocess
import sys

KEYS = ("VGPRs", "AGPRs", "VGPRs Spill", "SGPRs", "SGPRs Spill", r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]")


def main():
    src, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    extra = sys.argv[3:]
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-Wno-unused-value",
                        "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra, capture_output=True, text=True)
    cur, rows = None, {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
        for key in KEYS:
            m = re.search(r"remark:\s+" + key + r": (\d+)", line)
            if m and cur:
                rows[cur][key.replace("\\", "")] = int(m.group(1))
    if r.returncode:
        print(r.stderr[-3000:])
    names = list(rows)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines() if names else []
    for k, d in zip(names, dem):
        d = re.sub(r"\(.*", "", d)
        if sub in d:
            v = rows[k]
            g = lambda key: v.get(key, -1)      # noqa: E731
            print(f"{d[:110]:110s} vgpr {g('VGPRs'):3d} spill {g('VGPRs Spill'):2d} sgpr {g('SGPRs'):3d} sspill {g('SGPRs Spill'):2d} "
                  f"scratch {g('ScratchSize [bytes/lane]'):3d} occ {g('Occupancy [waves/SIMD]')} lds {g('LDS Size [bytes/block]')}")


if __name__ == "__main__":
    main()
