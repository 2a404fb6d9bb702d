#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into one small text summary for profiles/.

    python tools/summarize_profile.py gpurun_out/prof_TAG profiles/rNN_TAG_rocprof.txt
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = [f"rocprofv3 summary of {src} (bench.py, BASELINE config 2: 1 M x 512 fp32 frames per launch)", ""]
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        lines.append("== kernel-trace --stats (kernel_stats.csv; durations in ns)")
        for r in csv.DictReader(open(f)):
            name = r["Name"]
            name = name if len(name) < 90 else name[:87] + "..."
            lines.append(f"{name:90s} calls={r['Calls']:>4s} avg_ns={float(r['AverageNs']):12.0f} min_ns={r['MinNs']:>9s} max_ns={r['MaxNs']:>9s} pct={r['Percentage']}")
    tj = os.path.join(src, "trace.json")
    if os.path.exists(tj) and os.path.getsize(tj):
        try:
            d = json.loads(open(tj).read().strip().splitlines()[-1])
            lines += ["", f"bench line under the tracer: value={d['value']:.4g} {d['unit']}, ms_per_step={d['ms_per_step']:.4f}, kernel_ms={d['roofline']['kernel_ms']:.4f}"]
        except Exception as e:  # noqa: BLE001
            lines.append(f"(trace.json unreadable: {e})")
    lines += ["", "== PMC passes (one group per run; mean over the dispatches of mfcc512_wave_kernel)"]
    vals = {}
    meta = None
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mfcc512" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta = meta or {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r}
        for k, v in agg.items():
            vals[k] = sum(v) / len(v)
            lines.append(f"{k:28s} {vals[k]:14.6g}   (n={len(v)})")
    if meta:
        lines += ["", f"dispatch: {meta}"]
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        fetch = vals["FETCH_SIZE"] * 1024 * 2     # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads
        write = vals["WRITE_SIZE"] * 1024
        lines += ["", "== HBM traffic per launch (MI355X_MICROARCH.md: FETCH_SIZE x2 on gfx950, KB units)",
                  f"read  {fetch / 1e9:.4f} GB   write {write / 1e9:.4f} GB   total {(fetch + write) / 1e9:.4f} GB   algorithmic 2.1000 GB   ratio {(fetch + write) / 2.1e9:.4f}"]
        json.dump({"hbm_bytes_per_launch": fetch + write, "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
                   "correction": "FETCH_SIZE x2 (gfx950), KB units", "frames_per_launch": 1000000, "source": os.path.basename(dst)},
                  open(dst.replace("_rocprof.txt", "_traffic.json"), "w"))
    if "GRBM_GUI_ACTIVE" in vals:
        lines.append(f"GRBM_GUI_ACTIVE / 8 XCDs = {vals['GRBM_GUI_ACTIVE'] / 8:.4g} cycles per launch")
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
