#!/usr/bin/env python3
"""HBM traffic per step of a bench.py workload from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a
pass on gfx950: MI355X_MICROARCH.md, "rocprofv3 PMC slots").  Runs ON THE GPU BOX:

    python tools/traffic.py TAG WORKLOAD [bench args...]      e.g.  python tools/traffic.py r02a classify --clips 49152

Each pass is `rocprofv3 --kernel-trace --pmc <counter> -- python3 bench.py --workload W --settle 0 ...` (the program itself
behind `--`, counters in their own run).  Per step = sum over every dispatch of this library's kernels (dsp::*) / the number
of step() calls the bench made.  Corrections as that guide prescribes: both counters are in KB; FETCH_SIZE x 2 on gfx950
(it tallies 128-byte requests as 64 bytes for wide coalesced reads -- every kernel here loads 8 or 16 bytes per lane).
Writes gpurun_out/traffic_TAG_WORKLOAD.json (copy it to profiles/ to have bench.py quote it as roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, workload, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    out = os.path.join(ROOT, "gpurun_out", f"traffic_{tag}_{workload}")
    os.makedirs(out, exist_ok=True)
    res, per_kernel, line = {}, {}, None
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, ctr)
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"),
               "--workload", workload, "--no-cpu-baseline", "--no-config4", "--settle", "0", "--steps", "4", "--warmup", "4"] + extra
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=420)
        open(os.path.join(out, ctr + ".err"), "w").write(r.stderr[-20000:])
        line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        calls = line["step_calls"]
        tot, by = 0.0, collections.defaultdict(float)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == ctr and ("dsp::" in row["Kernel_Name"] or "_ZN3dsp" in row["Kernel_Name"]):
                    v = float(row["Counter_Value"])
                    tot += v
                    by[row["Kernel_Name"].split("(")[0][:70]] += v
        res[ctr] = tot / calls
        per_kernel[ctr] = {k: v / calls for k, v in by.items()}
    fetch, write = res["FETCH_SIZE"] * 1024 * 2, res["WRITE_SIZE"] * 1024
    alg = line["roofline"]["algorithmic_bytes_per_launch"]
    units = {"frames": line["config"].get("frames_per_gpu")}.get(workload)
    doc = {"workload": workload, "units_per_launch": units if units else round(alg / {"clips": 69096, "classify": 64004, "config3": 4148,
                                                                                      "config5": 64008, "config5_2048": 64008, "pcm16": 1076, "stop": 64004,
                                                                                      "classify_f64": 128004, "classify_pcm16": 32004, "classify_f64_pcm16": 32004}[workload]),
           "hbm_bytes_per_launch": fetch + write, "read_bytes": fetch, "write_bytes": write, "algorithmic_bytes_per_launch": alg,
           "ratio_to_algorithmic": (fetch + write) / alg, "fetch_size_kb": res["FETCH_SIZE"], "write_size_kb": res["WRITE_SIZE"],
           "correction": "FETCH_SIZE x2 (gfx950: 128-byte requests tallied as 64), both counters in KB; separate --pmc passes; "
                         "sum over the dsp::* dispatches of a step", "step_calls_per_pass": line["step_calls"],
           "per_kernel_kb_per_step": per_kernel, "command": "bench.py --workload " + workload + " " + " ".join(extra)}
    path = os.path.join(ROOT, "gpurun_out", f"traffic_{tag}_{workload}.json")
    json.dump(doc, open(path, "w"), indent=1)
    print(f"{workload}: read {fetch / 1e9:.4f} GB + write {write / 1e9:.4f} GB = {(fetch + write) / 1e9:.4f} GB per step, "
          f"algorithmic {alg / 1e9:.4f} GB, ratio {(fetch + write) / alg:.3f}  -> {path}")


if __name__ == "__main__":
    main()
