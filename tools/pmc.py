#!/usr/bin/env python3
"""SQ counters of a bench.py workload's dominant kernel, one rocprofv3 --pmc pass per counter group (ON THE GPU BOX):
    python tools/pmc.py TAG WORKLOAD "CTR_A CTR_B" "CTR_C ..." ...
Prints, per group, the per-dispatch mean of each counter over the dsp::* dispatches; writes gpurun_out/pmc_TAG_WORKLOAD.json."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, workload, groups = sys.argv[1], sys.argv[2], sys.argv[3:]
    out = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{workload}")
    os.makedirs(out, exist_ok=True)
    doc = {}
    for gi, grp in enumerate(groups):
        d = os.path.join(out, f"g{gi}")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + grp.split() + ["--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"),
               "--workload", workload, "--no-cpu-baseline", "--no-config4", "--settle", "0", "--steps", "4", "--warmup", "4"]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=420)
        open(os.path.join(out, f"g{gi}.err"), "w").write(r.stderr[-20000:])
        sums, cnt = collections.defaultdict(float), collections.defaultdict(int)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if "dsp::" in row["Kernel_Name"] or "_ZN3dsp" in row["Kernel_Name"]:
                    key = (row["Kernel_Name"].split("(")[0][:60], row["Counter_Name"])
                    sums[key] += float(row["Counter_Value"])
                    cnt[key] += 1
        for (k, c), v in sorted(sums.items()):
            doc.setdefault(k, {})[c] = v / cnt[(k, c)]
            print(f"{k:62s} {c:28s} {v / cnt[(k, c)]:16.1f}  ({cnt[(k, c)]} dispatches)")
    json.dump(doc, open(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{workload}.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
