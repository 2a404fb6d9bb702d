#!/usr/bin/env python3
"""VALU / LDS busy fractions of a bench.py workload's dominant kernel from one rocprofv3 PMC pass (SQ counters; their own run, with
--kernel-trace only).  Runs ON THE GPU BOX:   python tools/sq_fractions.py TAG WORKLOAD [bench args...]
valu_frac = 4 x SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES (quad-cycles of VALU issue over the cycles the kernel's CUs were busy, 4 SIMDs
per CU); lds_frac = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES; conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  Writes
gpurun_out/sq_TAG_WORKLOAD.json (copy to profiles/<round>_<workload>_sq_counters.json: bench.py quotes it, labelled as replayed)."""
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
    d = os.path.join(ROOT, "gpurun_out", f"sq_{tag}_{workload}")
    ctrs = ["SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS"]
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"),
           "--workload", workload, "--no-cpu-baseline", "--no-config4", "--settle", "0", "--steps", "4", "--warmup", "4"] + extra
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=420)
    open(d + ".err", "w").write(r.stderr[-20000:])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "dsp::" in row["Kernel_Name"] or "_ZN3dsp" in row["Kernel_Name"]:
                agg[row["Kernel_Name"].split("(")[0][:80]][row["Counter_Name"]] += float(row["Counter_Value"])
    kernels = {}
    for k, c in agg.items():
        busy = c.get("SQ_BUSY_CU_CYCLES", 0.0)
        if busy > 0:
            kernels[k] = {"valu_frac": 4.0 * c["SQ_ACTIVE_INST_VALU"] / busy / 4.0, "lds_frac": c["SQ_LDS_IDX_ACTIVE"] / busy,
                          "lds_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0), "busy_cu_cycles": busy, "counters": dict(c)}
    top = max(kernels, key=lambda k: kernels[k]["busy_cu_cycles"]) if kernels else None
    doc = {"workload": workload, "dominant_kernel": top, "valu_frac": kernels[top]["valu_frac"] if top else None,
           "lds_frac": kernels[top]["lds_frac"] if top else None, "per_kernel": kernels,
           "definition": "valu_frac = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (SQ_BUSY_CU_CYCLES x 4 SIMDs); lds_frac = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES; "
                         "one rocprofv3 --kernel-trace --pmc pass of bench.py --workload " + workload + " " + " ".join(extra)}
    path = os.path.join(ROOT, "gpurun_out", f"sq_{tag}_{workload}.json")
    json.dump(doc, open(path, "w"), indent=1)
    print(workload, top, "valu %.3f lds %.3f" % (doc["valu_frac"] or 0, doc["lds_frac"] or 0), "->", path)


if __name__ == "__main__":
    main()
