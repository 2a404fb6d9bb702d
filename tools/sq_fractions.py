#!/usr/bin/env python3
"""VALU / LDS busy fractions of a bench.py workload's dominant kernel from one rocprofv3 PMC pass (SQ counters; their own run, with
--kernel-trace only).  Runs ON THE GPU BOX:   python tools/sq_fractions.py TAG WORKLOAD [bench args...]
valu_frac = SQ_INSTS_VALU / SQ_BUSY_CU_CYCLES (vector instructions x 4 issue cycles over busy CU cycles x 4 SIMDs); lds_frac = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES; conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.  Writes
gpurun_out/sq_TAG_WORKLOAD.json (copy to profiles/<round>_<workload>_sq_counters.json: bench.py quotes it, labelled as replayed)."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def summarize(d, workload, extra, tag):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "dsp::" in row["Kernel_Name"] or "_ZN3dsp" in row["Kernel_Name"]:
                k = row["Kernel_Name"].split("(")[0][:80]
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[k].add(row["Dispatch_Id"])
    dur = collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "dsp::" in row["Kernel_Name"] or "_ZN3dsp" in row["Kernel_Name"]:
                dur[row["Kernel_Name"].split("(")[0][:80]] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    kernels = {}
    for k, c in agg.items():
        busy = c.get("SQ_BUSY_CU_CYCLES", 0.0)
        if busy > 0:
            n = max(len(launches[k]), 1)
            kernels[k] = {"valu_frac": c["SQ_INSTS_VALU"] / busy, "lds_frac": c["SQ_LDS_IDX_ACTIVE"] / busy,
                          "lds_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1.0),
                          "waves_per_cu": 4.0 * c["SQ_WAVE_CYCLES"] / busy, "launches": n, "ms_per_launch_under_pmc": dur[k] / n / 1e6,
                          "busy_cu_cycles": busy, "counters_per_launch": {a: b / n for a, b in c.items()}}
            if c.get("GRBM_GUI_ACTIVE", 0.0) > 0 and dur[k] > 0:
                # MI355X_MICROARCH.md (DVFS): GRBM_GUI_ACTIVE is summed over the 8 XCDs; / 8 = shader cycles of the dispatch
                cyc = c["GRBM_GUI_ACTIVE"] / 8.0
                kernels[k]["clock_ghz"] = cyc / dur[k]
                kernels[k]["valu_frac_of_wall"] = c["SQ_INSTS_VALU"] / (cyc * 256.0)      # 256 CUs x 4 SIMDs x (cycles / 4) issue slots
                kernels[k]["lds_frac_of_wall"] = c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256.0)
    top = max(kernels, key=lambda k: kernels[k]["busy_cu_cycles"]) if kernels else None
    # the quoted fractions: against the dispatch's own cycles (GRBM_GUI_ACTIVE) when that counter was collected -- SQ_BUSY_CU_CYCLES runs
    # ~7 % short on a kernel that fills every CU for its whole duration (the 512-point frame kernel reads 1.07 against it, 1.00 against the wall)
    pick = lambda k, a: kernels[k].get(a + "_of_wall", kernels[k][a])
    doc = {"workload": workload, "dominant_kernel": top, "valu_frac": pick(top, "valu_frac") if top else None,
           "lds_frac": pick(top, "lds_frac") if top else None, "clock_ghz": kernels[top].get("clock_ghz") if top else None, "per_kernel": kernels,
           "definition": "valu_frac (top level, *_of_wall) = SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 x 256 CUs), lds_frac likewise; per kernel also against SQ_BUSY_CU_CYCLES: a wave64 vector instruction holds its SIMD's issue for 4 cycles and a CU has 4 SIMDs, "
                         "so instructions x 4 / (busy CU cycles x 4) is the share of vector issue slots used (float64 and float32 alike; transcendentals "
                         "hold the pipe longer and make this a lower bound).  SQ_ACTIVE_INST_VALU (quad-cycles a wave is in a vector instruction, "
                         "issue to completion) runs 3 - 12 % above SQ_INSTS_VALU and is not an occupancy of the pipe.  lds_frac = SQ_LDS_IDX_ACTIVE / "
                         "SQ_BUSY_CU_CYCLES; waves_per_cu = 4 x SQ_WAVE_CYCLES (quad-cycles) / SQ_BUSY_CU_CYCLES.  One rocprofv3 --kernel-trace --pmc "
                         "pass of bench.py --workload " + workload + " " + " ".join(extra)}
    path = os.path.join(ROOT, "gpurun_out", f"sq_{tag}_{workload}.json")
    json.dump(doc, open(path, "w"), indent=1)
    print(workload, top, "valu %.3f lds %.3f" % (doc["valu_frac"] or 0, doc["lds_frac"] or 0), "->", path)


def main():
    tag, workload, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    d = os.path.join(ROOT, "gpurun_out", f"sq_{tag}_{workload}")
    if extra and extra[0] == "--recompute":      # from the counter files of an earlier run (no GPU)
        return summarize(d, workload, extra[1:], tag)
    ctrs = ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS",
            "GRBM_GUI_ACTIVE"]
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ctrs + ["--output-format", "csv", "-d", d, "--", "python3", os.path.join(ROOT, "bench.py"),
           "--workload", workload, "--no-cpu-baseline", "--no-config4", "--settle", "0", "--steps", "4", "--warmup", "4"] + extra
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=420)
    open(d + ".err", "w").write(r.stderr[-20000:])
    summarize(d, workload, extra, tag)


if __name__ == "__main__":
    main()
