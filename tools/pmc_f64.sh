#!/bin/bash
# one rocprofv3 --pmc pass over tools/prof_classify_f64.py; prints the counters per dsp:: kernel (averages per dispatch)
# usage (GPU box): bash tools/pmc_f64.sh "COUNTER1 COUNTER2 ..." [clips] [pcm16]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_f64
timeout -k 10 300 rocprofv3 --kernel-trace --pmc $1 --output-format csv -d $R/gpurun_out/pmc_f64 -- python3 $R/tools/prof_classify_f64.py ${2:-49152} $3 > $R/gpurun_out/pmc_f64.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/pmc_f64/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsp::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-44:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k, " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
PY
rm -rf $R/gpurun_out/pmc_f64
