#!/bin/bash
# kernel-trace of one bench workload with a given library build; prints per-kernel averages (us)
# usage (inside gpurun): bash tools/prof_lib.sh LIB TAG WORKLOAD [bench args...]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$1; TAG=$2; WL=$3; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
export DSP_AMD_LIB=$R/$LIB
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o t -- python $R/bench.py --workload $WL --no-config4 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/prof_$TAG/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:7]:
        print("$TAG", r["Name"].split("(")[0][-44:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
