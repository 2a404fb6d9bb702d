#!/bin/bash
# tools/pmc.sh TAG LIB.so "COUNTERS..." -- one rocprofv3 PMC pass of bench.py with a given library build
set -u
TAG=$1; LIB=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export DSP_AMD_LIB=$ROOT/$LIB
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc $@ --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 > "$OUT/bench.json" 2> "$OUT/err.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'mfcc512' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(' '.join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(agg.items())))
for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'mfcc512' in r['Kernel_Name']]
    print(f"kernel avg {sum(d)/len(d)/1e3:.1f} us over {len(d)} launches")
PY
