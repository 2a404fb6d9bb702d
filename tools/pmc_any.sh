#!/bin/bash
# tools/pmc_any.sh TAG KERNEL_SUBSTR "COUNTERS..." -- BENCH_ARGS...   one rocprofv3 PMC pass of bench.py, mean per kernel
set -u
TAG=$1; KSUB=$2; CTRS=$3; shift 3; [ "$1" == "--" ] && shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --no-cpu-baseline --steps 4 --warmup 2 "$@" > "$OUT/bench.json" 2> "$OUT/err.txt"
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if ksub in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print(' '.join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(agg.items())))
PY
