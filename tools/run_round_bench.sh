#!/bin/bash
# tools/run_round_bench.sh TAG -- on the GPU box: the default bench line (with cpu_baseline), the side workloads, and a
# kernel-trace of the classify workload.  Everything lands under gpurun_out/round_TAG/.
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round_$TAG; mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
for w in clips config3 config5 classify pcm16; do
    python bench.py --workload $w --no-cpu-baseline --steps 50 >> $OUT/side_workloads.jsonl 2>> $OUT/side.err
done
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/classify_trace -- python3 $R/bench.py --workload classify --no-cpu-baseline --steps 50 > $OUT/classify_trace.json 2> $OUT/classify_trace.err )
cut -c1-160 $OUT/bench.json $OUT/side_workloads.jsonl
find $OUT -name "*kernel_stats.csv"
