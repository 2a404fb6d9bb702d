#!/bin/bash
# tools/run_round2.sh TAG -- on the GPU box: the full GPU test tier, the default bench line, the side workloads, the
# kernel-trace summaries and the PMC traffic of every workload.  Everything lands under gpurun_out/round_TAG/.
set -u
TAG=${1:-x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round_$TAG; mkdir -p $OUT
cd $R
python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; tail -5 $OUT/tests.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json
for w in clips config3 config5 config5_2048 classify pcm16; do
    python bench.py --workload $w --no-cpu-baseline --steps 50 >> $OUT/side_workloads.jsonl 2>> $OUT/side.err
done
cut -c1-120 $OUT/side_workloads.jsonl
for w in frames classify config3; do
    ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline --no-config4 --steps 50 > $OUT/trace_$w.json 2> $OUT/trace_$w.err )
done
for w in frames clips config3 config5 config5_2048 classify; do
    python tools/traffic.py $TAG $w 2>&1 | tail -1
done
find $OUT -name "*kernel_stats.csv"
