#!/bin/bash
# tools/run_round4.sh TAG PART -- on the GPU box.  PART a: the full GPU test tier, the bench line as the driver runs it, the side
# workloads.  PART b: rocprofv3 --kernel-trace --stats of the driver's bench command and of the workloads this round changed.
# PART c: PMC traffic (tools/traffic.py) and SQ fractions (tools/sq_fractions.py) of those workloads.  PART d: determinism soak of the
# float64 classifier.  Everything lands under gpurun_out/round_TAG/; the summaries to be judged are copied to profiles/ by hand.
set -u
TAG=${1:-x}; PART=${2:-a}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round_$TAG; mkdir -p $OUT
cd $R
if [ "$PART" == "a" ]; then
    timeout -k 10 1000 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; tail -4 $OUT/tests.log | cut -c1-300
    cp gpurun_out/gate_report.json $OUT/gate_report.json 2>/dev/null
    python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-200 $OUT/bench.json
    for w in clips config3 config5 config5_ragged config5_2048 classify classify_ragged classify_pcm16 classify_f64 classify_f64_pcm16 pcm16 stop; do
        python bench.py --workload $w --no-cpu-baseline --steps 50 >> $OUT/side_workloads.jsonl 2>> $OUT/side.err
    done
    python - <<PY
import json
for l in open("$OUT/side_workloads.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print(d["metric"], "%.4g %s  %.4f ms  frac %.3f" % (d["value"], d["unit"], r["kernel_ms"], r["frac"]), (d.get("sensors") or {}).get("during"))
PY
elif [ "$PART" == "b" ]; then
    ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_frames -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/trace_frames.json 2> $OUT/trace_frames.err )
    for w in classify classify_pcm16 classify_f64 classify_f64_pcm16 classify_ragged config5 config5_ragged config5_2048 stop; do
        ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline --steps 20 > $OUT/trace_$w.json 2> $OUT/trace_$w.err )
    done
    find $OUT -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -6 "$f" | cut -c1-160; done
elif [ "$PART" == "c" ]; then
    for w in frames classify classify_pcm16 classify_f64 classify_f64_pcm16; do
        python tools/traffic.py $TAG $w 2>&1 | tail -1
    done
    for w in frames config5 config5_2048 classify classify_pcm16 classify_f64 classify_f64_pcm16; do
        python tools/sq_fractions.py $TAG $w 2>&1 | tail -1
    done
else
    python tools/soak_classify_f64_full.py 200 2>&1 | tail -1
    DSP_AMD_F64_GUARD=0.9 python tools/soak_classify_f64_full.py 30 2>&1 | tail -1
fi
