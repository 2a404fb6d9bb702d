#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3b/f64_final; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log | cut -c1-200
cp gpurun_out/gate_report.json $O/gate_report.json 2>/dev/null
python bench.py --workload classify_f64 --no-cpu-baseline --steps 50 > $O/bench.json 2> $O/bench.err; cut -c1-240 $O/bench.json
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --workload classify_f64 --no-cpu-baseline --steps 20 > $O/trace.json 2> $O/trace.err )
find $O/trace -name "*kernel_stats.csv" | head -1 | xargs -r head -7 | cut -c1-60,150-260
python tools/traffic.py r3h classify_f64 2>&1 | tail -1
