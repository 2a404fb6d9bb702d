#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r3b/tests47.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3b/tests47.log | cut -c1-300
[ $rc -eq 0 ] || { tail -40 gpurun_out/r3b/tests47.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b/f64_prof5 -- python3 $GRAFT_REPO_ROOT/bench.py --workload classify_f64 --no-cpu-baseline --steps 30 --settle 0 > $GRAFT_REPO_ROOT/gpurun_out/r3b/f64_prof5.json 2>/dev/null; echo "prof rc=$?"
find $GRAFT_REPO_ROOT/gpurun_out/r3b/f64_prof5 -name "*kernel_stats.csv" | head -1 | xargs -r head -4 | cut -c1-60,150-260
cd $GRAFT_REPO_ROOT && python bench.py --workload classify_f64 --no-cpu-baseline --steps 50 > gpurun_out/r3b/f64_bench_final.json 2>/dev/null; cut -c1-300 gpurun_out/r3b/f64_bench_final.json
