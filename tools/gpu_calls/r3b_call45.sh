#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_classify_f64.py tests/test_gpu_classify.py tests/test_gpu_scrubjay.py -m gpu -x -q > gpurun_out/r3b/tests45.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3b/tests45.log | cut -c1-300
[ $rc -eq 0 ] || { tail -40 gpurun_out/r3b/tests45.log; exit 1; }
for n in 8192 49152; do
timeout -k 10 300 python bench.py --workload classify_f64 --clips $n --no-cpu-baseline --steps 30 2> gpurun_out/r3b/f64_bench_$n.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('classify_f64 $n clips: %.3f ms  %.3e clips/s frac %.4f' % (r['kernel_ms'], d['value'], r['frac']))"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b/f64_prof3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload classify_f64 --clips 49152 --no-cpu-baseline --steps 30 --settle 0 > /dev/null 2>&1; echo "prof rc=$?"
find $GRAFT_REPO_ROOT/gpurun_out/r3b/f64_prof3 -name "*kernel_stats.csv" | head -1 | xargs -r head -5 | cut -c1-60,150-260
