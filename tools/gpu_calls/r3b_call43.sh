#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 600 python -m pytest tests/test_gpu_scrubjay.py tests/test_gpu_fuzz.py -m gpu -x -q > gpurun_out/r3b/tests43.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -1 gpurun_out/r3b/tests43.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
for v in head k2048sv head k2048sv; do DSP_AMD_LIB=variants/$v.so python bench.py --workload config5_2048 --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('config5_2048 $v %.4f ms' % r['kernel_ms'])"; done
timeout -k 10 300 python bench.py --workload classify_f64 --no-cpu-baseline --steps 30 > gpurun_out/r3b/f64_bench.json 2> gpurun_out/r3b/f64_bench.err; echo "f64 rc=$?"; cut -c1-600 gpurun_out/r3b/f64_bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b/f64_prof -- python3 $GRAFT_REPO_ROOT/bench.py --workload classify_f64 --no-cpu-baseline --steps 30 --settle 0 > /dev/null 2>&1; echo "prof rc=$?"
find $GRAFT_REPO_ROOT/gpurun_out/r3b/f64_prof -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-200
