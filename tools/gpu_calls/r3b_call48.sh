#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_classify_f64.py -m gpu -x -q > gpurun_out/r3b/tests48.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3b/tests48.log | cut -c1-300
[ $rc -eq 0 ] || { tail -40 gpurun_out/r3b/tests48.log; exit 1; }
for v in r2 head r2 head; do
if [ $v == head ]; then L=dsp_amd/libdsp_amd.so; else L=variants/$v.so; fi
DSP_AMD_LIB=$L python bench.py --workload classify_f64 --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('classify_f64 $v %.4f ms' % r['kernel_ms'])"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b/f64_prof6 -- python3 $GRAFT_REPO_ROOT/bench.py --workload classify_f64 --no-cpu-baseline --steps 30 --settle 0 > /dev/null 2>&1; echo "prof rc=$?"
find $GRAFT_REPO_ROOT/gpurun_out/r3b/f64_prof6 -name "*kernel_stats.csv" | head -1 | xargs -r head -5 | cut -c1-60,150-260
