#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_classify.py -m gpu -x -q > gpurun_out/r3b/tests54.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -1 gpurun_out/r3b/tests54.log | cut -c1-300
[ $rc -eq 0 ] || { tail -60 gpurun_out/r3b/tests54.log; exit 1; }
for v in 1 0 1 0 1 0; do
DSP_AMD_CLASSIFY_FULL_MAPS=$v python bench.py --workload classify --no-cpu-baseline --steps 300 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('classify full_maps=$v %.4f ms (min %.4f median %.4f)' % (r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median']))"; done
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
DSP_AMD_CLASSIFY_FULL_MAPS=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b/cls_prof_$v -- python3 $GRAFT_REPO_ROOT/bench.py --workload classify --no-cpu-baseline --steps 300 > /dev/null 2>&1; echo "prof $v rc=$?"
done
