#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_classify_f64.py -m gpu -x -q > gpurun_out/r3b/tests57.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -1 gpurun_out/r3b/tests57.log | cut -c1-300
[ $rc -eq 0 ] || { tail -60 gpurun_out/r3b/tests57.log; exit 1; }
cd /tmp && export TMPDIR=/tmp
for v in twlds head; do
if [ $v == head ]; then L=$GRAFT_REPO_ROOT/dsp_amd/libdsp_amd.so; else L=$GRAFT_REPO_ROOT/variants/$v.so; fi
export DSP_AMD_LIB=$L
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b/tw_$v -- python3 $GRAFT_REPO_ROOT/bench.py --workload classify_f64 --no-cpu-baseline --steps 100 > /dev/null 2>&1; echo "prof $v rc=$?"
done
