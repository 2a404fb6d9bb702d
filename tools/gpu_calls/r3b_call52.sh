#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_classify_f64.py -m gpu -x -q > gpurun_out/r3b/tests52.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r3b/tests52.log | cut -c1-300
[ $rc -eq 0 ] || { tail -40 gpurun_out/r3b/tests52.log; exit 1; }
DSP_AMD_LIB=variants/a3.so timeout -k 10 900 python -m pytest tests/test_gpu_classify_f64.py -m gpu -x -q 2>&1 | tail -1
for v in a1 head a3 a1 head a3; do
if [ $v == head ]; then L=dsp_amd/libdsp_amd.so; else L=variants/$v.so; fi
DSP_AMD_LIB=$L python bench.py --workload classify_f64 --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('classify_f64 $v %.4f ms (min %.4f)' % (r['kernel_ms'], r['kernel_ms_min']))"; done
