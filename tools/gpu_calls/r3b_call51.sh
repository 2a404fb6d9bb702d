#!/bin/bash
mkdir -p gpurun_out/r3b
DSP_AMD_LIB=variants/c3early.so timeout -k 10 900 python -m pytest tests/test_gpu_mfcc.py -m gpu -x -q -k "config3 or prefilter" > gpurun_out/r3b/tests51.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r3b/tests51.log | cut -c1-300
[ $rc -eq 0 ] || { tail -40 gpurun_out/r3b/tests51.log; exit 1; }
for v in head c3early head c3early head c3early; do
if [ $v == head ]; then L=dsp_amd/libdsp_amd.so; else L=variants/$v.so; fi
DSP_AMD_LIB=$L python bench.py --workload config3 --no-cpu-baseline --steps 100 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('config3 $v %.4f ms (min %.4f median %.4f)' % (r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_median']))"; done
