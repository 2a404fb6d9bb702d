#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_scrubjay.py tests/test_gpu_fuzz.py tests/test_gpu_consumers.py -m gpu -x -q > gpurun_out/r3b/tests42.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -1 gpurun_out/r3b/tests42.log | cut -c1-200
for v in head k2048karg head k2048karg; do DSP_AMD_LIB=variants/$v.so python bench.py --workload config5_2048 --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('config5_2048 $v %.4f ms' % r['kernel_ms'])"; done
