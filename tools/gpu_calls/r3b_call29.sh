#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_scrubjay.py tests/test_gpu_consumers.py tests/test_gpu_mfcc.py -m gpu -x -q > gpurun_out/r3b/tests29.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -1 gpurun_out/r3b/tests29.log | cut -c1-200
for w in stop config5 clips; do python bench.py --workload $w --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print(d['metric'], '%.4f ms frac %.3f' % (r['kernel_ms'], r['frac']), d['sensors']['during'])"; done
