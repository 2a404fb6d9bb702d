#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_scrubjay.py tests/test_gpu_consumers.py -m gpu -x -q > gpurun_out/r3b/tests33.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -1 gpurun_out/r3b/tests33.log | cut -c1-200
for v in cur c5pf1 cur c5pf1; do echo "== $v"; DSP_AMD_LIB=variants/$v.so timeout -k 10 200 python tools/time_config5.py 2>&1 | tail -2 | head -1; done
