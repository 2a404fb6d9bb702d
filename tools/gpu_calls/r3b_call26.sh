#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_scrubjay.py tests/test_gpu_consumers.py -m gpu -x -q > gpurun_out/r3b/tests26.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3b/tests26.log | cut -c1-300
for v in cur c5lds cur c5lds; do echo "== $v"; DSP_AMD_LIB=variants/$v.so timeout -k 10 200 python tools/time_config5.py 2>&1 | tail -2 | head -1; done
