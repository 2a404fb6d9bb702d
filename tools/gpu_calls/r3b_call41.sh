#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_classify.py tests/test_gpu_fuzz.py tests/test_boundary_cxx.py -m gpu -x -q > gpurun_out/r3b/tests41.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -1 gpurun_out/r3b/tests41.log | cut -c1-300
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ab_classify.py --rounds 10 variants/head.so variants/bdpar.so variants/head.so variants/bdpar.so > gpurun_out/r3b/ab_cls41.txt 2>&1
echo "ab rc=$?"; tail -5 gpurun_out/r3b/ab_cls41.txt
