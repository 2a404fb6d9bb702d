#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 600 python -m pytest tests/test_gpu_classify.py tests/test_gpu_fuzz.py -m gpu -x -q -k "classif or donut or clips" > gpurun_out/r3b/tests4.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r3b/tests4.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ab_classify.py variants/base.so variants/p1.so variants/p2.so variants/p2_nohalf.so variants/base.so variants/p2.so > gpurun_out/r3b/ab_cls4.txt 2>&1
echo "ab rc=$?"; tail -8 gpurun_out/r3b/ab_cls4.txt
