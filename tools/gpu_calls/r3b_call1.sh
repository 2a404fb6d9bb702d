#!/bin/bash
# round 3 (second session), call 1: classify with the persistent / prefetching recompute kernel and half-segment checkpoints
mkdir -p gpurun_out/r3b
timeout -k 10 600 python -m pytest tests/test_gpu_classify.py tests/test_gpu_fuzz.py tests/test_boundary_cxx.py -m gpu -x -q > gpurun_out/r3b/tests1.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r3b/tests1.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ab_classify.py variants/base.so variants/p1.so variants/p1_nohalf.so variants/base.so variants/p1.so > gpurun_out/r3b/ab_cls1.txt 2>&1
echo "ab rc=$?"; tail -8 gpurun_out/r3b/ab_cls1.txt
