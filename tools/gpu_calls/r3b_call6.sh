#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 600 python -m pytest tests/test_gpu_classify.py -m gpu -x -q > gpurun_out/r3b/tests6.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -1 gpurun_out/r3b/tests6.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for s in -1 0 1 2 3; do
DSP_AMD_CKPT_TAPS_SIMD=$s timeout -k 10 300 python tools/ab_classify.py --rounds 4 variants/p2.so variants/p3.so > gpurun_out/r3b/ab_cls6_s$s.txt 2>&1
echo "taps_simd=$s rc=$?"; tail -2 gpurun_out/r3b/ab_cls6_s$s.txt
done
