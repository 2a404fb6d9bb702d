#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 600 python -m pytest tests/test_gpu_classify.py tests/test_gpu_fuzz.py tests/test_boundary_cxx.py tests/test_integration_c.py -m gpu -x -q > gpurun_out/r3b/tests14.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -1 gpurun_out/r3b/tests14.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ab_classify.py variants/base.so variants/p5.so variants/p6.so variants/p5.so variants/p6.so > gpurun_out/r3b/ab_cls14.txt 2>&1
echo "ab rc=$?"; tail -6 gpurun_out/r3b/ab_cls14.txt
timeout -k 10 300 python tools/ab_classify.py --clips 65536 --rounds 5 variants/base.so variants/p5.so variants/p6.so > gpurun_out/r3b/ab_cls14_64k.txt 2>&1
echo "ab rc=$?"; tail -3 gpurun_out/r3b/ab_cls14_64k.txt
python tools/traffic.py r3b classify 2>&1 | tail -1
