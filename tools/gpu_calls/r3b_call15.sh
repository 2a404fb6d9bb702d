#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 600 python -m pytest tests/test_gpu_classify.py tests/test_gpu_fuzz.py -m gpu -x -q -k "classif or donut or clips" > gpurun_out/r3b/tests15.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -1 gpurun_out/r3b/tests15.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ab_classify.py variants/p6.so variants/p7.so variants/p6.so variants/p7.so > gpurun_out/r3b/ab_cls15.txt 2>&1
echo "ab rc=$?"; tail -5 gpurun_out/r3b/ab_cls15.txt
for n in 12288 24576 65536 131072; do timeout -k 10 300 python tools/ab_classify.py --clips $n --rounds 4 variants/p6.so variants/p7.so 2>&1 | tail -2 | sed "s/^/[$n] /"; done > gpurun_out/r3b/ab_cls15_sizes.txt 2>&1; cat gpurun_out/r3b/ab_cls15_sizes.txt
