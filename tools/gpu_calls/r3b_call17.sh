#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 600 python -m pytest tests/test_gpu_classify.py -m gpu -x -q > gpurun_out/r3b/tests17.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -1 gpurun_out/r3b/tests17.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/ab_classify.py variants/p6.so variants/p7.so variants/p8.so variants/p9.so variants/p6.so variants/p9.so > gpurun_out/r3b/ab_cls17.txt 2>&1
echo "ab rc=$?"; tail -6 gpurun_out/r3b/ab_cls17.txt
for n in 65536 98304; do timeout -k 10 300 python tools/ab_classify.py --clips $n --rounds 5 variants/p6.so variants/p9.so 2>&1 | tail -2 | sed "s/^/[$n] /"; done > gpurun_out/r3b/ab_cls17_sizes.txt 2>&1; cat gpurun_out/r3b/ab_cls17_sizes.txt
