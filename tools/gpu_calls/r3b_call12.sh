#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 900 python -m pytest tests/test_gpu_mfcc.py tests/test_gpu_fuzz.py -m gpu -x -q -k "config3 or 1024 or prefilter or fuzz or random" > gpurun_out/r3b/tests12.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r3b/tests12.log | cut -c1-600
timeout -k 10 400 python tools/ab.py --rounds 5 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/c3fixall.so variants/c3v2.so variants/c3fixall.so variants/c3v2.so > gpurun_out/r3b/ab_c3_v2.txt 2>&1
echo "rc=$?"; tail -5 gpurun_out/r3b/ab_c3_v2.txt
