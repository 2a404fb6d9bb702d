#!/bin/bash
mkdir -p gpurun_out/r3b
DSP_AMD_LIB=variants/c3f32s1.so timeout -k 10 900 python -m pytest tests/test_gpu_mfcc.py -m gpu -q -k "config3 or prefilter" > gpurun_out/r3b/tests37.log 2>&1; rc=$?; echo "tests (float32 section 1) rc=$rc"; tail -4 gpurun_out/r3b/tests37.log | cut -c1-600
timeout -k 10 400 python tools/ab.py --rounds 6 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/c3cur.so variants/c3f32s1.so variants/c3cur.so variants/c3f32s1.so > gpurun_out/r3b/ab_c3_f32s1.txt 2>&1
echo "rc=$?"; tail -5 gpurun_out/r3b/ab_c3_f32s1.txt
