#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 400 python tools/ab.py --rounds 6 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/cur.so variants/c3fix.so variants/c3fixall.so variants/cur.so variants/c3fixall.so > gpurun_out/r3b/ab_c3_fix.txt 2>&1
echo "rc=$?"; tail -8 gpurun_out/r3b/ab_c3_fix.txt
timeout -k 10 400 python tools/ab.py --rounds 4 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 1 variants/cur.so variants/c3fixall.so > gpurun_out/r3b/ab_c3_fix_p1.txt 2>&1
echo "rc=$?"; tail -3 gpurun_out/r3b/ab_c3_fix_p1.txt
