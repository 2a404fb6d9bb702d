#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 400 python tools/ab.py --rounds 6 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/cur.so variants/c3w3.so variants/cur.so variants/c3w3.so > gpurun_out/r3b/ab_c3w3.txt 2>&1
echo "rc=$?"; tail -6 gpurun_out/r3b/ab_c3w3.txt
