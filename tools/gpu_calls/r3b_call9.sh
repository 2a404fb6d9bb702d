#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 400 python tools/ab.py --rounds 5 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/cur.so variants/c3w3.so:2:0 variants/c3w3.so:3:0 variants/c3w3.so:4:0 variants/cur.so:2:0 variants/cur.so:3:0 > gpurun_out/r3b/ab_c3w3b.txt 2>&1
echo "rc=$?"; tail -8 gpurun_out/r3b/ab_c3w3b.txt
