#!/bin/bash
mkdir -p gpurun_out/r3b
./tools/micro/dpp_probe
timeout -k 10 400 python tools/ab.py --rounds 6 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/cur.so variants/c3mr.so variants/c3d1.so variants/c3mrd1.so variants/cur.so variants/c3mrd1.so > gpurun_out/r3b/ab_c3_lds.txt 2>&1
echo "rc=$?"; tail -9 gpurun_out/r3b/ab_c3_lds.txt
