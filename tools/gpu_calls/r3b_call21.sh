#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 300 python tools/ab_classify.py --rounds 6 variants/q0.so variants/q_prio.so variants/q_b32.so variants/q_b8.so variants/q0.so variants/q0.so > gpurun_out/r3b/ab_cls21.txt 2>&1
echo "ab rc=$?"; tail -7 gpurun_out/r3b/ab_cls21.txt
timeout -k 10 400 python tools/ab.py --rounds 6 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/q0.so variants/c3p0.so variants/c3p1.so variants/c3p2.so variants/q0.so > gpurun_out/r3b/ab_c3_prio.txt 2>&1
echo "rc=$?"; tail -6 gpurun_out/r3b/ab_c3_prio.txt
