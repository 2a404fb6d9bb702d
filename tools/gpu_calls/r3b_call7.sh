#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 400 python tools/ab.py --rounds 10 --iters 20 variants/cur.so variants/snop16.so variants/snop32.so variants/vnop16.so variants/vnop32.so > gpurun_out/r3b/ab_nops.txt 2>&1
echo "rc=$?"; tail -8 gpurun_out/r3b/ab_nops.txt
