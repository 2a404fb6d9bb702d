#!/bin/bash
mkdir -p gpurun_out/r3
python tools/ab.py --rounds 6 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/c3par.so variants/c3a.so variants/c3b.so variants/c3c.so variants/c3w2.so variants/c3par.so variants/c3a.so variants/c3b.so variants/c3c.so > gpurun_out/r3/ab_c3_diag.txt 2>&1
cat gpurun_out/r3/ab_c3_diag.txt
