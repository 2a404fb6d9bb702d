#!/bin/bash
mkdir -p gpurun_out/r3
python tools/ab.py --rounds 5 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 dsp_amd/libdsp_amd.so variants/c3d1.so variants/c3d2.so variants/c3d3.so variants/c3d4.so > gpurun_out/r3/ab_c3_parts.txt 2>&1
cat gpurun_out/r3/ab_c3_parts.txt
