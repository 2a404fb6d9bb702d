#!/bin/bash
mkdir -p gpurun_out/r3b
./tools/micro/dpp_probe > gpurun_out/r3b/dpp_probe.txt 2>&1; cat gpurun_out/r3b/dpp_probe.txt | cut -c1-330
timeout -k 10 900 python -m pytest tests/test_gpu_mfcc.py tests/test_gpu_fuzz.py -m gpu -x -q -k "config3 or 1024 or prefilter or fuzz or random" > gpurun_out/r3b/tests22.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r3b/tests22.log | cut -c1-500
timeout -k 10 400 python tools/ab.py --rounds 6 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/c3norow.so variants/c3row.so variants/c3norow.so variants/c3row.so > gpurun_out/r3b/ab_c3_row.txt 2>&1
echo "rc=$?"; tail -5 gpurun_out/r3b/ab_c3_row.txt
timeout -k 10 400 python tools/ab.py --rounds 4 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 1 variants/c3norow.so variants/c3row.so > gpurun_out/r3b/ab_c3_row_p1.txt 2>&1
tail -3 gpurun_out/r3b/ab_c3_row_p1.txt
