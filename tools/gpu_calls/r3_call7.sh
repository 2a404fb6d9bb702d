#!/bin/bash
mkdir -p gpurun_out/r3
python tools/ab.py --rounds 8 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/c3par.so variants/c3la1.so dsp_amd/libdsp_amd.so variants/c3par.so variants/c3la1.so dsp_amd/libdsp_amd.so > gpurun_out/r3/ab_c3_lookahead.txt 2>&1
cat gpurun_out/r3/ab_c3_lookahead.txt
python -m pytest tests/test_gpu_mfcc.py -m gpu -x -q -k "config3 or 1024 or prefilter" > gpurun_out/r3/tests7.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r3/tests7.log
