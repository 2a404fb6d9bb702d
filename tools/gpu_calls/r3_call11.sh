#!/bin/bash
mkdir -p gpurun_out/r3
python tools/ab.py --rounds 6 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/c3par.so variants/c3b.so dsp_amd/libdsp_amd.so variants/c3w3.so variants/c3par.so variants/c3b.so dsp_amd/libdsp_amd.so variants/c3w3.so > gpurun_out/r3/ab_c3_lds.txt 2>&1
cat gpurun_out/r3/ab_c3_lds.txt
python tools/ab.py --rounds 6 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 variants/c3par.so dsp_amd/libdsp_amd.so variants/c3par.so dsp_amd/libdsp_amd.so > gpurun_out/r3/ab_1024_plain.txt 2>&1
cat gpurun_out/r3/ab_1024_plain.txt
python -m pytest tests/test_gpu_mfcc.py tests/test_gpu_fuzz.py -m gpu -x -q -k "config3 or 1024 or prefilter or fuzz or random" > gpurun_out/r3/tests11.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r3/tests11.log
