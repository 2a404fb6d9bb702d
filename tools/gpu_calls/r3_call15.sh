#!/bin/bash
mkdir -p gpurun_out/r3
python tools/ab.py --rounds 6 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/c3par.so variants/c3np.so dsp_amd/libdsp_amd.so variants/c3par.so variants/c3np.so dsp_amd/libdsp_amd.so > gpurun_out/r3/ab_c3_pipe2.txt 2>&1
tail -6 gpurun_out/r3/ab_c3_pipe2.txt
python -m pytest tests/test_gpu_mfcc.py tests/test_gpu_fuzz.py -m gpu -x -q -k "config3 or 1024 or prefilter or fuzz or random" > gpurun_out/r3/tests15.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r3/tests15.log
