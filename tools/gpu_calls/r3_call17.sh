#!/bin/bash
mkdir -p gpurun_out/r3
python tools/ab_classify.py --rounds 6 variants/ck3.so dsp_amd/libdsp_amd.so variants/ck3.so dsp_amd/libdsp_amd.so > gpurun_out/r3/ab_ckpt_dual.txt 2>&1
cat gpurun_out/r3/ab_ckpt_dual.txt | tail -8
python -m pytest tests/test_gpu_classify.py tests/test_gpu_fuzz.py tests/test_boundary_cxx.py tests/test_integration_c.py -m gpu -x -q > gpurun_out/r3/tests17.log 2>&1; echo rc=$?; tail -3 gpurun_out/r3/tests17.log
