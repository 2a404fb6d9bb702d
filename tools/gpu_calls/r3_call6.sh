#!/bin/bash
# round 3, call 6: consumers + f64 classify tests, stop bench, config3 cascade vs parallel form A/B + tests
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_classify_f64.py tests/test_gpu_consumers.py -m gpu -x -q > gpurun_out/r3/tests6.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r3/tests6.log
python -m pytest tests/test_gpu_mfcc.py -m gpu -x -q -k "config3 or 1024 or prefilter" > gpurun_out/r3/tests6b.log 2>&1
echo "config3 tests rc=$?"; tail -6 gpurun_out/r3/tests6b.log
python bench.py --workload stop --steps 20 --warmup 8 > gpurun_out/r3/stop_fused.json 2> gpurun_out/r3/stop_fused.err; python -c "import json; d=json.load(open('gpurun_out/r3/stop_fused.json')); print('stop fused', d['roofline']['kernel_ms'], d['sensors']['during'])"
DSP_AMD_STOP_TWO_KERNELS=1 python bench.py --workload stop --steps 20 --warmup 8 > gpurun_out/r3/stop_two.json 2> gpurun_out/r3/stop_two.err; python -c "import json; d=json.load(open('gpurun_out/r3/stop_two.json')); print('stop two  ', d['roofline']['kernel_ms'], d['sensors']['during'])"
python tools/ab.py --rounds 8 --iters 10 --frame-length 1024 --n-fft 1024 --n-mels 128 --prefilter 2 variants/c3par.so dsp_amd/libdsp_amd.so variants/c3par.so dsp_amd/libdsp_amd.so > gpurun_out/r3/ab_c3_cascade.txt 2>&1
cat gpurun_out/r3/ab_c3_cascade.txt
