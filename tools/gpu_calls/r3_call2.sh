#!/bin/bash
# round 3, call 2: aubio-semantics mode on the GPU + config5_2048 bench with it
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_scrubjay.py -m gpu -x -q > gpurun_out/r3/tests2.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r3/tests2.log
python bench.py --workload config5_2048 --steps 20 --warmup 8 > gpurun_out/r3/c5_2048_aubio.json 2> gpurun_out/r3/c5_2048_aubio.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/r3/c5_2048_aubio.json; tail -3 gpurun_out/r3/c5_2048_aubio.err
