#!/bin/bash
# round 3, call 4: float64 classifier on the GPU; idle-gap probe for the bench protocol
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_classify_f64.py -m gpu -x -q > gpurun_out/r3/tests4.log 2>&1
echo "tests rc=$?"; tail -25 gpurun_out/r3/tests4.log
python tools/idle_gap_probe.py 20 > gpurun_out/r3/idle_gap_20.txt 2>&1; cat gpurun_out/r3/idle_gap_20.txt
