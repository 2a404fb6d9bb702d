#!/bin/bash
# round 3, call 5: float64 classifier (fixed test), fused classify_signal, stop bench (fused vs two kernels)
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_classify_f64.py tests/test_gpu_consumers.py -m gpu -x -q > gpurun_out/r3/tests5.log 2>&1
echo "tests rc=$?"; tail -12 gpurun_out/r3/tests5.log
python bench.py --workload stop --steps 20 --warmup 8 > gpurun_out/r3/stop_fused.json 2> gpurun_out/r3/stop_fused.err; echo "rc=$?"; cut -c1-250 gpurun_out/r3/stop_fused.json
DSP_AMD_STOP_TWO_KERNELS=1 python bench.py --workload stop --steps 20 --warmup 8 > gpurun_out/r3/stop_two.json 2> gpurun_out/r3/stop_two.err; echo "rc=$?"; cut -c1-250 gpurun_out/r3/stop_two.json
python bench.py --workload clips --clips 125000 --steps 20 --warmup 8 > gpurun_out/r3/clips_125k.json 2>/dev/null; cut -c1-200 gpurun_out/r3/clips_125k.json
