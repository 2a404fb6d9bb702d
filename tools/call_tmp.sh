set -u
mkdir -p gpurun_out/r4b
timeout -k 10 800 python -m pytest tests/test_gpu_ragged.py tests/test_gpu_classify.py tests/test_gpu_classify_pcm16.py -x -q > gpurun_out/r4b/ragged2.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/r4b/ragged2.log | cut -c1-250
