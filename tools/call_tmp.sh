set -u
mkdir -p gpurun_out/r4b
timeout -k 10 900 python -m pytest tests/test_gpu_scrubjay.py tests/test_gpu_fused_pcm16.py tests/test_gpu_ragged.py tests/test_gpu_mfcc.py -x -q > gpurun_out/r4b/t2048b.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4b/t2048b.log | cut -c1-250
for w in config5_2048 config5_2048; do python bench.py --workload $w --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['metric'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
python tools/sq_fractions.py r4g config5_2048 2>&1 | tail -1
