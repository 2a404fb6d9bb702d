set -u
mkdir -p gpurun_out/r4b
timeout -k 10 900 python -m pytest tests/test_gpu_ragged.py tests/test_gpu_classify_f64.py tests/test_gpu_classify_f64_ckpt.py -x -q > gpurun_out/r4b/ragged4.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4b/ragged4.log | cut -c1-250
for w in classify_f64_pcm16 classify_f64 classify_f64_pcm16 classify_f64; do python bench.py --workload $w --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['metric'], d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
python tools/prof_classify_f64.py > /dev/null 2>&1
