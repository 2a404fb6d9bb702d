set -u
mkdir -p gpurun_out/r4b
timeout -k 10 900 python -m pytest tests/test_gpu_ragged.py -x -q > gpurun_out/r4b/ragged6.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4b/ragged6.log | cut -c1-250
for w in config5 config5_ragged classify classify_ragged; do python bench.py --workload $w --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['metric'], d['config']['workload'][:30], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
