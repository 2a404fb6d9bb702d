set -u
mkdir -p gpurun_out/r4b
timeout -k 10 900 python -m pytest tests/test_gpu_ragged.py tests/test_gpu_classify_f64.py tests/test_gpu_classify_f64_ckpt.py -x -q > gpurun_out/r4b/f64b.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4b/f64b.log | cut -c1-250
for w in classify_f64 classify_f64_pcm16 classify_f64; do python bench.py --workload $w --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['metric'], d['config']['workload'][:30], d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4b/trace_f64b -- python3 $GRAFT_REPO_ROOT/bench.py --workload classify_f64 --no-cpu-baseline --steps 20 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r4b/trace_f64b -name "*kernel_stats.csv" | while read f; do head -6 "$f" | cut -c1-60,230-330; done
