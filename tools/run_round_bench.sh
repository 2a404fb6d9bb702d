set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_v12b; mkdir -p $OUT
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 200 --warmup 20 > $GRAFT_REPO_ROOT/$OUT/trace.json 2> $GRAFT_REPO_ROOT/$OUT/trace.err )
python bench.py > gpurun_out/bench_v12.json 2> gpurun_out/bench_v12.err
for w in clips config3 classify pcm16; do python bench.py --workload $w --no-cpu-baseline --steps 50 > gpurun_out/bench_v12_$w.json 2> gpurun_out/bench_v12_$w.err; done
cat gpurun_out/bench_v12*.json | cut -c1-600
