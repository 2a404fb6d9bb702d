set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/round_r4f; mkdir -p $OUT
for w in classify_ragged config5 config5_ragged config5_2048 stop; do
    ( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 $R/bench.py --workload $w --no-cpu-baseline --steps 20 > $OUT/trace_$w.json 2> $OUT/trace_$w.err )
done
bash tools/run_round4.sh r4f c 2>&1 | tail -14
bash tools/run_round4.sh r4f d 2>&1 | tail -3
