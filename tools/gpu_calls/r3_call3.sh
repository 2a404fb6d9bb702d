#!/bin/bash
# round 3, call 3: the new bench.py line (sensors, per-launch stats, reference on all cores), event-per-step perturbation, N = 2 rehearsals
mkdir -p gpurun_out/r3
python bench.py --steps 20 --warmup 5 > gpurun_out/r3/bench_default.json 2> gpurun_out/r3/bench_default.err
echo "default rc=$?"; cut -c1-300 gpurun_out/r3/bench_default.json
for i in 1 2 3; do
  BENCH_STEP_EVENTS=0 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-config4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two events   ', d['roofline']['kernel_ms'], d['ms_per_step'])"
  BENCH_STEP_EVENTS=1 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-config4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('event / step ', d['roofline']['kernel_ms'], d['ms_per_step'], d['roofline']['kernel_ms_min'], d['roofline']['kernel_ms_median'])"
done > gpurun_out/r3/step_events_ab.txt 2>&1
cat gpurun_out/r3/step_events_ab.txt
BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3/rehearsal2.json 2> gpurun_out/r3/rehearsal2.err
echo "rehearsal rc=$?"; python -c "import json; d=json.load(open('gpurun_out/r3/rehearsal2.json')); print(json.dumps(d.get('config4'))[:900])"; tail -3 gpurun_out/r3/rehearsal2.err
BENCH_REHEARSAL=1 timeout -k 10 300 python bench.py --gpus 2 --workload config5 --clips 20000 --steps 5 --warmup 4 > gpurun_out/r3/rehearsal2_c5.json 2> gpurun_out/r3/rehearsal2_c5.err
echo "rehearsal c5 rc=$?"; cut -c1-700 gpurun_out/r3/rehearsal2_c5.json; tail -3 gpurun_out/r3/rehearsal2_c5.err
