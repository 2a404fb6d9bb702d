#!/bin/bash
mkdir -p gpurun_out/r3
python bench.py --workload config3 --steps 20 --warmup 8 > gpurun_out/r3/c3_fused.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r3/c3_fused.json')); print('fused', d['roofline']['kernel_ms'])"
DSP_AMD_PREFILTER_TWO_PASS=1 python bench.py --workload config3 --steps 20 --warmup 8 > gpurun_out/r3/c3_two.json 2>/dev/null; python -c "import json; d=json.load(open('gpurun_out/r3/c3_two.json')); print('two-pass', d['roofline']['kernel_ms'])"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3/prof_c3 -o c3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload config3 --steps 10 --warmup 4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r3/prof_c3 -name "*kernel_stats*" | head -2; f=$(find gpurun_out/r3/prof_c3 -name "*kernel_stats.csv" | head -1); head -5 "$f" | cut -c1-200
