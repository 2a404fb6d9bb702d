#!/bin/bash
# kernel-trace of the classify bench on the GPU box; prints the per-kernel averages (us)
# usage (inside gpurun): bash tools/prof_classify.sh [clips]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_cls
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cls -o cls -- python $R/bench.py --workload classify --clips ${1:-49152} > $R/gpurun_out/cls_prof_bench.log 2>&1
python - <<PY
import sqlite3
db = sqlite3.connect("$R/gpurun_out/prof_cls/cls_results.db")
for r in db.execute("select * from top_kernels limit 5"):
    print(r[0].split("(")[0][-40:], r[1], round(r[3], 1))
PY
grep -v "^W2026\|^E2026" $R/gpurun_out/cls_prof_bench.log | tail -1 | cut -c1-140
