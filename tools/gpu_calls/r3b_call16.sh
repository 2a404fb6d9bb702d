#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 600 python -m pytest tests/test_gpu_classify.py tests/test_gpu_fuzz.py tests/test_boundary_cxx.py tests/test_integration_c.py -m gpu -x -q > gpurun_out/r3b/tests16.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -1 gpurun_out/r3b/tests16.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for n in 12288 49152 98304; do timeout -k 10 300 python tools/ab_classify.py --clips $n --rounds 5 variants/p6.so variants/p8.so 2>&1 | tail -2 | sed "s/^/[$n] /"; done > gpurun_out/r3b/ab_cls16_sizes.txt 2>&1; cat gpurun_out/r3b/ab_cls16_sizes.txt
bash tools/prof_lib.sh variants/p8.so cls_p8 classify > /dev/null 2>&1
python - <<PY
import sqlite3
db = sqlite3.connect("gpurun_out/prof_cls_p8/t_results.db")
for r in db.execute("select * from top_kernels limit 5"):
    print("p8", r[0].split("(")[0][-44:], r[1], round(r[3], 1), "us")
PY
