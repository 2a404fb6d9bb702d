#!/bin/bash
bash tools/prof_lib.sh dsp_amd/libdsp_amd.so cls_head classify > /dev/null 2>&1
python - <<PY
import sqlite3
db = sqlite3.connect("gpurun_out/prof_cls_head/t_results.db")
for r in db.execute("select * from top_kernels limit 6"):
    print("head", r[0].split("(")[0][-44:], r[1], round(r[3], 1), "us")
PY
