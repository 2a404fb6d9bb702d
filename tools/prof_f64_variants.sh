#!/bin/bash
# On the GPU box: kernel times of the float64 classifier for several library builds (variants/*.so) and batch sizes.
# usage: bash tools/prof_f64_variants.sh "name1 name2 ..." "clips1 clips2 ..."   (name "default" = dsp_amd/libdsp_amd.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in $1; do
  for n in $2; do
    if [ "$v" = default ]; then unset DSP_AMD_LIB; else export DSP_AMD_LIB=$R/variants/$v.so; fi
    rm -rf $R/gpurun_out/pv_$v_$n
    rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pv_${v}_$n -o t -- python3 $R/tools/prof_classify_f64.py $n $3 > $R/gpurun_out/pv_${v}_$n.log 2>&1
    python3 - <<PY
import sqlite3
db = sqlite3.connect("$R/gpurun_out/pv_${v}_$n/t_results.db")
for r in db.execute("select * from top_kernels limit 12"):
    if "dsp::" in r[0]: print("$v", $n, r[0].split("(")[0][-48:], r[1], round(r[3], 1), "us")
PY
    rm -rf $R/gpurun_out/pv_${v}_$n
  done
done
