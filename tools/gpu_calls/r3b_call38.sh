#!/bin/bash
mkdir -p gpurun_out/r3b
for v in c3cur c3f32s1; do DSP_AMD_LIB=variants/$v.so timeout -k 10 900 python -m pytest tests/test_gpu_mfcc.py -m gpu -q -k "stop_band" > gpurun_out/r3b/tests38_$v.log 2>&1; echo "$v rc=$?"; cp gpurun_out/gate_report.json gpurun_out/r3b/gate_$v.json; python - <<PY
import json
d = json.load(open("gpurun_out/gate_report.json"))
rows = [(k, x["pure_worst_rel"]) for k, x in d.items() if "stop-band" in k]
rows.sort(key=lambda r: -r[1])
for k, x in rows[:6]: print("   %-72s %.3e" % (k, x))
PY
done
