#!/bin/bash
mkdir -p gpurun_out/r3
for lib in variants/c3w3b.so dsp_amd/libdsp_amd.so; do
  tag=$(basename $lib .so)
  DSP_AMD_LIB=$GRAFT_REPO_ROOT/$lib bash tools/pmc_any.sh w3_${tag} mfcc1024 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" -- --workload config3 --no-config4 --settle 0 2>&1 | tail -1
done
python - <<'PY'
import ctypes as C, torch
for path in ("variants/c3w3b.so", "dsp_amd/libdsp_amd.so"):
    L = C.CDLL(path)
    print(path, "loaded")
PY
