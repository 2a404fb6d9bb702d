#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for v in head iirns; do
if [ $v == head ]; then L=$GRAFT_REPO_ROOT/dsp_amd/libdsp_amd.so; else L=$GRAFT_REPO_ROOT/variants/$v.so; fi
export DSP_AMD_LIB=$L
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b/iir_$v -- python3 $GRAFT_REPO_ROOT/bench.py --workload classify_f64 --no-cpu-baseline --steps 100 > /dev/null 2>&1; echo "prof $v rc=$?"
done
