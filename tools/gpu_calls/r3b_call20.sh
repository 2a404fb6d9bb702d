#!/bin/bash
mkdir -p gpurun_out/r3b
for v in cur c5nosvm c5nopool; do echo "== $v"; DSP_AMD_LIB=variants/$v.so timeout -k 10 200 python tools/time_config5.py 2>&1 | tail -2; done > gpurun_out/r3b/c5_parts.txt 2>&1; cat gpurun_out/r3b/c5_parts.txt
