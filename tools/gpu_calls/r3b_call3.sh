#!/bin/bash
mkdir -p gpurun_out/r3b
for v in base p1_nohalf p1; do bash tools/prof_lib.sh variants/$v.so cls_$v classify 2>&1 | tail -7; done > gpurun_out/r3b/prof_cls3.txt 2>&1
cat gpurun_out/r3b/prof_cls3.txt
