#!/bin/bash
mkdir -p gpurun_out/r3b
for n in 12288 32768 65536 98304 131072; do timeout -k 10 300 python tools/ab_classify.py --clips $n --rounds 5 variants/p6.so variants/p10.so 2>&1 | tail -2 | sed "s/^/[$n] /"; done > gpurun_out/r3b/ab_cls19_sizes.txt 2>&1; cat gpurun_out/r3b/ab_cls19_sizes.txt
