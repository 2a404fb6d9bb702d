#!/bin/bash
mkdir -p gpurun_out/r3b
timeout -k 10 300 python tools/ab_classify.py variants/p6.so variants/p7.so variants/p9.so variants/p10.so variants/p7.so variants/p10.so > gpurun_out/r3b/ab_cls18.txt 2>&1
echo "ab rc=$?"; tail -6 gpurun_out/r3b/ab_cls18.txt
