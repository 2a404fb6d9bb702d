#!/bin/bash
mkdir -p gpurun_out/r3b
DSP_AMD_LIB=variants/stamps.so timeout -k 10 300 python tools/rc_stamps.py > gpurun_out/r3b/stamps_p2.txt 2>&1
echo "rc=$?"; tail -14 gpurun_out/r3b/stamps_p2.txt
