#!/bin/bash
mkdir -p gpurun_out/r3b
DSP_AMD_DEBUG=1 DSP_AMD_LIB=variants/stamps.so timeout -k 10 300 python tools/rc_stamps.py > gpurun_out/r3b/stamps_p1.txt 2>&1
echo "rc=$?"; tail -18 gpurun_out/r3b/stamps_p1.txt
for b in 1 2; do
DSP_AMD_RC_BLOCKS_PER_CU=$b timeout -k 10 300 python tools/ab_classify.py --rounds 4 variants/base.so variants/p1.so > gpurun_out/r3b/ab_cls2_b$b.txt 2>&1
echo "b=$b rc=$?"; tail -2 gpurun_out/r3b/ab_cls2_b$b.txt
done
