#!/bin/bash
for v in cur pf1 cur pf1; do echo "== $v"; DSP_AMD_LIB=variants/$v.so timeout -k 10 200 python tools/time_config5.py 2>&1 | tail -2 | head -1; DSP_AMD_LIB=variants/$v.so timeout -k 10 200 python tools/time_stop.py 2>&1 | tail -1; done
timeout -k 10 300 python tools/ab.py --rounds 6 --iters 20 variants/cur.so variants/pf1.so 2>&1 | tail -2
