#!/bin/bash
timeout -k 10 300 python tools/ab.py --rounds 12 --iters 20 variants/cur.so variants/pf1.so variants/cur.so variants/pf1.so 2>&1 | tail -4
for w in clips pcm16; do for v in cur pf1 cur pf1; do DSP_AMD_LIB=variants/$v.so python bench.py --workload $w --no-cpu-baseline --steps 50 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('$w $v %.4f ms' % r['kernel_ms'])"; done; done
