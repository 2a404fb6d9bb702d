#!/bin/bash
for v in cur c5notile c5nopool; do echo "== $v"; DSP_AMD_LIB=variants/$v.so timeout -k 10 200 python tools/time_config5.py 2>&1 | tail -2 | head -1; done
