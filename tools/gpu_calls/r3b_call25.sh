#!/bin/bash
for v in cur c5nofinish c5notail c5nosvm; do echo "== $v"; DSP_AMD_LIB=variants/$v.so timeout -k 10 200 python tools/time_config5.py 2>&1 | tail -2 | head -1; done
