#!/bin/bash
# tools/mkvariant.sh NAME [extra hipcc flags...] -> variants/NAME.so (A/B builds for tools/ab.py)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/variants"
cd "$ROOT/dsp_amd/csrc"
LOG=$(mktemp)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Wno-unused-value \
    -Rpass-analysis=kernel-resource-usage "$@" -o "$ROOT/variants/$NAME.so" capi.cpp capi_consumers.cpp tables.cpp mfcc_kernels.hip mfcc_row_kernel.hip mfcc1024_kernel.hip mfcc1024_wave_kernel.hip classify_kernels.hip svm_kernels.hip consumer_kernels.hip > "$LOG" 2>&1 || { grep -E "error" "$LOG" | head; rm -f "$LOG"; exit 1; }
# resource usage of the default instantiation (reference shape, float input, tile epilogue)
grep -A12 "Function Name: _ZN3dsp19mfcc512_wave_kernelILi4ELi10ELi3ELi512ELi0ELi1ELb0ELb0EE" "$LOG" | grep -E "VGPRs:|Occupancy \[|ScratchSize" | head -3
rm -f "$LOG"
