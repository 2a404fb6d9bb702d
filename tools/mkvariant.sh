#!/bin/bash
# tools/mkvariant.sh NAME [extra hipcc flags...] -> variants/NAME.so (A/B builds for tools/ab.py), same sources and flags as
# dsp_amd/build.py plus the extra ones
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/variants"
cd "$ROOT"
DSP_AMD_LIB="$ROOT/variants/$NAME.so" DSP_AMD_EXTRA_FLAGS="$*" python -m dsp_amd.build 2>&1 | grep -E "error|_ZN3dsp19mfcc512_wave_kernelILi4ELi10ELi3ELi512ELi0ELi1ELb0ELb0EE" -A 12 | grep -E "error|VGPRs:|Occupancy \[|ScratchSize" | head -4
ls -la "$ROOT/variants/$NAME.so" | cut -c1-80
