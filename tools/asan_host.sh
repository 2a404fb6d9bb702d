#!/bin/bash
# Sanitizer tier of the CPU-testable code (reference precedent 2fa/audio/CMakeLists.txt:9).  Never on the GPU box.
#   leg 1  oracle/*.c under gcc's AddressSanitizer + UBSan (make -C oracle asan), driven by the oracle's own tests
#   leg 2  the product's HOST code -- tables.cpp (bipartite matching, chunk packing, Durand-Kerner), the argument handling of capi*.cpp,
#          the planner -- as a build of libdsp_amd.so with -Xarch_host -fsanitize=address,undefined (device code untouched), driven
#          by the CPU tests of the C ABI
# -fno-sanitize-recover=all: the first finding aborts the run.  Usage: tools/asan_host.sh [oracle|product|all]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
WHAT=${1:-all}
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
if [ "$WHAT" = oracle ] || [ "$WHAT" = all ]; then
    make -C oracle asan > /dev/null
    LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" DSP_ORACLE_LIB="$ROOT/oracle/liboracle_asan.so" \
        python -m pytest -x -q -p no:cacheprovider tests/test_oracle_mfcc.py tests/test_oracle_classifier.py tests/test_oracle_classify_f64.py \
        tests/test_oracle_aubio.py tests/test_oracle_svm.py tests/test_oracle_consumers.py
    echo "ASAN-ORACLE-OK"
fi
if [ "$WHAT" = product ] || [ "$WHAT" = all ]; then
    mkdir -p variants
    DSP_AMD_LIB="$ROOT/variants/asan_host.so" DSP_AMD_EXTRA_FLAGS="-g -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-sanitize-recover=all" \
        DSP_AMD_EXTRA_LDFLAGS="-fsanitize=address,undefined -shared-libsan" python -m dsp_amd.build > /dev/null
    RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
    LD_PRELOAD="$RT" DSP_AMD_LIB="$ROOT/variants/asan_host.so" \
        python -m pytest -x -q -p no:cacheprovider tests/test_capi_cpu.py tests/test_planner_cpu.py tests/test_tables_grid_cpu.py -k "not oracle_tables and not product_never"
    echo "ASAN-PRODUCT-OK"
fi
