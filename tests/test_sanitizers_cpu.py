"""CPU: the sanitizer tier (reference precedent 2fa/audio/CMakeLists.txt:9 -fsanitize=...).  tools/asan_host.sh builds the
oracle's C restatement with gcc's AddressSanitizer + UBSan and the product's HOST code (tables.cpp, capi*.cpp; device code untouched)
with hipcc's -Xarch_host -fsanitize=address,undefined, and drives both through their own CPU tests with -fno-sanitize-recover: the
first finding aborts the leg.  Never run on the GPU box (GPU AddressSanitizer is not available on the pool)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("leg,marker", [("oracle", "ASAN-ORACLE-OK"), ("product", "ASAN-PRODUCT-OK")])
def test_sanitizer_leg(leg, marker):
    if os.environ.get("DSP_AMD_LIB") or os.environ.get("DSP_ORACLE_LIB"):
        pytest.skip("already inside a sanitizer leg")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host.sh"), leg], capture_output=True, text=True, timeout=1500, cwd=ROOT)
    out = r.stdout + r.stderr
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out, out[-6000:]
    assert r.returncode == 0 and marker in r.stdout, out[-6000:]
