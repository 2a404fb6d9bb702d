"""The C++ side of the drop-in boundary: libdsp_amd.so exports the reference's own C++-linkage names
(sync/lib/classifier.h:14-19), so a caller compiled against the REFERENCE'S header links the library instead of
classifier.cpp + PlainFFT.cpp (what sync/sync.cpp:202 needs).

CPU tier: the export list is exactly include/dsp_amd.h + include/dsp_amd_classifier.h; examples/main_classify.cpp
compiles with g++ against /root/reference/sync/lib/classifier.h (where present) and against
include/dsp_amd_classifier.h, links with -ldsp_amd only, and fails loudly without a GPU.
GPU tier: the binary's six entry points reproduce the goldens of the reference's compiled classifier.cpp bit for bit."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import dsp_amd
from dsp_amd import lib as dl

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DSP_REF", "/root/reference")
LIBDIR = os.path.join(ROOT, "dsp_amd")
CASES = ["noise", "burst_2k", "jay_like", "scrub_a", "scrub_b", "silence", "birdq_ch0_1s"]

needs_gxx = pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")


def _gxx(out, against_reference):
    dsp_amd.load()                                  # builds libdsp_amd.so when stale
    inc = ["-DUSE_REFERENCE_HEADER", f"-I{os.path.join(REF, 'sync/lib')}"] if against_reference else [f"-I{os.path.join(ROOT, 'include')}"]
    cmd = ["g++", "-O2", "-std=c++17"] + inc + [os.path.join(ROOT, "examples", "main_classify.cpp"),
           f"-L{LIBDIR}", "-ldsp_amd", f"-Wl,-rpath,{LIBDIR}", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", out]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return out


def test_export_list_is_exactly_the_two_headers():
    dsp_amd.load()
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIBDIR, "libdsp_amd.so")], check=True, capture_output=True, text=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == sorted(dl.SYMBOLS + list(dl.CXX_SYMBOLS.values()))


@needs_gxx
@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "sync/lib")), reason="needs the reference checkout")
def test_caller_compiled_against_the_reference_header_links_the_library(tmp_path):
    """No classifier.cpp, no PlainFFT.cpp on the link line: every symbol classifier.h declares resolves in libdsp_amd.so."""
    exe = _gxx(str(tmp_path / "main_classify_ref"), against_reference=True)
    undefined = subprocess.run(["nm", "-u", "-C", exe], check=True, capture_output=True, text=True).stdout
    for name in dl.CXX_SYMBOLS:
        assert f" {name}(" in undefined, name                         # bound at load time ...
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libdsp_amd.so" in ldd                                      # ... from this library


@needs_gxx
def test_caller_fails_loudly_without_a_gpu(tmp_path):
    import torch
    exe = _gxx(str(tmp_path / "main_classify"), against_reference=False)
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    clip = tmp_path / "clip.f32"
    np.zeros(16000, np.float32).tofile(clip)
    r = subprocess.run([exe, str(clip), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert r.returncode == 0 and "no HIP device" in r.stderr           # reference return conventions, reason on stderr
    blob = open(tmp_path / "out.bin", "rb").read()
    assert struct.unpack_from("<i", blob, 0)[0] == 1                   # butter_bandpass is host-only and still answers
    assert struct.unpack_from("<i", blob, len(blob) - 4)[0] == 0       # classify -> 0 on failure (classifier.cpp:87-91)


def _parse(blob, with_sum):
    off = 0

    def i32():
        nonlocal off
        v = struct.unpack_from("<i", blob, off)[0]; off += 4
        return v

    def f32(k):
        nonlocal off
        v = np.frombuffer(blob, np.float32, k, off).copy(); off += 4 * k
        return v

    r = {"ok": i32(), "b": f32(9), "a": f32(9), "bad_ok": i32()}
    n = i32(); r["filtered"] = f32(n)
    nf, nt = i32(), i32()
    r["freqs"], r["times"] = f32(nf), f32(nt)
    r["sxx"] = f32(nf * nt).reshape(nf, nt)
    nm = i32(); r["midpoints"] = f32(nm)
    r["label"] = i32()
    if with_sum:
        r["sum"] = f32(1)[0]
    assert off == len(blob)
    return r


@pytest.mark.gpu
@needs_gxx
def test_cxx_entry_points_match_the_compiled_reference(tmp_path, golden):
    g = golden("classifier_ref.npz")
    s = golden("sum_intense_ref.npz")
    exe = _gxx(str(tmp_path / "main_classify"), against_reference=os.path.isdir(os.path.join(REF, "sync/lib")))
    n_sum = int(s["n_cases"])
    for k, name in enumerate(CASES):
        clip = tmp_path / f"{name}.f32"
        g[f"{name}__input"].tofile(clip)
        out = tmp_path / f"{name}.bin"
        c = k * 5 % n_sum                                               # one sum_intense case rides along with each clip
        for key in ("db", "freqs", "times"):
            s[f"c{c}_{key}"].tofile(tmp_path / f"{key}.f32")
        nf, nt = s[f"c{c}_db"].shape
        lo, hi, half, mid = (repr(float(v)) for v in s[f"c{c}_params"])
        subprocess.run([exe, str(clip), str(out), str(tmp_path / "db.f32"), str(nf), str(nt), lo, hi, half, mid,
                        str(tmp_path / "freqs.f32"), str(tmp_path / "times.f32")], check=True, capture_output=True, text=True)
        r = _parse(open(out, "rb").read(), True)
        assert r["ok"] == 1 and r["bad_ok"] == int(g["bad_band_ok"]) == 0
        assert np.array_equal(r["b"], g["b_3000_7500"]) and np.array_equal(r["a"], g["a_3000_7500"])
        assert np.array_equal(r["filtered"], g[f"{name}__filtered"])
        assert np.array_equal(r["freqs"], g["freqs"]) and np.array_equal(r["times"], g["times_16000"])
        assert np.array_equal(r["sxx"], g[f"{name}__sxx"])
        assert np.array_equal(r["midpoints"], g[f"{name}__midpoints"])
        assert r["label"] == int(g[f"{name}__label"])
        assert r["sum"].tobytes() == s[f"c{c}_sum"].tobytes()


@pytest.mark.gpu
def test_sum_intense_every_reference_case(golden):
    import ctypes as C
    s = golden("sum_intense_ref.npz")
    L = dsp_amd.load()
    for c in range(int(s["n_cases"])):
        db, fr, tm = (np.ascontiguousarray(s[f"c{c}_{k}"]) for k in ("db", "freqs", "times"))
        lo, hi, half, mid = (float(v) for v in s[f"c{c}_params"])
        out = C.c_float()
        dl.check(L.dsp_sum_intense_f32(lo, hi, half, fr.ctypes.data, fr.size, tm.ctypes.data, tm.size, db.ctypes.data, mid, C.byref(out)), "sum_intense")
        assert np.float32(out.value).tobytes() == s[f"c{c}_sum"].tobytes(), c
