"""Builds libdsp_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.environ.get("DSP_AMD_LIB") or os.path.join(PKG, "libdsp_amd.so")
SOURCES = ["capi.cpp", "capi_consumers.cpp", "capi_classifier_cxx.cpp", "capi_classify_f64.cpp", "capi_gather.cpp", "classify_f64_kernels.hip", "classify_f64_ckpt_kernels.hip", "tables.cpp", "mfcc_kernels.hip", "mfcc1024_kernel.hip", "mfcc1024_wave_kernel.hip", "mfcc2048_kernel.hip", "classify_kernels.hip",
           "svm_kernels.hip", "consumer_kernels.hip"]
# Measured dead ends of the 512-point kernel (row per frame: 0.537 ms, two frames per wavefront step: 0.44-0.45 ms against 0.41 for
# the default kernel; A/B records in profiles/r02_wave_priority_ab.txt): kept buildable, outside the product library.
# DSP_AMD_EXPERIMENTS=1 python -m dsp_amd.build adds them (dsp_version() then carries "+experiments", DSP_KERNEL_ROW / _PAIR work).
EXPERIMENT_SOURCES = ["mfcc_row_kernel.hip", "mfcc512_pair_kernel.hip"]
EXPERIMENTS = os.environ.get("DSP_AMD_EXPERIMENTS", "") not in ("", "0")
HEADERS = ["exports.map", "tables.hpp", "clip_span.hpp", "mfcc_kernels.hpp", "mfcc_device.hpp", "classify_kernels.hpp", "svm_kernels.hpp", "classify_f64_device.hpp", "diag_guard.hpp", "consumer_kernels.hpp", "capi_util.hpp",
           os.path.join("..", "..", "include", "dsp_amd.h"), os.path.join("..", "..", "include", "dsp_amd_classifier.h")]


def source_hash() -> str:
    """sha256 over every source and header of the library, compiled into it (dsp_version()) so that a loaded
    binary can be checked against the tree it claims to come from."""
    import hashlib
    h = hashlib.sha256()
    h.update(b"+experiments" if EXPERIMENTS else b"")
    for name in sorted(SOURCES + (EXPERIMENT_SOURCES if EXPERIMENTS else []) + HEADERS):
        h.update(name.encode())
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libdsp_amd cannot be built (there is no CPU fallback)")


def is_stale() -> bool:
    """True when the library on disk was not built from the sources in the tree: by the hash the build left next to it
    (mtimes do not survive a copy of the tree to another machine), with the mtime comparison as a second opinion.  The
    authoritative check is made by lib.load() AFTER loading: the hash compiled into the binary (dsp_version())."""
    if not os.path.exists(LIB):
        return True
    try:
        with open(LIB + ".hash") as f:
            if f.read().strip() != source_hash():
                return True
    except OSError:
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + (EXPERIMENT_SOURCES if EXPERIMENTS else []) + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into dsp_amd/libdsp_amd.so (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB
    # several ranks of one job may get here together (bench.py under torch.distributed.run): one of them builds
    # into a temporary file that is renamed into place, the others wait on the lock and find the library fresh
    import fcntl
    lock = open(LIB + ".lock", "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not is_stale():
            return LIB
        return _build_locked(verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(verbose: bool) -> str:
    tmp = LIB + f".tmp{os.getpid()}"
    # -fno-slp-vectorize: packed fp32 VALU (v_pk_fma_f32 ...) issues at half rate on
    # gfx950, so SLP packing only adds register shuffles (measured: 292 -> 222
    # issue slots per frame, 126 -> 108 VGPRs).
    # -fvisibility=hidden: only what include/dsp_amd.h and include/dsp_amd_classifier.h declare is exported
    # (#pragma GCC visibility push(default) in those headers); tests/test_capi_cpu.py asserts the export list.
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
             "-fno-slp-vectorize", "-Wno-unused-value", f'-DDSP_AMD_SRC_HASH="{source_hash()}"']
    flags += os.environ.get("DSP_AMD_EXTRA_FLAGS", "").split()
    sources = list(SOURCES)
    if EXPERIMENTS:
        flags.append("-DDSP_AMD_EXPERIMENTS")
        sources += EXPERIMENT_SOURCES
    if verbose:
        flags.append("-Rpass-analysis=kernel-resource-usage")
    # one hipcc -c per source, side by side (the kernels dominate: ~50 s in sequence, ~20 s on 8 cores), then one link
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    objdir = tempfile.mkdtemp(prefix="dsp_amd_build_")
    try:
        h = source_hash()

        def compile_one(src):
            obj = os.path.join(objdir, src + ".o")
            r = subprocess.run([_hipcc(), "-c", *flags, "-o", obj, os.path.join(CSRC, src)], cwd=CSRC, capture_output=True, text=True)
            return src, obj, r

        jobs = int(os.environ.get("DSP_AMD_BUILD_JOBS", "0")) or min(len(sources), max(1, (os.cpu_count() or 2)))
        with ThreadPoolExecutor(jobs) as pool:
            results = list(pool.map(compile_one, sources))
        log = "".join(f"---- {src}\n{r.stdout}{r.stderr}" for src, _, r in results if r.stdout or r.stderr)
        if verbose and log:
            print(log)
        bad = [src for src, _, r in results if r.returncode != 0]
        if bad:
            raise RuntimeError(f"hipcc failed on {bad}:\n{log}")
        subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
                               "-o", tmp] + os.environ.get("DSP_AMD_EXTRA_LDFLAGS", "").split() + [obj for _, obj, _ in results], cwd=CSRC)
        os.replace(tmp, LIB)
        with open(LIB + ".hash", "w") as f:
            f.write(h + "\n")
    finally:
        shutil.rmtree(objdir, ignore_errors=True)
        if os.path.exists(tmp):
            os.remove(tmp)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
