"""The drop-in claims of INTEGRATION.md at the linker: plain C callers built with gcc against include/dsp_amd.h and
libdsp_amd.so.  CPU tier: they compile, link and fail loudly without a GPU; where /root/reference exists the
reference-side shims of examples/reference_shims compile against the reference's OWN headers (model_params.h,
gmm_params.inc are included from where they lie, nothing is copied).  GPU tier: the binaries run and agree with the
reference's compiled outputs (goldens)."""
import os
import shutil
import subprocess
import wave

import numpy as np
import pytest

import dsp_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DSP_REF", "/root/reference")
LIBDIR = os.path.join(ROOT, "dsp_amd")


def _gcc(out, srcs, incs, extra=()):
    dsp_amd.load()                                  # builds libdsp_amd.so when stale
    cmd = ["gcc", "-O2", "-std=gnu11", "-D__HIP_PLATFORM_AMD__"] + [f"-I{i}" for i in incs] + list(srcs) + \
          [f"-L{LIBDIR}", "-ldsp_amd", f"-Wl,-rpath,{LIBDIR}", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", out]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return out


def _write_wav(path, pcm):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(np.ascontiguousarray(pcm, np.int16).tobytes())


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_caller_links_against_the_library(tmp_path):
    exe = _gcc(str(tmp_path / "main_mfcc"), [os.path.join(ROOT, "examples", "main_mfcc.c")], [os.path.join(ROOT, "include")])
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([exe], capture_output=True, text=True)
        assert "0 frames" in r.stdout and "no HIP device" in r.stderr        # loud, no CPU fallback


@pytest.mark.skipif(shutil.which("gcc") is None or not os.path.isdir(REF), reason="needs gcc and the reference checkout")
def test_reference_side_shims_compile_against_the_reference_headers(tmp_path):
    main = tmp_path / "main.c"
    main.write_text('#include "stop_detector.h"\n#include <stdio.h>\nint main(void){static float x[16000]; printf("%f\\n", classify_signal(x, 16000)); return 0;}\n')
    _gcc(str(tmp_path / "stop"), [str(main), os.path.join(ROOT, "examples", "reference_shims", "stop_detector_amd.c")],
         [os.path.join(REF, "2fa/audio/word/c"), os.path.join(ROOT, "include")])
    main2 = tmp_path / "main2.c"
    main2.write_text('#include "speaker_gmm.h"\n#include <stdio.h>\nint main(void){static float m[98*13]; printf("%d\\n", classify_speaker(m, 98)); return 0;}\n')
    _gcc(str(tmp_path / "spk"), [str(main2), os.path.join(ROOT, "examples", "reference_shims", "speaker_gmm_amd.c")],
         [os.path.join(REF, "2fa/audio/pico-audio/src"), os.path.join(ROOT, "include"), "/opt/rocm/include"])


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_caller_runs_and_matches_the_reference_golden(tmp_path, golden):
    g = golden("mfcc_ref.npz")
    wav = str(tmp_path / "stop.wav")
    _write_wav(wav, g["stop_pcm"])
    exe = _gcc(str(tmp_path / "main_mfcc"), [os.path.join(ROOT, "examples", "main_mfcc.c")], [os.path.join(ROOT, "include")])
    r = subprocess.run([exe, wav], capture_output=True, text=True, check=True)
    ref = g["mfcc__stop_121417"]
    assert f"{g['stop_pcm'].size} samples -> {ref.shape[0]} frames" in r.stdout
    frame0 = np.array([float(v) for v in r.stdout.split("frame 0:")[1].split()[:13]], np.float32)
    assert np.abs(frame0 - ref[0]).max() <= 1e-4 * np.abs(ref[0]).max() + 3e-4 + 5e-5      # printed with 4 decimals


def _write_wav_ch(path, pcm):
    pcm = np.ascontiguousarray(pcm, np.int16)
    with wave.open(path, "wb") as w:
        w.setnchannels(1 if pcm.ndim == 1 else pcm.shape[1])
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(pcm.tobytes())


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_ragged_c_caller_links_against_the_library(tmp_path):
    exe = _gcc(str(tmp_path / "main_files"), [os.path.join(ROOT, "examples", "main_files.c")], [os.path.join(ROOT, "include")])
    import torch
    if not torch.cuda.is_available():
        wav = str(tmp_path / "a.wav")
        _write_wav_ch(wav, np.zeros(4000, np.int16))
        r = subprocess.run([exe, wav], capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stderr             # loud, no CPU fallback


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_ragged_c_caller_labels_the_donut_recordings_in_one_call(tmp_path, golden):
    """examples/main_files.c: the donut classifier's own recordings (stereo int16, 0.15 - 3 s) as WAV files on one command line -> one
    ragged call -> the reference's line per file, with the labels the compiled reference returned."""
    g = golden("donut16k_ref.npz")
    names = sorted({k.split("__")[0] for k in g.files})
    files = []
    for n in names:
        files.append(str(tmp_path / (n + ".wav")))
        _write_wav_ch(files[-1], g[n + "__pcm"])
    exe = _gcc(str(tmp_path / "main_files"), [os.path.join(ROOT, "examples", "main_files.c")], [os.path.join(ROOT, "include")])
    for flag in ([], ["-d"]):
        r = subprocess.run([exe] + flag + files, capture_output=True, text=True, check=True)
        lines = r.stdout.strip().splitlines()
        assert len(lines) == len(names)
        for n, line in zip(names, lines):
            assert line.startswith(str(tmp_path / (n + ".wav")))
            assert ("has a Scrub Jay" in line) == bool(int(g[n + "__label"])), (flag, line)
