"""CPU: the oracle's MFCC chain against goldens produced by the reference's own
compiled mfcc.c (tests/golden/make_golden.py) -- pins oracle/dsp_oracle.c."""
import hashlib

import numpy as np
import pytest

from oracle import oracle as O
from tests import signals as S
from tests.conftest import frame_linf_close


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _cases(g):
    c = S.mfcc_cases()
    c["chirp"] = g["input__chirp"]
    bird = g["birdq_pcm"]
    c["birdq_ch0"] = (bird[:, 0] / np.float32(32768.0)).astype(np.float32)
    c["birdq_avg"] = (np.float32(0.5) * (bird[:, 0] / np.float32(32768.0) + bird[:, 1] / np.float32(32768.0))).astype(np.float32)
    c["stop_121417"] = (g["stop_pcm"] / np.float32(32768.0)).astype(np.float32)
    return c


def test_tables_match_reference_digests(golden):
    t = golden("tables_ref.npz")
    assert _sha(O.window(O.WINDOW_HANN, 400)) == str(t["hann_sha"])
    mel = O.mel_filterbank()
    assert _sha(mel) == str(t["mel_sha"])
    assert int((mel != 0).sum()) == int(t["mel_nonzero"]) == 494
    dct = O.dct_ortho(13, 40).reshape(-1)
    idx, val = t["dct_mismatch_idx"], t["dct_mismatch_val"]
    # numpy-SIMD-cos vs glibc cosf: <= 1 ulp on the listed entries, exact elsewhere
    assert np.abs(dct[idx] - val).max() <= 1.5e-8
    dct[idx] = val
    assert _sha(dct) == str(t["dct_sha"])


def test_fft_reference_order_is_bit_exact(golden):
    g = golden("mfcc_ref.npz")
    frame = S.mfcc_cases()["noise0"][:400]
    assert np.array_equal(O.fft_real_forward(frame, 512, O.FFT_REFERENCE_ORDER), g["fft__noise0_frame0"])


@pytest.mark.parametrize("name", ["noise0", "noise1", "noise2", "chirp", "silence", "tiny", "dc", "impulse",
                                  "half_silent", "len399", "len400", "len559", "len560", "long",
                                  "birdq_ch0", "birdq_avg", "stop_121417"])
def test_compute_mfcc_matches_reference(golden, name):
    g = golden("mfcc_ref.npz")
    ref = g["mfcc__" + name]
    got = O.compute_mfcc(_cases(g)[name], 500)
    assert got.shape == ref.shape
    # bit-faithful up to the 1-ulp DCT-table entries: far inside the 1e-4 gate
    ok, worst = frame_linf_close(got, ref, rtol=2e-7)
    assert ok, worst


def test_frame_count_rules(golden):
    g = golden("mfcc_ref.npz")
    c = S.mfcc_cases()
    assert O.compute_mfcc(c["len399"], 500).shape[0] == 0          # mfcc.c:117
    assert O.compute_mfcc(c["noise0"], 0).shape[0] == 0            # max_frames <= 0
    assert O.compute_mfcc(c["long"], 500).shape[0] == 500          # clamp, mfcc.c:137
    got = O.compute_mfcc(c["noise0"], 7)
    assert np.allclose(got, g["mfcc__noise0_max7"], rtol=0, atol=1e-5)
    assert g["mfcc__birdq_ch0"].shape == (148, 13)                 # BASELINE config 1


def test_silent_frames_are_exact_zero(golden):
    got = O.compute_mfcc(np.zeros(16000, np.float32), 500)
    assert got.shape == (98, 13) and not got.any()


def test_float64_fft_mode_within_gate(golden):
    """The reference's own fp32 noise vs an exact transform stays inside the gate."""
    g = golden("mfcc_ref.npz")
    for name in ("noise0", "stop_121417", "birdq_ch0"):
        got = O.compute_mfcc(_cases(g)[name], 500, O.default_cfg(fft_mode=O.FFT_FLOAT64))
        ok, worst = frame_linf_close(got, g["mfcc__" + name], rtol=1e-4)
        assert ok, (name, worst)


def test_frames_api_equals_clip_api():
    x = S.uniform_pm1(400 + 160 * 9, 5)
    cfg = O.default_cfg()
    clip = O.compute_mfcc(x, 500, cfg)
    frames = np.stack([x[160 * t: 160 * t + 400] for t in range(10)])
    assert np.array_equal(O.mfcc_frames(frames, cfg), clip)
    assert np.array_equal(O.mfcc_frames(frames, cfg, threads=3), clip)


def test_pcm16_conversion_known_answer(golden):
    """donut-classifier/data/*.wav.txt: channel-0 PCM16 / 32768 printed with %f."""
    k = golden("pcm_kat.npz")
    x = (k["pcm"][:, 0] / 32768.0)
    assert np.abs(x - k["dump"]).max() <= 5.1e-7


def test_librosa_default_mel_filterbank_vs_an_independent_restatement():
    """ORC_MELNORM_LIBROSA = librosa.filters.mel's defaults (Slaney's mel scale, htk = False, with Slaney's area
    normalisation): what librosa.feature.mfcc, hence cepstrum/train.py:45-52, builds.  Checked against the float64 numpy
    restatement of librosa's formulas in tools/pin_svm_libsvm.py (written from librosa's documentation for the SVM polarity
    pin, independently of the C code)."""
    import importlib.util
    import os
    from oracle import oracle as O
    spec = importlib.util.spec_from_file_location("pin_svm", os.path.join(os.path.dirname(__file__), "..", "tools", "pin_svm_libsvm.py"))
    pin = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pin)
    for sr, n_fft, n_mels in ((16000, 2048, 128), (22050, 2048, 128), (44100, 2048, 40), (16000, 512, 40)):
        got = O.mel_filterbank(sr, n_fft, n_mels, 0.0, sr / 2.0, O.MELNORM_LIBROSA)
        want = pin.slaney_mel(sr, n_fft, n_mels)
        assert got.shape == want.shape
        assert np.abs(got - want).max() <= 2e-7 * np.abs(want).max()
        assert (got.sum(axis=1) > 0).all()
