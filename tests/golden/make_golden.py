#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz from the reference checkout.

Runs only where /root/reference exists (the build container).  It executes the
reference's OWN code -- its C sources compiled unmodified into oracle/_ref/ by
oracle/Makefile -- on seeded / excerpted inputs and stores inputs + outputs as
small numpy fixtures.  No reference source text is stored, only data:
inputs (PCM excerpts of the reference's WAV files, seeds), expected outputs,
and digests of the reference's constant tables.

    python tests/golden/make_golden.py [--ref /root/reference]

Fixtures written (all float32 unless noted):
  mfcc_ref.npz        compute_mfcc goldens (2fa/audio/word/c/mfcc.c:108)
  tables_ref.npz      digests of HANN_WINDOW / MEL_FILTER / DCT_MATRIX (mfcc_params.h)
  iir_kat.npz         donut-classifier/_postbutter.txt excerpt + matching input
  blobtimes_kat.npz   donut-classifier/_blobtimes.txt + PCM excerpt
  classifier_ref.npz  butter filter / spectrogram / midpoints / classify goldens
                      (sync/lib/classifier.cpp)
  pcm_kat.npz         donut-classifier/data/*.wav.txt PCM16->float dump excerpt
  librosa_mfcc_kat.npz  TEST_MFCC of 2fa/audio/word/c/test_mfcc.h (librosa-mode MFCC of stop_121417.wav)
"""
from __future__ import annotations

import argparse
import hashlib
import os
import sys
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from tests import signals as S  # noqa: E402  (seeded input recipes shared with the tests)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def read_wav_i16(path):
    w = wave.open(path)
    assert w.getsampwidth() == 2
    d = np.frombuffer(w.readframes(w.getnframes()), np.int16)
    return d.reshape(-1, w.getnchannels()), w.getframerate()


STOP_CLIPS = [  # 2fa/audio/data/testing: clips on which the reference's net is not in its all-ReLUs-dead plateau
    "stop_121417.wav", "four__common_voice_en_20726027.wav", "four__common_voice_en_22102474.wav",
    "off__common_voice_en_20947144.wav", "house__common_voice_en_21901759.wav", "down__common_voice_en_20545290.wav",
    "bed__common_voice_en_504965.wav",
]


def consumers(ref):
    """Goldens for the consumers of the MFCC matrix: the reference's stop detector
    (2fa/audio/word/c/stop_detector.c + audio_classifier_inference.c, parameters model_params.h) and speaker GMM
    (2fa/audio/pico-audio/src/speaker_gmm.c, parameters gmm_params.inc), both compiled from the reference's
    own sources into oracle/_ref.  The trained parameters are stored as arrays (data, read out of the compiled
    reference), the reference's outputs as the expected values."""
    L = O.ref_stop_lib()
    model = O.ref_stop_model()
    out = {k: v for k, v in model.items() if isinstance(v, np.ndarray)}
    out["n_coef"], out["max_frames"] = np.int32(model["n_coef"]), np.int32(model["max_frames"])
    np.savez_compressed(os.path.join(HERE, "stop_model.npz"), **out)
    g = {}
    rng = np.random.default_rng(11)
    for i, name in enumerate(STOP_CLIPS):
        pcm, sr = read_wav_i16(os.path.join(ref, "2fa/audio/data/testing", name))
        assert sr == 16000 and pcm.shape[1] == 1
        x = (pcm[:, 0] / np.float32(32768.0)).astype(np.float32)
        g[f"clip{i}__pcm"] = pcm[:, 0].copy()
        g[f"clip{i}__prob"] = np.float32(L.classify_signal(x, x.size))
        print(f"classify_signal {name:45s} -> {g[f'clip{i}__prob']:.6f}")
    # feature-level cases (audio_classifier_predict): perturbations around the scaler mean keep the net off its plateaus
    feats = np.stack([model["scaler_mean"] + rng.standard_normal(6500).astype(np.float32) * model["scaler_scale"] * np.float32(s)
                      for s in (0.0, 0.5, 1.0, 2.0, 4.0, 1.0, 1.0, 3.0)]).astype(np.float32)
    g["feats"] = feats
    g["feats_prob"] = np.array([L.audio_classifier_predict(f) for f in feats], np.float32)
    print("audio_classifier_predict on synthetic features:", g["feats_prob"])
    np.savez_compressed(os.path.join(HERE, "stop_ref.npz"), **g)

    G = O.ref_gmm_lib()
    target, ubm = O.ref_gmm_params()
    s = {"target_means": target["means"], "target_inv_covs": target["inv_covs"], "target_log_consts": target["log_consts"],
         "ubm_means": ubm["means"], "ubm_inv_covs": ubm["inv_covs"], "ubm_log_consts": ubm["log_consts"]}
    # Q6 conversion, including values beyond int16 after the x64 (speaker_gmm.c:118-122)
    xs = np.array([0.0, 0.49, -0.49, 1.0 / 64, -1.0 / 64, 3.999, -3.999, 511.99, -512.0, 600.0, -700.25, 1e4, -1e4], np.float32)
    q = np.empty(xs.size, np.int16)
    G.float_to_g6int16_arr(xs.copy(), q, xs.size)
    s["q6_in"], s["q6_out"] = xs, q
    # per-frame log-likelihoods and per-clip LLR means on the reference's own MFCC of the stop clips
    for i, name in enumerate(STOP_CLIPS[:4]):
        pcm, _ = read_wav_i16(os.path.join(ref, "2fa/audio/data/testing", name))
        x = (pcm[:, 0] / np.float32(32768.0)).astype(np.float32)
        mf = O.ref_compute_mfcc(x, 500)
        xq = np.empty(mf.shape, np.int16)
        G.float_to_g6int16_arr(mf.reshape(-1).copy(), xq.reshape(-1), mf.size)
        s[f"clip{i}__mfcc"] = mf
        s[f"clip{i}__ll_target"] = np.array([G.target_gmm_log_likelihood(r.copy()) for r in xq], np.int64)
        s[f"clip{i}__ll_ubm"] = np.array([G.ubm_gmm_log_likelihood(r.copy()) for r in xq], np.int64)
        s[f"clip{i}__llr_mean"] = np.int64(G.mfcc_target_speaker_llr_mean(mf.reshape(-1).copy(), mf.shape[0]))
        s[f"clip{i}__label"] = np.int32(G.classify_speaker(mf.reshape(-1).copy(), mf.shape[0]))
        print(f"speaker gmm {name:45s} llr_mean={s[f'clip{i}__llr_mean']} label={s[f'clip{i}__label']}")
    # synthetic frames near the mixture means (so that different mixtures win)
    synth = (target["means"][rng.integers(0, 32, 200)] / np.float32(64.0) + rng.standard_normal((200, 13)).astype(np.float32) * np.float32(0.3)).astype(np.float32)
    s["synth__mfcc"] = synth
    s["synth__llr_mean"] = np.int64(G.mfcc_target_speaker_llr_mean(synth.reshape(-1).copy(), 200))
    s["synth__label"] = np.int32(G.classify_speaker(synth.reshape(-1).copy(), 200))
    print("speaker gmm synthetic:", s["synth__llr_mean"], s["synth__label"])
    np.savez_compressed(os.path.join(HERE, "speaker_gmm_ref.npz"), **s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="", help="'consumers': regenerate only stop_model / stop_ref / speaker_gmm_ref")
    args = ap.parse_args()
    ref = args.ref
    O.build(force=True)
    assert O.have_ref(), "oracle/_ref not built (reference checkout missing?)"
    if args.only == "consumers":
        consumers(ref)
        return

    # ---- compute_mfcc goldens ------------------------------------------------
    cases = {}
    for name, sig in S.mfcc_cases().items():
        cases[name] = sig
    bird, sr = read_wav_i16(os.path.join(ref, "sound-processing/birdQ_stereo_16k.wav"))
    assert sr == 16000 and bird.shape == (24029, 2)
    stop, sr = read_wav_i16(os.path.join(ref, "2fa/audio/data/testing/stop_121417.wav"))
    assert sr == 16000 and stop.shape[1] == 1
    out = {"birdq_pcm": bird.copy(), "stop_pcm": stop[:, 0].copy()}
    # channel 0 convention: donut-classifier/classifier.c:292-297
    cases["birdq_ch0"] = (bird[:, 0] / np.float32(32768.0)).astype(np.float32)
    # stereo average convention: 2fa/audio/word/c/main_test.c:205-217
    cases["birdq_avg"] = (np.float32(0.5) * (bird[:, 0] / np.float32(32768.0) + bird[:, 1] / np.float32(32768.0))).astype(np.float32)
    cases["stop_121417"] = (stop[:, 0] / np.float32(32768.0)).astype(np.float32)
    for name, sig in cases.items():
        got = O.ref_compute_mfcc(sig, 500)
        out["mfcc__" + name] = got
        if name == "chirp":
            out["input__chirp"] = sig  # np.sin may differ by an ulp across libm builds
        print(f"mfcc {name:20s} n={sig.size:6d} -> T={got.shape[0]}")
    # max_frames clamp (mfcc.c:137-139)
    out["mfcc__noise0_max7"] = O.ref_compute_mfcc(cases["noise0"], 7)
    # reference fft_real_forward on one frame (mfcc.c:16)
    spec = np.empty(1024, np.float32)
    O.ref_mfcc_lib().fft_real_forward(np.ascontiguousarray(cases["noise0"][:400]), spec)
    out["fft__noise0_frame0"] = spec
    np.savez_compressed(os.path.join(HERE, "mfcc_ref.npz"), **out)

    # ---- table digests -------------------------------------------------------
    hann, mel, dct = O.ref_tables()
    odct = O.dct_ortho(13, 40)
    mism = np.flatnonzero(odct.reshape(-1) != dct.reshape(-1)).astype(np.int32)
    np.savez_compressed(
        os.path.join(HERE, "tables_ref.npz"),
        hann_sha=sha(hann), mel_sha=sha(mel), dct_sha=sha(dct),
        mel_nonzero=np.int32((mel != 0).sum()),
        # the reference DCT was exported with numpy's float32 SIMD cos; glibc cosf
        # differs by 1 ulp in a few dozen entries: keep those entries so the
        # oracle table can be checked digest-exact after patching them.
        dct_mismatch_idx=mism, dct_mismatch_val=dct.reshape(-1)[mism],
    )
    print("tables: hann/mel exact =", np.array_equal(O.window(0, 400), hann), np.array_equal(O.mel_filterbank(), mel),
          " dct 1-ulp entries:", mism.size)

    # ---- IIR known-answer: _postbutter.txt ------------------------------------
    ctl, sr = read_wav_i16(os.path.join(ref, "donut-classifier/testing/1060-control.wav"))
    assert sr == 96000
    n_iir = 8192
    post = np.loadtxt(os.path.join(ref, "donut-classifier/_postbutter.txt"), max_rows=n_iir)
    # coefficient block the dump was produced with: donut-classifier/classifier.c:323-341
    b96 = np.array([0.00021314, 0., -0.00085255, 0., 0.00127883, 0., -0.00085255, 0., 0.00021314])
    a96 = np.array([1., -7.12847885, 22.41882266, -40.62891245, 46.40780141, -34.21333503, 15.89913237, -4.25840048, 0.50337536])
    n_blob = 65536
    np.savez_compressed(os.path.join(HERE, "iir_kat.npz"), pcm=ctl[:n_iir, 0].copy(), b=b96, a=a96, postbutter=post)
    blob = np.loadtxt(os.path.join(ref, "donut-classifier/_blobtimes.txt"))
    np.savez_compressed(os.path.join(HERE, "blobtimes_kat.npz"), pcm=ctl[:n_blob, 0].copy(), b=b96, a=a96,
                        blobtimes=blob, fs=np.int32(96000), threshold_db=np.float64(45.0))

    # ---- classifier goldens ---------------------------------------------------
    L = O.ref_classifier_lib()
    cout = {}
    for lo, hi in ((1000, 3000), (3000, 7500)):
        b = np.zeros(9, np.float32); a = np.zeros(9, np.float32)
        assert L.ref_butter_bandpass(lo, hi, b, a) == 1
        cout[f"b_{lo}_{hi}"] = b; cout[f"a_{lo}_{hi}"] = a
    b = np.zeros(9, np.float32); a = np.zeros(9, np.float32)
    cout["bad_band_ok"] = np.int32(L.ref_butter_bandpass(2000, 6000, b, a))
    ccases = S.classify_cases()
    ccases["birdq_ch0_1s"] = cases["birdq_ch0"][:16000].copy()
    for name, sig in ccases.items():
        sig = np.ascontiguousarray(sig, np.float32)
        y = np.empty_like(sig)
        L.ref_butter_bandpass_filter(sig.copy(), sig.size, cout["b_3000_7500"].copy(), cout["a_3000_7500"].copy(), y)
        T = O.lib().orc_spectrogram_bins(sig.size)
        fr = np.empty(129, np.float32); tm = np.empty(T, np.float32); sx = np.empty((129, T), np.float32)
        L.ref_compute_spectrogram(y.copy(), y.size, 16000, fr, tm, sx.reshape(-1))
        mids = np.zeros(64, np.float32)
        nm = L.ref_find_midpoints(sig.copy(), sig.size, 16000, mids, 64)
        label = L.ref_classify(sig.copy(), sig.size)
        cout[f"{name}__input"] = sig
        cout[f"{name}__filtered"] = y
        cout[f"{name}__sxx"] = sx
        cout[f"{name}__midpoints"] = mids[:nm].copy()
        cout[f"{name}__label"] = np.int32(label)
        print(f"classify {name:16s} T={T} midpoints={nm} label={label}")
    cout["freqs"] = fr; cout["times_16000"] = tm
    np.savez_compressed(os.path.join(HERE, "classifier_ref.npz"), **cout)

    # ---- the reference's only in-repo MFCC golden: TEST_MFCC (2fa/audio/word/c/test_mfcc.h:8) -------
    # librosa-mode features of data/testing/stop_121417.wav, coefficient-major [13][1000], zero padded
    # (exporter: 2fa/audio/word/python/export_test_mfcc.py:21-39).  Numbers only, parsed from the header.
    import re
    txt = open(os.path.join(ref, "2fa/audio/word/c/test_mfcc.h")).read()
    body = txt[txt.index("TEST_MFCC[13000]"):]
    vals = np.array([float(v) for v in re.findall(r"[-+]?\d\.\d+e[-+]\d+", body)], np.float32)
    assert vals.size == 13000
    np.savez_compressed(os.path.join(HERE, "librosa_mfcc_kat.npz"), test_mfcc=vals.reshape(13, 1000), pcm=stop[:, 0].copy())
    print("TEST_MFCC golden:", vals.reshape(13, 1000)[:, :3])

    # ---- PCM16 -> float dump (donut-classifier/classifier.c:64-81) -------------
    txt = os.path.join(ref, "donut-classifier/data/birdQ_stereo_16k.wav.txt")
    if os.path.exists(txt):
        vals = np.array(open(txt).read().split(",")[:4096], dtype=np.float64)  # one comma-separated row, "%f"
        np.savez_compressed(os.path.join(HERE, "pcm_kat.npz"), pcm=bird[: vals.shape[0]].copy(), dump=vals)
        print("pcm dump rows:", vals.shape)

    consumers(ref)


if __name__ == "__main__":
    main()
