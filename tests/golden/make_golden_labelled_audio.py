#!/usr/bin/env python3
"""Two of the reference's labelled test recordings as an end-to-end fixture for the scrub-jay path.

Runs only in the build container (needs /root/reference and sklearn).  Writes tests/golden/labelled_audio.npz -- data only:
    <name>__pcm    int16 [n][channels]   the WAV's samples as they are in the file
    <name>__sr     sample rate           <name>__label  1 = scrub jay (cepstrum/train.py:45), from the file's P_ / N_ prefix
    <name>__feat   float32 [40]          what cepstrum/train.py:45-52 computes for the file (librosa.load(sr=None) mono average,
                                         librosa.feature.mfcc(n_mfcc=20) defaults, mean | std), by the float64 numpy restatement
                                         tools/pin_svm_libsvm.py:librosa_like_features
    <name>__decision / __proba / __vote  libsvm's outputs for that vector with the decoded scrubjay_svm.onnx model
Files: cepstrum/testing/P_1363v2-sj-short.WAV (0.5 s, 96 kHz stereo) and cepstrum/testing/N_1809v2-not-sj.wav (1.7 s, 96 kHz
stereo): the two shortest labelled WAVs (the MP3s cannot be decoded offline)."""
import importlib.util
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("DSP_REF", "/root/reference")
spec = importlib.util.spec_from_file_location("pin_svm", os.path.join(ROOT, "tools", "pin_svm_libsvm.py"))
pin = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pin)


def main():
    m = dict(np.load(os.path.join(ROOT, "tests", "golden", "scrubjay_svm.npz")))
    out = {}
    for name, rel, label in (("sj_short", "cepstrum/testing/P_1363v2-sj-short.WAV", 1), ("not_sj", "cepstrum/testing/N_1809v2-not-sj.wav", 0)):
        path = os.path.join(REF, rel)
        w = wave.open(path)
        assert w.getsampwidth() == 2
        pcm = np.frombuffer(w.readframes(w.getnframes()), "<i2").reshape(-1, w.getnchannels()).copy()
        y, sr = pin.read_wav_mono(path)
        feat = pin.librosa_like_features(y, sr)
        dec, proba, vote = pin.libsvm_eval(m, feat[None, :])
        print(f"{name}: {pcm.shape} @ {sr} Hz, label {label}, decision {dec[0]:+.4f}, P(1) {proba[0, 1]:.4f}, libsvm predict {vote[0]}")
        assert int(vote[0]) == label
        out.update({f"{name}__pcm": pcm, f"{name}__sr": np.int32(sr), f"{name}__label": np.int32(label), f"{name}__feat": feat,
                    f"{name}__decision": dec[0], f"{name}__proba": proba[0], f"{name}__vote": np.int64(vote[0])})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "labelled_audio.npz"), **out)
    print("wrote tests/golden/labelled_audio.npz")


if __name__ == "__main__":
    sys.exit(main())
