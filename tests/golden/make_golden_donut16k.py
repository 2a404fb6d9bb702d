#!/usr/bin/env python3
"""The donut classifier's own 16 kHz recordings as classify() goldens.

Runs only in the build container (needs /root/reference and oracle/_ref, the reference's classifier.cpp compiled unmodified).
Writes tests/golden/donut16k_ref.npz -- data only, per recording of donut-classifier/16k/*.wav (the 7 s one is cut to its first
3 s to keep the fixture small; birdQ is already in classifier_ref.npz):
    <name>__pcm        int16 [n][2]        the WAV's samples
    <name>__label      what the COMPILED REFERENCE's classify() returns on channel 0 / 32768 (classifier.c:292-297)
    <name>__midpoints  the compiled reference's find_midpoints() on the same signal
    <name>__mic_*      label / midpoints / sums under microphone/src/classifier.cpp's thresholds, from the oracle
    <name>__sums       band sums per midpoint from the oracle (zero rows after the first hit), which the tests pin to the compiled
                       reference on label, midpoints, filter and spectrogram bit for bit (tests/test_oracle_classifier.py)"""
import glob
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = os.environ.get("DSP_REF", "/root/reference")


def main():
    from oracle import oracle as O
    L = O.ref_classifier_lib()
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, "donut-classifier", "16k", "*.wav"))):
        name = os.path.basename(path)[:-4].replace("-", "_")
        if name.startswith("birdQ"):
            continue
        w = wave.open(path)
        assert w.getframerate() == 16000 and w.getsampwidth() == 2
        pcm = np.frombuffer(w.readframes(w.getnframes()), "<i2").reshape(-1, w.getnchannels()).copy()
        if pcm.shape[0] > 48000:
            pcm = pcm[:48000]
        x = (pcm[:, 0].astype(np.float32) / np.float32(32768.0)).astype(np.float32)
        label = int(L.ref_classify(x.copy(), x.size))
        mids = np.zeros(64, np.float32)
        n_mid = int(L.ref_find_midpoints(x.copy(), x.size, 16000, mids, 64)) if x.size >= 256 else 0
        olab, omids, osums = O.classify(x)
        assert olab == label and np.array_equal(omids, mids[:n_mid]), name
        print(f"{name}: {pcm.shape[0]} samples, reference label {label}, {n_mid} midpoints {np.round(mids[:n_mid], 3)}")
        # the same recording under the thresholds of microphone/src/classifier.cpp (45 dB midpoint threshold: these recordings
        # have midpoints there); that firmware file is not buildable here, the values are the oracle's
        mlab, mmids, msums = O.classify(x, O.CLASSIFY_MICROPHONE)
        print(f"    microphone thresholds: label {mlab}, midpoints {np.round(mmids, 3)}")
        out.update({f"{name}__pcm": pcm, f"{name}__label": np.int32(label), f"{name}__midpoints": mids[:n_mid].copy(), f"{name}__sums": osums,
                    f"{name}__mic_label": np.int32(mlab), f"{name}__mic_midpoints": mmids, f"{name}__mic_sums": msums})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "donut16k_ref.npz"), **out)
    print("wrote tests/golden/donut16k_ref.npz")


if __name__ == "__main__":
    sys.exit(main())
