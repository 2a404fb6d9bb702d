#!/usr/bin/env python3
"""Pin the RBF-SVM tail against libsvm itself and against the reference's labelled audio.

Runs only in the build container (needs sklearn and, for part B, /root/reference).  Writes
tests/golden/svm_libsvm_ref.npz -- data only: feature vectors, and what libsvm / the labelled files say about them.

A. Arithmetic and label rule vs LIBSVM.  scrubjay_svm.onnx was exported by skl2onnx from
   SVC(kernel="rbf", probability=True) (cepstrum/train.py:72-75, convert_model.py:9-12); sklearn's SVC IS libsvm.
   The attributes decoded from the ONNX file (tools/decode_onnx_svm.py -> tests/golden/scrubjay_svm.npz) are fed,
   raw arrays and no unpickling, to sklearn.svm._libsvm.decision_function / predict_proba / predict:
       SV = support_vectors, nSV = vectors_per_class, sv_coef = coefficients, intercept = rho, probA / probB = prob_a / prob_b
   libsvm's decision value is sum_i sv_coef_i K(x, sv_i) - model.rho with model.rho = -intercept (libsvm_helper.c
   set_model), i.e. exactly ONNX SVMClassifier's "sum + rho".  A positive value votes for libsvm's first label (class 0).
   Recorded per vector: decision value, predict_proba (pairwise Platt sigmoid -> multiclass_probability) and the vote
   label (svm_predict: what sklearn's .predict returns in cepstrum/run.py; ONNX Runtime's SVMClassifier counts the same
   votes in SVC mode).  Two things a "sigmoid + arg max" reading of the ONNX attributes gets wrong are pinned here:
   (1) the label is the VOTE (decision > 0 -> class 0), which differs from the sigmoid's arg max for decision values
   between 0 and -prob_b / prob_a = 0.00796; (2) predict_proba is libsvm's multiclass_probability ITERATION (tolerance
   0.005 / k from the start point (1/2, 1/2)), which returns exactly (0.5, 0.5) in a dead zone |decision| <~ 0.005 and
   differs from the sigmoid by up to 5e-3 elsewhere.  Vectors bisected into that zone are part of the fixture.

B. Polarity vs the reference's own labelled audio.  Label 1 = "pos" = scrub jay (train.py:45).  Every labelled WAV of
   cepstrum/data/{pos,neg} and cepstrum/testing/{P_,N_}* (the MP3s cannot be decoded offline) goes through a float64
   numpy restatement of what train.py computes (librosa.load(sr=None) mono average, librosa.feature.mfcc(n_mfcc=20)
   defaults: n_fft 2048, hop 512, centred, Hann, 128 Slaney mel, power_to_db, ortho DCT-II; mean | std) and then through
   part A's libsvm model.  Most of these files were training data, so the right polarity classifies them almost
   perfectly and the flipped one almost never: the script asserts >= 85 % and stores features + file labels.
"""
from __future__ import annotations

import glob
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF = os.environ.get("DSP_REF", "/root/reference")


def libsvm_model(m):
    sv = np.ascontiguousarray(m["sv"], np.float64)
    n_sv = sv.shape[0]
    return dict(support=np.arange(n_sv, dtype=np.int32), SV=sv, nSV=np.ascontiguousarray(m["vectors_per_class"], np.int32),
                sv_coef=np.ascontiguousarray(m["coef"], np.float64).reshape(1, n_sv), intercept=np.array([float(m["rho"][0])], np.float64),
                probA=np.array([float(m["prob_a"][0])], np.float64), probB=np.array([float(m["prob_b"][0])], np.float64),
                svm_type=0, kernel="rbf", degree=3, gamma=float(m["kernel_params"][0]), coef0=0.0, cache_size=100.0)


def libsvm_eval(m, feats):
    """feats [N][40] raw features (float32, as the C program hands them to ORT) -> decision, proba[N][2], vote label."""
    from sklearn.svm import _libsvm
    z = (feats.astype(np.float32) - m["offset"]) * m["scale"]                # ONNX Scaler, float32
    z = np.ascontiguousarray(z, np.float64)
    kw = libsvm_model(m)
    dec = _libsvm.decision_function(z, **kw).reshape(-1)
    proba = _libsvm.predict_proba(z, **kw)
    vote = _libsvm.predict(z, **kw).astype(np.int64)
    return dec, proba, vote


# ---- part B: numpy restatement of train.py's feature extraction (float64) -------------------------------------------------
def read_wav_mono(path):
    w = wave.open(path)
    n, ch, sw = w.getnframes(), w.getnchannels(), w.getsampwidth()
    raw = w.readframes(n)
    if sw == 2:
        x = np.frombuffer(raw, "<i2").astype(np.float64) / 32768.0
    elif sw == 3:
        b = np.frombuffer(raw, np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float64) / 8388608.0
    else:
        raise ValueError("sample width")
    return x.reshape(-1, ch).mean(axis=1), w.getframerate()


def slaney_mel(sr, n_fft, n_mels=128):
    def hz_to_mel(f):
        f = np.asarray(f, np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) / (np.log(6.4) / 27.0), f / (200.0 / 3))

    def mel_to_hz(mm):
        mm = np.asarray(mm, np.float64)
        return np.where(mm >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (mm - 15.0)), (200.0 / 3) * mm)

    fft_f = np.linspace(0, sr / 2, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(0.0), hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fft_f[None, :]
    w = np.maximum(0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    return w * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]


def librosa_like_features(y, sr, n_mfcc=20, n_fft=2048, hop=512):
    from scipy.fft import dct
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    yp = np.pad(y, n_fft // 2)                                           # center=True, pad_mode="constant" (librosa >= 0.10)
    n_frames = 1 + (yp.size - n_fft) // hop
    mel = slaney_mel(sr, n_fft)
    out = np.empty((128, n_frames))
    for t0 in range(0, n_frames, 512):
        idx = (np.arange(t0, min(n_frames, t0 + 512)) * hop)[:, None] + np.arange(n_fft)[None, :]
        p = np.abs(np.fft.rfft(yp[idx] * win, axis=1)) ** 2
        out[:, t0:t0 + idx.shape[0]] = mel @ p.T
    db = 10 * np.log10(np.maximum(out, 1e-10))
    db = np.maximum(db, db.max() - 80.0)
    c = dct(db, axis=0, type=2, norm="ortho")[:n_mfcc]
    return np.concatenate([c.mean(axis=1), c.std(axis=1)]).astype(np.float32)


def labelled_files():
    out = []
    for lab, pat in ((1, "cepstrum/data/pos/*"), (0, "cepstrum/data/neg/*"), (1, "cepstrum/testing/P_*"), (0, "cepstrum/testing/N_*")):
        for p in sorted(glob.glob(os.path.join(REF, pat))):
            if p.lower().endswith(".wav") and os.path.getsize(p) > 4096:          # placeholders of missing blobs are tiny
                out.append((lab, p))
    return out


def main():
    m = dict(np.load(os.path.join(GOLDEN, "scrubjay_svm.npz")))
    rng = np.random.default_rng(20260104)
    # ---- A: 256 vectors around the training distribution + 64 bisected into the sliver where the two label rules disagree
    feats = (m["offset"][None, :] + rng.uniform(-2.5, 2.5, (256, 40)) / m["scale"][None, :]).astype(np.float32)
    dec, _, _ = libsvm_eval(m, feats)
    pos, neg = feats[dec > 0.05], feats[dec < -0.05]
    assert len(pos) >= 8 and len(neg) >= 8
    edge_hi = -float(m["prob_b"][0]) / float(m["prob_a"][0])                     # 0.00796: P = 0.5 exactly there
    sliver = []
    for i in range(64):
        a, b = pos[i % len(pos)].astype(np.float64), neg[(7 * i) % len(neg)].astype(np.float64)
        target = edge_hi * (i % 16 + 0.5) / 16 if i < 48 else (edge_hi * 1.5 if i % 2 else -edge_hi * 0.5)
        lo, hi = 0.0, 1.0                                                         # decision(a) > target > decision(b)
        for _ in range(60):
            mid = 0.5 * (lo + hi)
            d = libsvm_eval(m, ((1 - mid) * a + mid * b).astype(np.float32)[None, :])[0][0]
            if d > target: lo = mid
            else: hi = mid
        sliver.append(((1 - lo) * a + lo * b).astype(np.float32))
    feats = np.concatenate([feats, np.stack(sliver)])
    dec, proba, vote = libsvm_eval(m, feats)
    sig = 1.0 / (1.0 + np.exp(dec * float(m["prob_a"][0]) + float(m["prob_b"][0])))
    by_sigmoid = (1 - sig > sig).astype(np.int64)
    in_sliver = (dec > 0) & (dec < edge_hi)
    assert np.array_equal(vote, (dec <= 0).astype(np.int64))                      # svm_predict: decision > 0 votes class 0
    assert int(in_sliver.sum()) >= 16 and np.all(by_sigmoid[in_sliver] != vote[in_sliver])
    assert np.all(proba[in_sliver & (dec > 0.002) & (dec < 0.006)] == 0.5)        # the iteration's dead zone
    print(f"A: {len(feats)} vectors, decision in [{dec.min():.3f}, {dec.max():.3f}], vote label 1: {int(vote.sum())}; "
          f"{int(in_sliver.sum())} vectors with 0 < decision < {edge_hi:.5f} where sigmoid-arg-max would say 1 and libsvm says 0; "
          f"predict_proba vs the plain sigmoid: max |diff| {float(np.abs(proba[:, 0] - sig).max()):.2e}")
    out = dict(feat=feats, decision=dec, proba=proba, label_vote=vote)

    # ---- B: polarity on the reference's labelled WAVs
    files = labelled_files() if os.path.isdir(REF) else []
    if files:
        lf, ly, names = [], [], []
        for lab, p in files:
            y, sr = read_wav_mono(p)
            lf.append(librosa_like_features(y, sr)); ly.append(lab); names.append(os.path.basename(p))
        lf, ly = np.stack(lf), np.array(ly, np.int64)
        d, pr, v = libsvm_eval(m, lf)
        for nm, lab, dd, pp, vv in zip(names, ly, d, pr, v):
            print(f"B: {nm[:48]:48s} file label {lab}  decision {dd:+.3f}  P(1) {pp[1]:.3f}  libsvm predict {vv}")
        acc = float((v == ly).mean())
        print(f"B: {len(ly)} labelled WAVs, libsvm predict == file label on {acc:.0%}; flipped polarity would give {1 - acc:.0%}")
        assert acc >= 0.85
        out.update(labelled_feat=lf, labelled_y=ly, labelled_decision=d, labelled_proba=pr, labelled_vote=v)
    np.savez_compressed(os.path.join(GOLDEN, "svm_libsvm_ref.npz"), **out)
    print("wrote tests/golden/svm_libsvm_ref.npz")


if __name__ == "__main__":
    sys.exit(main())
