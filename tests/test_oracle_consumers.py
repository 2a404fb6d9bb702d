"""Oracle restatements of the MFCC consumers and the resampler (8f-4) against goldens from
the reference's own compiled sources (tests/golden/make_golden.py --only consumers):
  stop detector   2fa/audio/word/c/stop_detector.c:12-55, audio_classifier_inference.c:18-90, model_params.h
  speaker GMM     2fa/audio/pico-audio/src/speaker_gmm.c:29-141, gmm_params.inc
  resampler       sync/particle/main.cpp:62-77 (firmware file, not buildable here: checked against an independent
                  numpy float32 restatement only -> parity unpinned)"""
import numpy as np
import pytest

from oracle import oracle as O


def _model(golden):
    return dict(golden("stop_model.npz"))


def _gmms(golden):
    s = golden("speaker_gmm_ref.npz")
    t = {k: s[f"target_{k}"] for k in ("means", "inv_covs", "log_consts")}
    u = {k: s[f"ubm_{k}"] for k in ("means", "inv_covs", "log_consts")}
    return s, t, u


def test_stop_net_matches_reference_on_features(golden):
    m, g = _model(golden), golden("stop_ref.npz")
    got = np.array([O.stop_predict(m, f) for f in g["feats"]], np.float32)
    assert np.array_equal(got, g["feats_prob"])                 # same order of fp32 operations: bit-exact
    assert 0.3 < got[1] < 0.9 and got[4] > 0.99                 # the cases are not all on the dead-ReLU plateau


def test_classify_signal_matches_reference_on_its_test_clips(golden):
    m, g = _model(golden), golden("stop_ref.npz")
    for i in range(7):
        x = (g[f"clip{i}__pcm"] / np.float32(32768.0)).astype(np.float32)
        p = O.classify_signal(m, x)
        assert abs(p - float(g[f"clip{i}__prob"])) <= 2e-6, i    # compute_mfcc restatement is within the DCT-table ulp


def test_stop_features_layout_and_clamp(golden):
    m = _model(golden)
    mf = np.arange(3 * 13, dtype=np.float32).reshape(3, 13)
    f = O.stop_features(m, mf).reshape(13, 500)
    assert np.array_equal(f[:, :3], mf.T) and not f[:, 3:].any()           # coefficient-major, zero padded
    big = np.ones((600, 13), np.float32)
    assert O.stop_features(m, big).reshape(13, 500).all()                    # truncated at max_frames


def test_q6_conversion_matches_reference(golden):
    s, _, _ = _gmms(golden)
    assert np.array_equal(O.float_to_q6(s["q6_in"]), s["q6_out"])


def test_gmm_log_likelihoods_bit_exact(golden):
    s, t, u = _gmms(golden)
    for i in range(4):
        mf = s[f"clip{i}__mfcc"]
        xq = O.float_to_q6(mf)
        assert np.array_equal(np.array([O.gmm_log_likelihood(t, r) for r in xq], np.int64), s[f"clip{i}__ll_target"])
        assert np.array_equal(np.array([O.gmm_log_likelihood(u, r) for r in xq], np.int64), s[f"clip{i}__ll_ubm"])
        assert O.speaker_llr_mean(t, u, mf) == int(s[f"clip{i}__llr_mean"])
        assert O.classify_speaker(t, u, mf) == int(s[f"clip{i}__label"])
    assert O.speaker_llr_mean(t, u, s["synth__mfcc"]) == int(s["synth__llr_mean"])
    assert O.classify_speaker(t, u, s["synth__mfcc"]) == int(s["synth__label"]) == 1


def test_upsample_linear_against_numpy_float32():
    rng = np.random.default_rng(3)
    for old, new in ((8000, 16000), (7, 19), (100, 100), (2, 5)):
        x = rng.standard_normal(old).astype(np.float32)
        got = O.upsample_linear(x, new)
        i = np.arange(new, dtype=np.float32)
        idx = i * (np.float32(old - 1) / np.float32(new - 1))
        lo = np.floor(idx).astype(np.int64)
        hi = np.where(lo == old - 1, old - 1, lo + 1)
        frac = idx - lo.astype(np.float32)
        ref = x[lo] + (x[hi] - x[lo]) * frac
        assert np.array_equal(got, ref.astype(np.float32)), (old, new)
        assert got[0] == x[0] and abs(got[-1] - x[-1]) <= 1e-6 * max(1.0, abs(x[-1]))
