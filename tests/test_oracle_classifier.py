"""CPU: Butterworth / spectrogram / classify oracle against the reference's
known-answer dumps and goldens from its compiled sync/lib/classifier.cpp."""
import numpy as np
import pytest

from oracle import oracle as O

CASES = ["noise", "burst_2k", "jay_like", "scrub_a", "scrub_b", "silence", "birdq_ch0_1s"]


def test_butter_tables(golden):
    g = golden("classifier_ref.npz")
    for lo, hi in ((1000, 3000), (3000, 7500)):
        ok, b, a = O.butter_bandpass(lo, hi)
        assert ok
        assert np.array_equal(b.astype(np.float32), g[f"b_{lo}_{hi}"])
        assert np.array_equal(a.astype(np.float32), g[f"a_{lo}_{hi}"])
    ok, _, _ = O.butter_bandpass(2000, 6000)   # classifier.cpp:184-189: unknown band -> false
    assert not ok and int(g["bad_band_ok"]) == 0


def test_iir_f64_postbutter_known_answer(golden):
    """donut-classifier/_postbutter.txt = DF-II in float64 with the coefficient
    block at classifier.c:323-341 on channel 0 of testing/1060-control.wav."""
    k = golden("iir_kat.npz")
    x = k["pcm"].astype(np.float64) / 32768.0
    y = O.iir_f64(x, k["b"], k["a"])
    ref = k["postbutter"]
    # the dump is printed with %e (7 significant digits); the delay line runs at
    # ~1e4 x the output level (b ~ 2e-4), so float64 rounding / FMA contraction
    # on the machine that wrote the dump adds ~1e-11 absolute
    assert np.abs(y - ref).max() <= 5e-7 * np.abs(ref).max()
    assert np.all(np.abs(y - ref) <= 5.1e-7 * np.abs(ref) + 1e-11)


def test_blobtimes_known_answer(golden):
    """donut-classifier/_blobtimes.txt: time bins where any PSD cell of the
    filtered signal exceeds 45 dB re 1e-12 (fs 96 kHz)."""
    k = golden("blobtimes_kat.npz")
    fs = int(k["fs"])
    x = k["pcm"].astype(np.float64) / 32768.0
    y = O.iir_f64(x, k["b"], k["a"])
    f, t, sxx = O.spectrogram_f64(y, fs)
    with np.errstate(divide="ignore"):
        db = 10 * np.log10(sxx / 1e-12)
    mine = t[(db > float(k["threshold_db"])).any(axis=0)]
    ref = k["blobtimes"]
    ref = ref[ref <= t[-1] + 1e-9]
    assert ref.size >= 10
    assert mine.size == ref.size
    assert np.abs(mine - ref).max() <= 1e-6


@pytest.mark.parametrize("name", CASES)
def test_filter_and_spectrogram_bit_exact(golden, name):
    g = golden("classifier_ref.npz")
    x = g[f"{name}__input"]
    y = O.iir_f32(x, g["b_3000_7500"], g["a_3000_7500"])
    assert np.array_equal(y, g[f"{name}__filtered"])
    f, t, sxx = O.spectrogram_f32(y)
    assert np.array_equal(f, g["freqs"]) and np.array_equal(t, g["times_16000"])
    assert np.array_equal(sxx, g[f"{name}__sxx"])


@pytest.mark.parametrize("name", CASES)
def test_midpoints_and_label(golden, name):
    g = golden("classifier_ref.npz")
    x = g[f"{name}__input"]
    assert np.array_equal(O.find_midpoints(x), g[f"{name}__midpoints"])
    label, mids, sums = O.classify(x)
    assert label == int(g[f"{name}__label"])


def test_label_one_is_exercised(golden):
    g = golden("classifier_ref.npz")
    assert {int(g[f"{n}__label"]) for n in CASES} == {0, 1}


def test_spectrogram_f64_matches_scipy():
    """compute_spectrogram == scipy.signal.spectrogram defaults."""
    from scipy import signal as ss
    from tests import signals as S
    x = S.uniform_pm1(4000, 3).astype(np.float64)
    f, t, sxx = O.spectrogram_f64(x, 16000)
    f2, t2, s2 = ss.spectrogram(x, fs=16000)
    assert np.allclose(f, f2) and np.allclose(t, t2)
    assert np.allclose(sxx, s2, rtol=1e-9, atol=1e-18)


def test_mfcc_stats_population_std():
    from tests import signals as S
    m = S.uniform_pm1(37 * 20, 8).reshape(37, 20) * 30
    out = O.mfcc_stats(m)
    assert np.allclose(out[:20], m.mean(0), rtol=1e-6, atol=1e-6)
    assert np.allclose(out[20:], m.std(0), rtol=1e-5, atol=1e-5)


def test_sum_intense_vs_the_compiled_reference(golden):
    """sum_intense of the reference's compiled classifier.cpp (tests/golden/make_golden_round2.py): real band-kept maps,
    a random map, and the clamp / swap branches of the index searches (classifier.cpp:383-412)."""
    s = golden("sum_intense_ref.npz")
    for c in range(int(s["n_cases"])):
        lo, hi, half, mid = (float(v) for v in s[f"c{c}_params"])
        got = O.sum_intense(lo, hi, half, s[f"c{c}_freqs"], s[f"c{c}_times"], s[f"c{c}_db"], mid)
        assert np.float32(got).tobytes() == s[f"c{c}_sum"].tobytes(), c


def test_donut_classifier_recordings_vs_compiled_reference(golden):
    """The donut classifier's own 16 kHz recordings (donut-classifier/16k/*.wav, fixture tests/golden/donut16k_ref.npz):
    the oracle's classify() and find_midpoints() against what the compiled reference returned for them."""
    from oracle import oracle as O
    g = golden("donut16k_ref.npz")
    names = sorted({k.split("__")[0] for k in g.files})
    assert len(names) == 6
    for n in names:
        x = (g[n + "__pcm"][:, 0].astype(np.float32) / np.float32(32768.0)).astype(np.float32)       # classifier.c:292-297
        lab, mids, sums = O.classify(x)
        assert lab == int(g[n + "__label"]) and np.array_equal(mids, g[n + "__midpoints"]) and np.array_equal(sums, g[n + "__sums"])
        mlab, mmids, msums = O.classify(x, O.CLASSIFY_MICROPHONE)
        assert mlab == int(g[n + "__mic_label"]) and np.array_equal(mmids, g[n + "__mic_midpoints"]) and np.array_equal(msums, g[n + "__mic_sums"])
    assert sum(len(g[n + "__mic_midpoints"]) for n in names) >= 8
