"""The oracle's restatement of the float64 classifier, donut-classifier/classifier.c (oracle/classify_f64_oracle.c).  The file
itself cannot be built here (sndfile.h, fftw3.h), so the chain has no compiled-reference output; its stages are pinned by the
reference's own dumps (test_oracle_classifier.py: _postbutter.txt for the filter, _blobtimes.txt for spectrogram + 45 dB mask)
and here the chain is checked against an independent numpy / scipy evaluation written from the Python prototype's calls
(classifier16k.py: scipy.signal.lfilter on the literal coefficients, scipy.signal.spectrogram) and plain numpy for the tail."""
import numpy as np
from scipy import signal as ss

from oracle import oracle as O
from tests import signals as S

DONUT = dict(keep_lo=0.70, keep_hi=0.85, midpoint_db=45.0, middle_max=75.0, above_min=300.0, below_min=100.0)


def numpy_classify(x, cfg=DONUT, fs=16000):
    x = np.asarray(x, np.float64)
    _, b_bp, a_bp = O.butter_bandpass(3000, 7500)
    _, b_mp, a_mp = O.butter_bandpass(1000, 3000)

    def db_map(sig):
        f, t, s = ss.spectrogram(sig, fs=fs)
        with np.errstate(divide="ignore", invalid="ignore"):
            d = np.where(s > 0, 10 * np.log10(s / 1e-12), np.nan)
        return f, t, d

    f, t, d_mp = db_map(ss.lfilter(b_mp, a_mp, x))
    blob = t[np.nansum(d_mp > cfg["midpoint_db"], axis=0) > 0]
    mids, i0 = [], 0
    while i0 < blob.size:
        i1 = i0
        while i1 + 1 < blob.size and blob[i1 + 1] - blob[i1] <= 0.05:
            i1 += 1
        if blob[i1] - blob[i0] >= 0.15:
            mids.append(blob[i0:i1 + 1].sum() / (i1 - i0 + 1))
        i0 = i1 + 1
    f, t, d = db_map(ss.lfilter(b_bp, a_bp, x))
    v = (d - np.nanmin(d)) / (np.nanmax(d) - np.nanmin(d))
    kept = np.where((v > cfg["keep_lo"]) & (v < cfg["keep_hi"]), v, np.nan)

    def band(lo, hi, half, m):
        fi = np.where((f >= lo) & (f <= hi))[0]
        ti = np.where((t >= m - half) & (t <= m + half))[0]
        return float(np.nansum(kept[np.ix_(fi, ti)])) if fi.size and ti.size else 0.0

    label, sums = 0, []
    for m in mids:
        s3 = (band(5000, 7000, 0.18, m), band(2500, 5000, 0.05, m), band(500, 2500, 0.18, m))
        sums.append(s3)
        if s3[1] < cfg["middle_max"] and s3[0] > cfg["above_min"] and s3[2] > cfg["below_min"]:
            label = 1
            break
    return label, np.array(mids), np.array(sums).reshape(-1, 3)


def test_chain_against_numpy_scipy():
    seen = set()
    for name, x in S.classify_cases().items():
        lab, mids, sums = O.classify_f64(x.astype(np.float64))
        nlab, nmids, nsums = numpy_classify(x)
        assert lab == nlab and mids.shape == nmids.shape, name
        assert np.allclose(mids, nmids, rtol=0, atol=1e-12)
        assert sums[:len(nsums)].shape == nsums.shape and np.allclose(sums[:len(nsums)], nsums, rtol=1e-9, atol=1e-9), name
        seen.add(lab)
    assert seen == {0, 1}


def test_thresholds_are_the_files_doubles():
    # cfg = None is classifier.c's own set; the same numbers passed explicitly give the same answer, others do not
    x = S.classify_cases()["scrub_a"].astype(np.float64)
    assert O.classify_f64(x)[0] == O.classify_f64(x, DONUT)[0] == 1
    assert O.classify_f64(x, dict(DONUT, above_min=1e9))[0] == 0
    assert O.find_midpoints_f64(x).size == O.classify_f64(x)[1].size


def test_float64_chain_agrees_with_the_float32_twin_on_labels():
    # same thresholds on sync/lib's float32 arithmetic (orc_classify_with): labels agree on the test clips, sums to ~1 %
    for name, x in S.classify_cases().items():
        l32, m32, s32 = O.classify(x, (0.70, 0.85, 45.0, 75.0, 300.0, 100.0))
        l64, m64, s64 = O.classify_f64(x.astype(np.float64))
        assert l32 == l64 and len(m32) == len(m64), name
        assert np.allclose(m32, m64, atol=1e-6)
