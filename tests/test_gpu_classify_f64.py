"""GPU: the float64 classifier of donut-classifier/classifier.c (dsp_classify_batch_*_f64: both Butterworth filters, both
spectrograms, dB maps, 45 dB midpoints, normalisation, keep band, band sums and rule in double) against the oracle's float64
restatement (oracle/classify_f64_oracle.c).  The reference transforms with FFTW (unvendored): the GPU spectrogram is a float64
transform of its own (batches: a 128-point complex FFT per wavefront; DSP_AMD_F64_DFT=1: the direct DFT), so the bar is tolerance
-- labels and midpoint counts equal, midpoints to 1e-12 s, band sums to 1e-8 relative."""
import os

import numpy as np
import pytest

from tests import signals as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dsp():
    import dsp_amd
    return dsp_amd


def _check(got_label, got_trace, x, cfg=None, what=""):
    from oracle import oracle as O
    olab, omids, osums = O.classify_f64(np.asarray(x, np.float64), cfg)
    mids, sums = got_trace
    assert int(got_label) == olab, what
    assert mids.shape == omids.shape and np.allclose(mids, omids, rtol=0, atol=1e-12), what
    # the reference stops at the first midpoint that fires the rule: rows after it are 0 on both sides
    assert sums.shape == osums.shape and np.allclose(sums, osums, rtol=1e-8, atol=1e-8), what
    return olab, len(omids)


def test_classify_cases_vs_oracle(dsp):
    cases = S.classify_cases()
    clips = np.stack([c.astype(np.float64) for c in cases.values()])
    labels, trace = dsp.classify_batch_f64(clips, with_trace=True)
    seen, n_mid = set(), 0
    for i, name in enumerate(cases):
        lab, k = _check(labels[i], trace[i], clips[i], what=name)
        seen.add(lab)
        n_mid += k
    assert seen == {0, 1} and n_mid >= 5
    assert np.array_equal(dsp.classify_batch_f64(clips), labels)            # without the trace: same labels


def test_other_thresholds_and_the_rule(dsp):
    x = S.classify_cases()["scrub_a"].astype(np.float64)[None, :]
    donut = (0.70, 0.85, 45.0, 75.0, 300.0, 100.0)
    assert dsp.classify_batch_f64(x)[0] == dsp.classify_batch_f64(x, config=donut)[0] == 1
    for cfg in ((0.70, 0.85, 45.0, 75.0, 1e9, 100.0), (0.65, 0.80, 70.0, 100.0, 200.0, 80.0), (0.70, 0.85, 45.0, 50.0, 200.0, 200.0)):
        labels, trace = dsp.classify_batch_f64(x, with_trace=True, config=cfg)
        _check(labels[0], trace[0], x[0], dict(zip(("keep_lo", "keep_hi", "midpoint_db", "middle_max", "above_min", "below_min"), cfg)), str(cfg))
    with pytest.raises(dsp.DspError):
        dsp.classify_batch_f64(x, config=(0.9, 0.8, 45.0, 75.0, 300.0, 100.0))      # keep_lo >= keep_hi


def test_donut_classifier_recordings_in_float64(dsp, golden):
    """The classifier's own 16 kHz recordings (donut-classifier/16k/*.wav, channel 0 / 32768.0 as classifier.c:55-59)."""
    g = golden("donut16k_ref.npz")
    names = sorted({k.split("__")[0] for k in g.files})
    n_mid = 0
    for n in names:
        x = g[n + "__pcm"][:, 0].astype(np.float64) / 32768.0
        labels, trace = dsp.classify_batch_f64(x[None, :], with_trace=True)
        n_mid += _check(labels[0], trace[0], x, what=n)[1]
    assert n_mid >= 5


def test_lengths_sub_batches_and_the_device_entry_point(dsp):
    import torch
    from oracle import oracle as O
    # shorter than one spectrogram segment -> label 0, no midpoints; exactly one segment; odd lengths
    for n in (10, 255):
        labels, trace = dsp.classify_batch_f64(np.ones((3, n)), with_trace=True)
        assert not labels.any() and all(len(m) == 0 for m, _ in trace)
    for n in (256, 479, 480, 5001):
        x = S.uniform_pm1(n, 40 + n).astype(np.float64)[None, :] * 0.3
        labels, trace = dsp.classify_batch_f64(x, with_trace=True)
        _check(labels[0], trace[0], x[0], what=f"len{n}")
    # more clips than one scratch pass (the library reads DSP_AMD_F64_SUB_BATCH per call; default 65 536): every clip is its own
    # problem, whatever its place in the batch
    call = S.classify_cases()["scrub_a"].astype(np.float64)
    rng = np.random.default_rng(5)
    clips = rng.uniform(-0.05, 0.05, (2100, 16000))
    clips[::7] = clips[::7] * 0.01 + call                          # the call over a quiet floor: label 1
    os.environ["DSP_AMD_F64_SUB_BATCH"] = "2048"
    try:
        labels = dsp.classify_batch_f64(clips)
        assert np.array_equal(dsp.classify_device_f64(torch.from_numpy(clips).cuda()).cpu().numpy(), labels)
    finally:
        del os.environ["DSP_AMD_F64_SUB_BATCH"]
    assert np.array_equal(dsp.classify_batch_f64(clips), labels)   # one pass: the same labels
    for i in (0, 1, 6, 7, 2047, 2048, 2093, 2099):
        assert labels[i] == O.classify_f64(clips[i])[0], i
    assert not labels[1::7].any() and labels[::7].sum() >= 290
    assert labels[::7].tolist() == [O.classify_f64(c)[0] for c in clips[::7]]     # every clip with the call (one of the 300 misses the rule)
    dl = dsp.classify_device_f64(torch.from_numpy(clips).cuda())
    assert np.array_equal(dl.cpu().numpy(), labels)
    # a strided device batch (rows padded): same labels
    padded = torch.zeros((64, 16016), dtype=torch.float64, device="cuda")
    padded[:, :16000] = torch.from_numpy(clips[:64]).cuda()
    assert np.array_equal(dsp.classify_device_f64(padded[:, :16000]).cpu().numpy(), labels[:64])


def test_fft_spectrogram_path_against_the_direct_dft_path(dsp):
    """The batch path's spectrogram is a 128-point complex FFT per wavefront (frame-major maps); DSP_AMD_F64_DFT=1 (read per call)
    routes the same batch through the direct 256-point DFT ([129][T] maps) -- two independent transforms and two map layouts under
    the same tail: labels and midpoints equal, band sums to 1e-10 relative."""
    cases = S.classify_cases()
    rng = np.random.default_rng(11)
    clips = np.concatenate([np.stack([c.astype(np.float64) for c in cases.values()]),
                            rng.uniform(-0.3, 0.3, (40, 16000)),
                            rng.uniform(-0.01, 0.01, (20, 16000)) + cases["scrub_a"].astype(np.float64)])
    labels, trace = dsp.classify_batch_f64(clips, with_trace=True)
    os.environ["DSP_AMD_F64_DFT"] = "1"
    try:
        labels_d, trace_d = dsp.classify_batch_f64(clips, with_trace=True)
    finally:
        del os.environ["DSP_AMD_F64_DFT"]
    assert np.array_equal(labels, labels_d) and labels.any() and not labels.all()
    n_mid = 0
    for (m, s), (md, sd) in zip(trace, trace_d):
        assert m.shape == md.shape and np.array_equal(m, md)
        assert np.allclose(s, sd, rtol=1e-10, atol=1e-10)
        n_mid += len(m)
    assert n_mid >= 25
    # odd lengths (rows of the workspace are padded to 16 bytes) and a single segment
    for n in (256, 479, 5001):
        x = rng.uniform(-0.3, 0.3, (3, n))
        a = dsp.classify_batch_f64(x, with_trace=True)
        os.environ["DSP_AMD_F64_DFT"] = "1"
        try:
            b = dsp.classify_batch_f64(x, with_trace=True)
        finally:
            del os.environ["DSP_AMD_F64_DFT"]
        assert np.array_equal(a[0], b[0])
        for (m, s), (md, sd) in zip(a[1], b[1]):
            assert np.array_equal(m, md) and np.allclose(s, sd, rtol=1e-10, atol=1e-10)


def test_long_clips_many_midpoints_and_run_to_run_identity(dsp):
    """Clips of several seconds (T = 272 and 714 columns): three calls in one clip (the rule fires at the first midpoint or later),
    33 bursts (33 midpoints, the rule never fires: every band window of every midpoint is summed).  The work list of clips with
    midpoints is filled with atomics -- its order varies from run to run, a clip's results must not: three runs, identical bits."""
    import torch
    from oracle import oracle as O
    x = S.classify_cases()["scrub_a"].astype(np.float64)
    rng = np.random.default_rng(3)
    three = np.concatenate([x, rng.uniform(-1, 1, 8000) * 0.001, x[::-1] * 0.7, x, rng.uniform(-1, 1, 5000) * 0.001])
    labels, trace = dsp.classify_batch_f64(three[None, :], with_trace=True)
    assert _check(labels[0], trace[0], three, what="three calls") == (1, 3)
    quiet_first = np.concatenate([x[::-1] * 0.7, rng.uniform(-1, 1, 8000) * 0.001, x])       # the first midpoint misses the rule
    labels, trace = dsp.classify_batch_f64(quiet_first[None, :], with_trace=True)
    _check(labels[0], trace[0], quiet_first, what="reversed call first")
    n = 160000
    bursts = rng.uniform(-1, 1, n) * 0.0005
    for k in range(0, n - 4000, 4800):
        bursts[k:k + 3200] += rng.uniform(-1, 1, 3200) * 0.3
    labels, trace = dsp.classify_batch_f64(bursts[None, :], with_trace=True)
    assert _check(labels[0], trace[0], bursts, what="bursts") == (0, 33)
    # a batch of the long clip among quiet ones, three times
    batch = rng.uniform(-1, 1, (96, n)) * 0.0005
    batch[::3] = bursts
    batch[1::6, :len(three)] += three
    d = torch.from_numpy(batch).cuda()
    runs = []
    for _ in range(3):
        lab = torch.empty(96, dtype=torch.int32, device="cuda")
        runs.append((dsp.classify_device_f64(d, lab).cpu().numpy().copy(), dsp.classify_batch_f64(batch, with_trace=True)))
    for lab, (hl, tr) in runs:
        assert np.array_equal(lab, runs[0][0]) and np.array_equal(hl, runs[0][0])
        for (m, s), (m0, s0) in zip(tr, runs[0][1][1]):
            assert np.array_equal(m, m0) and np.array_equal(s, s0)
    assert runs[0][0][::3].sum() == 0 and runs[0][0][1::6].sum() >= 12 and [len(m) for m, _ in runs[0][1][1][:3]] == [33, 3, 0]
    assert runs[0][0][1::6].tolist() == [O.classify_f64(c)[0] for c in batch[1::6]]        # (a noise floor can make the call miss the rule)


def test_threshold_guard_band_path_decides_like_the_comparison(dsp):
    """A cell's "above the midpoint threshold" is a comparison with U x the threshold's power unless the cell lies within a guard band
    of it (2e-9), where the reference's 10 log10(s / 1e-12) > midpoint_db is evaluated.  DSP_AMD_F64_GUARD (read per call) widens the
    band to +-90 %, so that most cells of these clips go through the exact path: same labels, midpoints, sums, bit for bit."""
    cases = S.classify_cases()
    rng = np.random.default_rng(17)
    clips = np.concatenate([np.stack([c.astype(np.float64) for c in cases.values()]),
                            rng.uniform(-1, 1, (24, 16000)) * np.logspace(-3.5, -1, 24)[:, None]])     # floors around the 45 dB threshold
    labels, trace = dsp.classify_batch_f64(clips, with_trace=True)
    os.environ["DSP_AMD_F64_GUARD"] = "0.9"
    try:
        labels_g, trace_g = dsp.classify_batch_f64(clips, with_trace=True)
    finally:
        del os.environ["DSP_AMD_F64_GUARD"]
    assert np.array_equal(labels, labels_g)
    counts = [len(m) for m, _ in trace]
    assert 0 in counts and max(counts) >= 1 and len(set(counts[len(cases):])) >= 2      # floors below and above the threshold
    for (m, s), (mg, sg) in zip(trace, trace_g):
        assert np.array_equal(m, mg) and np.array_equal(s, sg)
