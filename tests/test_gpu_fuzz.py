"""Randomised-configuration parity (GPU): seeded draws of the MFCC configuration space -- n_fft 512 / 1024 / 2048, frame and hop
lengths, mel / coefficient counts, sample rates, band edges, window kinds, mel normalisation -- each compared with the oracle
through the C-ABI on a few short clips (noise, a chirp, a quiet clip with a silent tail).  What a fixed list of shapes misses:
table-builder corner cases (narrow / wide mel filters, segment and chunk limits and their fallbacks, short frames in a long
transform, odd clip tails).  Configurations a plan refuses (DSP_EINVAL with a reason, e.g. a filter wider than the sparse
tables hold) are counted, not failed -- but most draws must run.  The chirp stays inside [fmin, fmax]: a tone above the bank's
upper edge leaves only window leakage below float32's own FFT noise in every filter (-140 dB), and then the per-frame-max
normalised log-mels of ANY float32 implementation -- the reference's included -- are rounding noise (found by this test's first
version: relative errors of 0.1 .. 1 in exactly those frames, none elsewhere)."""
import numpy as np
import pytest

from tests import signals as S

pytestmark = pytest.mark.gpu

LIMITS = {512: dict(n_mels=(8, 64), n_mfcc=(1, 32)), 1024: dict(n_mels=(8, 128), n_mfcc=(1, 16)), 2048: dict(n_mels=(8, 128), n_mfcc=(1, 32))}


def draw(rng, n_fft):
    lim = LIMITS[n_fft]
    sr = int(rng.choice([8000, 16000, 22050, 44100]))
    frame = int(rng.integers(n_fft // 4, n_fft // 2 + 1)) * 2                # even, in [n_fft / 2, n_fft]
    if rng.random() < 0.3:
        frame = n_fft
    hop = int(rng.integers(8, frame // 2 + 1)) * 2                           # even, <= frame
    n_mels = int(rng.integers(lim["n_mels"][0], lim["n_mels"][1] + 1))
    n_mfcc = int(rng.integers(lim["n_mfcc"][0], min(lim["n_mfcc"][1], n_mels) + 1))
    fmin = float(rng.choice([0.0, 0.0, 20.0, 300.0]))
    fmax = float(rng.choice([sr / 2.0, sr / 2.0, 0.45 * sr, 0.3 * sr]))
    return dict(sample_rate=sr, n_fft=n_fft, frame_length=frame, hop_length=hop, n_mels=n_mels, n_mfcc=n_mfcc, fmin=fmin, fmax=fmax,
                window=int(rng.integers(0, 3)), mel_norm=int(rng.integers(0, 3)),
                log_mode=int(rng.integers(0, 2)) if n_fft != 1024 else 0)       # clip-global log mode: 512 and 2048


@pytest.mark.parametrize("n_fft", [512, 1024, 2048])
def test_random_configurations_match_the_oracle(n_fft):
    import torch
    import dsp_amd
    from dsp_amd import lib as L
    from oracle import oracle as O
    from tests.conftest import gate
    rng = np.random.default_rng(20261004 + n_fft)
    ran = refused = 0
    for it in range(14):
        over = draw(rng, n_fft)
        try:
            plan = dsp_amd.MfccPlan(dsp_amd.default_config(**over))
        except L.DspError as exc:                            # the plan says why (table limits)
            assert str(exc), "a refusal carries its reason"
            refused += 1
            continue
        ocfg = O.default_cfg(**over)
        n = over["frame_length"] + over["hop_length"] * int(rng.integers(3, 40)) + int(rng.integers(0, over["hop_length"]))
        n += n & 1                                                           # the clip stride must be even (8-byte aligned frame loads)
        x = np.stack([S.uniform_pm1(n, 1000 + it), S.chirp(n, over["fmin"] + 100.0, 0.9 * over["fmax"], fs=float(over["sample_rate"])).astype(np.float32),
                      S.uniform_pm1(n, 2000 + it) * np.float32(0.003)])
        x[2, n // 2:] = 0.0
        out = plan.clips(torch.from_numpy(x).cuda(), 1000).cpu().numpy()
        for i in range(3):
            ref = O.compute_mfcc(x[i], 1000, ocfg)
            assert out[i].shape == ref.shape, (over, out[i].shape, ref.shape)
            gate(out[i], ref, f"fuzz n_fft {n_fft} #{it} clip{i} {over}")
        ran += 1
    assert ran >= 7, f"only {ran} of {ran + refused} random configurations ran"


def test_random_clip_lengths_through_classify_match_the_oracle():
    """classify() on batches of random length (odd lengths, lengths that are no multiple of the 32-sample tiles or the 224-sample
    spectrogram hop, 0.1 s .. 2.4 s) built from shifted pieces of the call-like patterns over noise: labels, midpoints and band
    sums against the oracle, bit exact."""
    import dsp_amd
    from oracle import oracle as O
    rng = np.random.default_rng(4104)
    c = S.classify_cases()
    calls = [c["scrub_a"], c["scrub_b"], c["jay_like"], c["burst_2k"]]
    with_mids = hits = 0
    for it in range(12):
        n = int(rng.integers(1600, 38400))
        clips = np.empty((6, n), np.float32)
        for i in range(6):
            x = rng.uniform(-1, 1, n) * 10.0 ** rng.uniform(-3.0, -1.0)
            call = np.tile(calls[int(rng.integers(0, 4))], 3)[int(rng.integers(0, 16000)):][:n].astype(np.float64)
            if i % 3 != 2:
                x[:call.size] += call * 10.0 ** rng.uniform(-1.0, 0.2)
            clips[i] = x.astype(np.float32)
        labels, trace = dsp_amd.classify_batch(clips, with_trace=True)
        for x, lab, (mids, sums) in zip(clips, labels, trace):
            olab, _, osums = O.classify(x)
            omids = O.find_midpoints(x)
            assert lab == olab, (it, n)
            assert np.array_equal(mids, omids), (it, n)
            assert np.array_equal(sums, osums), (it, n)
            with_mids += len(omids) > 0
            hits += int(olab)
    assert with_mids >= 12 and hits >= 3, (with_mids, hits)      # the draws exercise the midpoint and the rule paths
