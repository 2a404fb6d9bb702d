"""The oracle's restatement of the aubio front end of cepstrum/scrubjay_infer.c:21-53 (oracle/aubio_oracle.c).

PARITY UNPINNED: aubio is not in the image and the reference holds no vector at that boundary.  What can be checked here is
checked: an independent float64 numpy / scipy restatement of the same published algorithm (written from the formulas, not from
the C code: clean triangle formula, np.fft, scipy's DCT), the frame count and history of the reference's own loop, and the
properties the chain must have (log10 of a magnitude spectrum: gain shifts only c0; silence hits the SAFE_LOG10 floor)."""
import numpy as np
import pytest
import scipy.fft

from oracle import oracle as O
from tests import signals as S

WIN, HOP, NF, NC = 2048, 1024, 40, 20      # scrubjay_infer.c:9-13


def slaney_edges():
    """Malcolm Slaney's Auditory Toolbox band edges: 13 linear from 133.33 Hz every 66.67 Hz, then 27 + 2 log-spaced (x 1.0711703)."""
    lin = 133.3333 + 66.66666666 * np.arange(13)
    log = lin[-1] * 1.0711703 ** np.arange(1, 30)
    return np.concatenate([lin, log])


def slaney_bank_f64(sr, win_s):
    """Unit-area triangles on the bin frequencies k sr / win_s, evaluated from the triangle formula."""
    e = slaney_edges()
    f = np.arange(win_s // 2 + 1) * sr / win_s
    fb = np.zeros((NF, f.size))
    for m in range(NF):
        lo, ce, hi = e[m], e[m + 1], e[m + 2]
        h = 2.0 / (hi - lo)
        up = (f - lo) / (ce - lo) * h
        dn = (hi - f) / (hi - ce) * h
        fb[m] = np.clip(np.minimum(up, dn), 0.0, None)
    fb[:, -1] = 0.0          # aubio's loops stop before the Nyquist bin
    return fb


def aubio_mfcc_f64(x, sr=16000):
    x = np.asarray(x, np.float64)
    T = -(-x.size // HOP)
    pad = np.concatenate([np.zeros(WIN - HOP), x, np.zeros(T * HOP - x.size)])
    w = 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(WIN) / WIN))
    fb = slaney_bank_f64(sr, WIN)
    out = np.empty((T, NC))
    for t in range(T):
        fr = pad[t * HOP: t * HOP + WIN] * w
        mag = np.abs(np.fft.rfft(np.roll(fr, WIN // 2)))           # fvec_shift: zero phase; |.| does not see it
        e = fb @ mag
        out[t] = scipy.fft.dct(np.log10(np.maximum(e, 2e-42)), type=2, norm="ortho")[:NC]
    return out


def test_frame_count_is_the_do_while_of_scrubjay_infer():
    # scrubjay_infer.c:39-53: one frame per aubio_source_do that returned samples, loop ends after the first short read
    for n, t in ((0, 0), (1, 1), (1023, 1), (1024, 1), (1025, 2), (16000, 16), (16384, 16), (16385, 17)):
        assert O.aubio_frames_for(n, HOP) == t
    assert O.aubio_mfcc_clip(S.uniform_pm1(16000, 1)).shape == (16, 20)


def test_filterbank_against_the_triangle_formula():
    for sr in (16000, 96000):
        fb = O.aubio_filterbank_slaney(sr, WIN).astype(np.float64)
        ref = slaney_bank_f64(sr, WIN)
        assert fb.shape == ref.shape
        # float32 construction, boundary bins decided by float compares: within 1e-5 of the largest weight everywhere
        assert np.abs(fb - ref).max() <= 1e-5 * ref.max()
        if sr == 16000:        # unit area: the wide upper filters sample their triangle finely, the 17-bin lower ones coarsely
            area = fb.sum(1) * sr / WIN
            assert np.all(np.abs(area[20:] - 1.0) < 1e-2) and np.all(np.abs(area - 1.0) < 0.15)
    e = slaney_edges()
    assert abs(e[13] - 999.78) < 0.05 and abs(e[-1] - 6853.8) < 2.0     # the bank tops out near 6.85 kHz


def test_window_is_the_periodic_hann():
    w = O.aubio_window_hanningz(WIN)
    assert np.abs(w - (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(WIN) / WIN))).max() < 2e-7
    assert w[0] == 0.0


@pytest.mark.parametrize("name", ["noise", "chirp", "tone", "short", "quiet"])
def test_chain_against_the_float64_restatement(name):
    x = {"noise": S.uniform_pm1(16000, 3), "chirp": S.chirp(16000, 200.0, 6000.0) + np.float32(1e-3) * S.uniform_pm1(16000, 9), "tone": (0.3 * np.sin(2 * np.pi * 1000.0 * np.arange(16000) / 16000.0)).astype(np.float32) + np.float32(1e-3) * S.uniform_pm1(16000, 10),
         "short": S.uniform_pm1(2500, 4), "quiet": (S.uniform_pm1(9000, 5) * np.float32(1e-3))}[name]
    got = O.aubio_mfcc_clip(x)
    ref = aubio_mfcc_f64(x)
    assert got.shape == ref.shape
    # float32 window / filterbank / sums against float64: log10 compresses, so errors sit near 1e-6 of c0.  (The chirp rides on
    # noise 54 dB down: this chain takes the log of every filter WITHOUT a floor relative to the frame's peak, so for a clean
    # sweep the filters far from it hold only window leakage below float32's rounding noise and no float32 evaluation has a
    # defined value there.)
    assert np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max(axis=-1, keepdims=True).max()


def test_history_and_partial_hop():
    x = S.uniform_pm1(3000, 6)
    m = O.aubio_mfcc_clip(x)
    # frame 0 = [1024 zeros | first hop]; the last frame's new hop is zero padded past the end of the file
    assert m.shape[0] == 3
    x_pad = np.concatenate([x, np.zeros(3 * HOP - x.size, np.float32)])
    assert np.array_equal(O.aubio_mfcc_clip(x_pad), m)
    head = O.aubio_mfcc_clip(x[:HOP])
    assert np.array_equal(head[0], m[0])                        # frame 0 sees nothing after its own hop


def test_gain_moves_only_c0_and_silence_hits_the_floor():
    x = S.uniform_pm1(8000, 7)
    a, b = O.aubio_mfcc_clip(x), O.aubio_mfcc_clip(x * np.float32(0.25))
    assert np.abs((a - b)[:, 1:]).max() < 2e-4                  # log10(g |X|) = log10 g + log10 |X|: constant over the filters
    assert np.abs((a - b)[:, 0] - np.sqrt(40.0) * np.log10(4.0)).max() < 2e-4
    z = O.aubio_mfcc_clip(np.zeros(4096, np.float32))
    floor = np.log10(np.float64(np.float32(2e-42)))
    assert np.abs(z[:, 0] - np.sqrt(40.0) * floor).max() < 1e-3 and np.abs(z[:, 1:]).max() < 1e-4


def test_pooling_is_scrubjay_infers():
    m = O.aubio_mfcc_clip(S.uniform_pm1(16000, 8))
    f = O.mfcc_stats(m)
    assert np.allclose(f[:20], m.astype(np.float64).mean(0), atol=1e-6)
    assert np.allclose(f[20:], m.astype(np.float64).std(0), atol=1e-5)
