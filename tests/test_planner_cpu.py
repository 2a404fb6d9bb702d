"""CPU: host-side planner of the wave kernel (dsp_amd/csrc/tables.cpp) through the
C-ABI introspection call -- the sparse mel chunks must reproduce the dense
filterbank exactly and the read windows must be LDS-bank-conflict free."""
import ctypes as C

import numpy as np
import pytest

import dsp_amd
from dsp_amd import lib as dl
from oracle import oracle as O


def _lane_tables(**over):
    L = dsp_amd.load()
    cfg = dsp_amd.default_config(**over)
    t = dl.LaneTables512()
    assert L.dsp_mfcc_lane_tables(C.byref(cfg), None, 0) == C.sizeof(t)
    rc = L.dsp_mfcc_lane_tables(C.byref(cfg), C.byref(t), C.sizeof(t))
    return cfg, t, rc


@pytest.mark.parametrize("over", [dict(), dict(frame_length=512, hop_length=512), dict(n_mels=32), dict(n_mels=24, n_mfcc=8),
                                  dict(fmin=100.0, fmax=6000.0), dict(mel_norm=1), dict(n_mfcc=20)])
def test_mel_chunks_reproduce_dense_filterbank(over):
    cfg, t, rc = _lane_tables(**over)
    assert rc == 0, dl.last_error()
    k0 = np.array(t.mel_k0[:])
    w = np.array([list(t.mel_w[i]) for i in range(12)])
    src = np.array([list(t.mel_src[g]) for g in range(6)])
    mel = O.mel_filterbank(16000, 512, cfg.n_mels, cfg.fmin, cfg.fmax, cfg.mel_norm)
    rec = np.zeros_like(mel)
    used = set()
    for m in range(cfg.n_mels):
        for g in range(6):
            lane = src[g][m]
            if lane == 64:      # zero slot
                continue
            assert g < t.mel_gather and lane not in used
            used.add(lane)
            assert 0 <= k0[lane] <= 257 - 12       # every read stays inside P[0..256]
            for i in range(12):
                rec[m, k0[lane] + i] += w[i][lane]
    assert np.array_equal(rec, mel)
    # lanes that own no chunk carry zero weights
    for lane in set(range(64)) - used:
        assert not w[:, lane].any()
    # ds_read_b32: lanes 0-31 and 32-63 are served separately over 32 banks
    assert t.mel_conflict_free == 1
    for half in (k0[:32], k0[32:]):
        assert len(set(half % 32)) == 32


def test_untangle_partner_map_and_twiddles():
    cfg, t, rc = _lane_tables()
    kap = np.array(t.kappa[:]); partner = np.array(t.partner[:])
    assert sorted(kap) == list(range(64))
    assert np.array_equal(kap[partner], (64 - kap) % 64)
    twp = np.array([list(t.twp[i]) for i in range(4)], np.float64)
    ang = -2 * np.pi * kap / 512
    assert np.allclose(twp[0], np.cos(ang), atol=1e-7) and np.allclose(twp[1], np.sin(ang), atol=1e-7)
    ang = -2 * np.pi * (kap + 64) / 512
    assert np.allclose(twp[2], np.cos(ang), atol=1e-7) and np.allclose(twp[3], np.sin(ang), atol=1e-7)


def test_window_is_prescaled_by_half_and_zero_padded():
    cfg, t, rc = _lane_tables()          # frame 400 inside n_fft 512
    win = np.array([list(t.win[i]) for i in range(8)])
    hann = np.zeros(512, np.float32); hann[:400] = O.window(O.WINDOW_HANN, 400)
    for a in range(4):
        n = np.arange(64) + 64 * a
        assert np.array_equal(win[2 * a], np.float32(0.5) * hann[2 * n])
        assert np.array_equal(win[2 * a + 1], np.float32(0.5) * hann[2 * n + 1])


def test_dct_rows_are_split_over_lanes():
    cfg, t, rc = _lane_tables()
    assert (t.dct_split, t.dct_len) == (4, 10)
    d = O.dct_ortho(13, 40)
    w = np.array([list(t.dct_w[i]) for i in range(20)])
    for c in range(13):
        for q in range(4):
            assert np.array_equal(w[:10, 4 * c + q], d[c, 10 * q: 10 * q + 10])


def test_too_many_chunks_is_an_error_not_a_wrong_answer():
    cfg, t, rc = _lane_tables(n_mels=64)
    assert rc == -1 and "chunks" in dl.last_error()
