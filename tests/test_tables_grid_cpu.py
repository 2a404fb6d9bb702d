"""CPU: every host-side table builder over the randomised-configuration grid of tests/test_gpu_fuzz.py (window, mel filterbank, DCT,
the 512-point kernel's per-lane layout with its bipartite matching and chunk packing) plus the prefilter planner's self-check: the
builders either produce finite tables or refuse with a reason.  The same file runs under AddressSanitizer / UBSan in
tools/asan_host.sh (tests/test_sanitizers_cpu.py), which is where out-of-bounds indexing in a builder would show."""
import ctypes as C

import numpy as np
import pytest

import dsp_amd
from dsp_amd import lib as dl
from tests.test_gpu_fuzz import draw


@pytest.mark.parametrize("n_fft", [512, 1024, 2048])
def test_table_builders_over_the_fuzz_grid(n_fft):
    L = dl.load()
    rng = np.random.default_rng(77 + n_fft)
    built = refused = 0
    for _ in range(40):
        over = draw(rng, n_fft)
        cfg = dsp_amd.default_config(**over)
        try:
            win, mel, dct = dsp_amd.tables(cfg)
        except dl.DspError as exc:
            assert str(exc)
            refused += 1
            continue
        assert np.isfinite(win).all() and np.isfinite(mel).all() and np.isfinite(dct).all()
        assert mel.shape == (over["n_mels"], n_fft // 2 + 1) and dct.shape == (over["n_mfcc"], over["n_mels"]) and (mel >= 0).all()
        built += 1
        if n_fft == 512:
            t = dl.LaneTables512()
            rc = L.dsp_mfcc_lane_tables(C.byref(cfg), C.byref(t), C.sizeof(t))
            assert rc >= 0 or dl.last_error()
    assert built >= 25, (built, refused)


def test_prefilter_planner_self_check_and_degenerate_configs():
    L = dl.load()
    steps = (C.c_int * 4)()
    for pre in (1, 2):
        assert L.dsp_prefilter_scan_check(pre, steps) == 3 and all(1 <= s <= 6 for s in steps)
    assert L.dsp_prefilter_scan_check(0, None) < 0 and L.dsp_prefilter_scan_check(7, steps) < 0
    for over in (dict(n_mels=0), dict(n_mfcc=0), dict(hop_length=0), dict(frame_length=3), dict(frame_length=0), dict(fmin=9000.0), dict(n_fft=333), dict(sample_rate=0)):
        with pytest.raises(dl.DspError):
            dsp_amd.tables(dsp_amd.default_config(**over))
