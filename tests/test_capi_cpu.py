"""CPU: the C-ABI library loads, exports every symbol include/dsp_amd.h declares,
and its host-only logic (config defaults, frame counting, tables, error paths)
matches the reference.  No compute call is made here (no GPU in this tier)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import dsp_amd
from dsp_amd import lib as dl
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "dsp_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(compute_mfcc|fft_real_forward|classify\w*|dsp_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    L = dsp_amd.load()
    names = _declared_symbols()
    assert "compute_mfcc" in names and len(names) >= 12
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dsp_amd.h but not exported"
    assert sorted(dl.SYMBOLS) == names


def test_default_config_is_the_reference():
    c = dsp_amd.default_config()
    assert (c.sample_rate, c.n_fft, c.frame_length, c.hop_length, c.n_mels, c.n_mfcc) == (16000, 512, 400, 160, 40, 13)
    assert c.amin == np.float32(1e-10) and c.top_db == 80.0 and c.fmin == 0.0 and c.fmax == 8000.0


@pytest.mark.parametrize("n,mx,want", [(399, 500, 0), (400, 500, 1), (559, 500, 1), (560, 500, 2), (16000, 500, 98),
                                       (24029, 500, 148), (96000, 500, 500), (16000, 0, 0), (16000, -3, 0), (16000, 7, 7)])
def test_frame_count_rule(n, mx, want):
    assert dsp_amd.frames_for(dsp_amd.default_config(), n, mx) == want


def test_tables_equal_the_oracle_tables():
    cfg = dsp_amd.default_config()
    w, m, d = dsp_amd.tables(cfg)
    assert np.array_equal(w, O.window(O.WINDOW_HANN, 400))
    assert np.array_equal(m, O.mel_filterbank())
    assert np.array_equal(d, O.dct_ortho())
    cfg2 = dsp_amd.default_config(n_mels=32, n_mfcc=20, mel_norm=dl.MELNORM_SLANEY, window=dl.WINDOW_HAMMING, fmin=50.0, fmax=7000.0)
    w, m, d = dsp_amd.tables(cfg2)
    assert np.array_equal(w, O.window(O.WINDOW_HAMMING, 400))
    assert np.array_equal(m, O.mel_filterbank(16000, 512, 32, 50.0, 7000.0, O.MELNORM_SLANEY))
    assert np.array_equal(d, O.dct_ortho(20, 32))


def test_butter_bandpass_tables_and_rejection():
    L = dsp_amd.load()
    b = (C.c_double * 9)(); a = (C.c_double * 9)()
    for lo, hi in ((1000, 3000), (3000, 7500)):
        assert L.dsp_butter_bandpass(lo, hi, b, a) == 1
        ok, ob, oa = O.butter_bandpass(lo, hi)
        assert list(b) == list(ob) and list(a) == list(oa)
    assert L.dsp_butter_bandpass(2000, 6000, b, a) == 0      # classifier.c:402-407
    assert "invalid bandpass range" in dl.last_error()


def test_config3_prefilter_planner_checks_itself_for_both_literal_filters():
    """The fused prefilter of BASELINE config 3 (cascade of lane scans, its scan in row form) is checked by the planner against the
    direct-form recurrence at plan creation; the same check is callable without a GPU."""
    L = dsp_amd.load()
    steps = (C.c_int * 4)()
    assert L.dsp_prefilter_scan_check(1, steps) == 3 and list(steps) == [2, 3, 3, 4]          # 1000-3000 Hz
    assert L.dsp_prefilter_scan_check(2, steps) == 3 and list(steps) == [1, 3, 3, 5]          # 3000-7500 Hz (config 3)
    assert L.dsp_prefilter_scan_check(0, None) < 0 and "literal band-passes" in dl.last_error()


def test_bad_configs_are_rejected_before_touching_a_device():
    L = dsp_amd.load()
    h = C.c_void_p()
    for over, msg in ((dict(n_fft=256), "n_fft"), (dict(n_fft=1024, n_mels=200), "n_mels"), (dict(frame_length=401), "even"), (dict(hop_length=0), "hop"),
                      (dict(n_mels=100), "n_mels"), (dict(log_mode=7), "log_mode"), (dict(win_length=999), "win_length")):
        cfg = dsp_amd.default_config(**over)
        assert L.dsp_mfcc_plan_create(C.byref(cfg), 0, C.byref(h)) == -1
        assert msg.lower() in dl.last_error().lower()


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dsp_amd.DspError, match="no HIP device"):
        dsp_amd.MfccPlan()
    # the reference entry point keeps its return convention: 0 frames, reason via dsp_last_error()
    out = dsp_amd.compute_mfcc(np.zeros(16000, np.float32), 500)
    assert out.shape[0] == 0 and "no HIP device" in dl.last_error()


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "dsp_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f"{f} mentions the oracle"


def _stop_params(golden_dir, **over):
    m = dict(np.load(os.path.join(golden_dir, "stop_model.npz")))
    m.update(over)
    return m


def test_consumer_models_reject_bad_parameters_and_fail_loudly_without_a_gpu():
    import torch
    gd = os.path.join(ROOT, "tests", "golden")
    m = _stop_params(gd)
    with pytest.raises(dsp_amd.DspError, match="n_coef"):                     # scaler length does not match the shape
        dsp_amd.StopModel(dict(m, max_frames=400))
    with pytest.raises(dsp_amd.DspError, match="kernel1"):
        dsp_amd.StopModel(dict(m, kernel1=np.zeros(7, np.float32)))
    wide = dict(m, bias0=np.zeros(17, np.float32), kernel0=np.zeros(6500 * 17, np.float32), kernel1=np.zeros(17 * 2, np.float32))
    with pytest.raises(dsp_amd.DspError, match="1..16 units"):                # the C ABI's own check
        dsp_amd.StopModel(wide)
    s = np.load(os.path.join(gd, "speaker_gmm_ref.npz"))
    t = {k: s[f"target_{k}"] for k in ("means", "inv_covs", "log_consts")}
    u = {k: s[f"ubm_{k}"] for k in ("means", "inv_covs", "log_consts")}
    with pytest.raises(dsp_amd.DspError, match="same shape"):
        dsp_amd.SpeakerModel(t, dict(u, means=u["means"][:16], inv_covs=u["inv_covs"][:16], log_consts=u["log_consts"][:16]))
    with pytest.raises(dsp_amd.DspError, match="new_size"):
        dsp_amd.upsample_linear(np.zeros(8, np.float32), 1)
    if not torch.cuda.is_available():
        for make in (lambda: dsp_amd.StopModel(m), lambda: dsp_amd.SpeakerModel(t, u),
                     lambda: dsp_amd.upsample_linear(np.zeros(8, np.float32), 16)):
            with pytest.raises(dsp_amd.DspError, match="no HIP device"):
                make()


def test_classifier_contexts_are_per_device_locks():
    """The classifier keeps one context (tables, workspace, mutex) per device: two threads that drive two GPUs from one process must
    not queue on one lock, two threads on one device must.  dsp_debug_hold_classify_ctx holds a device's default context for a given
    time without touching the GPU."""
    import threading
    import time
    L = dl.load()
    hold = 300

    def run(devices):
        ts = [threading.Thread(target=lambda d=d: dl.check(L.dsp_debug_hold_classify_ctx(d, hold), "hold")) for d in devices]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return (time.perf_counter() - t0) * 1e3

    two = run([0, 1])
    one = run([3, 3])
    assert two < 1.6 * hold, two            # overlapped
    assert one > 1.9 * hold, one            # queued
    assert L.dsp_debug_hold_classify_ctx(64, 1) < 0 and L.dsp_debug_hold_classify_ctx(-1, 1) < 0


def test_gather_entry_points_reject_bad_arguments_and_fail_loudly_without_a_gpu():
    L = dl.load()
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 1)
    assert L.dsp_gather_create(None, 1, C.byref(h)) < 0 and L.dsp_gather_create(devs, 0, C.byref(h)) < 0 and L.dsp_gather_create(devs, 65, C.byref(h)) < 0
    assert L.dsp_gather_create(devs, 2, None) < 0
    if L.dsp_device_count() <= 0:
        assert L.dsp_gather_create(devs, 2, C.byref(h)) == -2 and not h.value        # DSP_ENODEV
    assert L.dsp_gather_all(None, None, None, 4, None) < 0 and L.dsp_gather_n_devices(None) < 0
    L.dsp_gather_destroy(None)


def test_ragged_batches_are_ordered_for_an_even_deal():
    """dsp_debug_fused_spans (host only): every clip once, with the frames its own length gives, and -- wave w walking positions
    w, w + n_waves, ... -- every wave's frame total within one clip of the mean; clips without a frame and offsets that run backwards
    are refused by name."""
    import ctypes as C
    from dsp_amd import lib as L
    lib = L.load()
    cfg = L.MfccConfig()
    lib.dsp_mfcc_default_config(C.byref(cfg))
    rng = np.random.default_rng(11)
    for n_clips, n_waves in ((1, 4), (7, 4), (1000, 64), (5000, 4096), (4097, 4096)):
        lens = rng.integers(400, 48000, n_clips)
        off = np.zeros(n_clips + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)
        c_off, _ = L.c_offsets(off)
        out = (C.c_long * (4 * n_clips))()
        tm = lib.dsp_debug_fused_spans(C.byref(cfg), c_off, n_clips, 500, n_waves, out)
        spans = np.array(out[:], dtype=np.int64).reshape(n_clips, 4)
        frames = np.minimum(1 + (lens - 400) // 160, 500)
        assert tm == frames.max()
        assert sorted(spans[:, 3].tolist()) == list(range(n_clips))
        assert np.array_equal(spans[:, 0], off[spans[:, 3]]) and np.array_equal(spans[:, 1], lens[spans[:, 3]]) and np.array_equal(spans[:, 2], frames[spans[:, 3]])
        per_wave = np.array([spans[w::n_waves, 2].sum() for w in range(min(n_waves, n_clips))])
        if n_clips >= 2 * n_waves:
            assert per_wave.max() - per_wave.min() <= frames.max(), (n_clips, n_waves, per_wave.max(), per_wave.min())
    bad = np.array([0, 16000, 16399], dtype=np.int64)
    c_bad, _ = L.c_offsets(bad)
    out = (C.c_long * 8)()
    assert lib.dsp_debug_fused_spans(C.byref(cfg), c_bad, 2, 500, 4, out) == -1 and b"clip 1" in lib.dsp_last_error()
    back = np.array([0, 16000, 8000], dtype=np.int64)
    c_back, _ = L.c_offsets(back)
    assert lib.dsp_debug_fused_spans(C.byref(cfg), c_back, 2, 500, 4, out) == -1 and b"non-decreasing" in lib.dsp_last_error()
    # the classifiers' ragged entry points refuse bad arguments before any GPU call
    lab = (C.c_int * 2)()
    assert lib.dsp_classify_batch_ragged_host(None, None, 2, c_back, lab, None) == -1
    assert lib.dsp_classify_batch_ragged_host_f64(None, None, 2, c_back, lab, None) == -1
