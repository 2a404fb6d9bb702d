"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present,
the reference's own compiled sources (oracle/_ref/*.so).

TEST INFRASTRUCTURE ONLY.  Importable only from tests/, bench.py's cpu_baseline
leg and __graft_entry__.smoke(); the product package dsp_amd never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
_D = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class MfccCfg(C.Structure):
    """Mirror of orc_mfcc_cfg (oracle/dsp_oracle.h)."""

    _fields_ = [
        ("sample_rate", C.c_int), ("n_fft", C.c_int), ("frame_length", C.c_int),
        ("hop_length", C.c_int), ("n_mels", C.c_int), ("n_mfcc", C.c_int),
        ("window", C.c_int), ("mel_norm", C.c_int), ("log_mode", C.c_int),
        ("fft_mode", C.c_int), ("prefilter", C.c_int), ("win_length", C.c_int),
        ("fmin", C.c_float), ("fmax", C.c_float), ("amin", C.c_float),
        ("top_db", C.c_float),
    ]


class ClassifyTrace(C.Structure):
    _fields_ = [("n_midpoints", C.c_int), ("midpoints", C.c_float * 64),
                ("sums", (C.c_float * 3) * 64)]


class ClassifyCfgF64(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("keep_lo", "keep_hi", "midpoint_db", "middle_max", "above_min", "below_min")]


class ClassifyTraceF64(C.Structure):
    _fields_ = [("n_midpoints", C.c_int), ("midpoints", C.c_double * 64), ("sums", (C.c_double * 3) * 64)]


class SvmModel(C.Structure):
    _fields_ = [("n_features", C.c_int), ("n_sv", C.c_int), ("gamma", C.c_float),
                ("rho", C.c_float), ("prob_a", C.c_float), ("prob_b", C.c_float),
                ("offset", C.POINTER(C.c_float)), ("scale", C.POINTER(C.c_float)),
                ("sv", C.POINTER(C.c_float)), ("coef", C.POINTER(C.c_float))]


class StopModel(C.Structure):
    _fields_ = [("n_coef", C.c_int), ("max_frames", C.c_int), ("units", C.c_int * 4),
                ("scaler_mean", C.POINTER(C.c_float)), ("scaler_scale", C.POINTER(C.c_float)),
                ("kernel", C.POINTER(C.c_float) * 4), ("bias", C.POINTER(C.c_float) * 4)]


class Gmm(C.Structure):
    _fields_ = [("k", C.c_int), ("d", C.c_int), ("means", C.POINTER(C.c_int8)),
                ("inv_covs", C.POINTER(C.c_int32)), ("log_consts", C.POINTER(C.c_int16))]


WINDOW_HANN, WINDOW_HAMMING, WINDOW_RECT = 0, 1, 2
MELNORM_NONE, MELNORM_SLANEY, MELNORM_LIBROSA = 0, 1, 2
LOG_PER_FRAME_MAX, LOG_GLOBAL_REF1 = 0, 1
FFT_REFERENCE_ORDER, FFT_FLOAT64 = 0, 1
PREFILTER_NONE, PREFILTER_BUTTER_1000_3000, PREFILTER_BUTTER_3000_7500 = 0, 1, 2


def build(force: bool = False) -> None:
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("dsp_oracle.c", "aubio_oracle.c", "classify_f64_oracle.c", "dsp_oracle.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(so) < os.path.getmtime(f) for f in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.isdir(os.environ.get("DSP_REF", "/root/reference")):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        # DSP_ORACLE_LIB: another build of the same sources (tools/asan_host.sh: oracle/liboracle_asan.so under AddressSanitizer / UBSan)
        L = C.CDLL(os.environ.get("DSP_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so"))
        L.orc_mfcc_default_cfg.argtypes = [C.POINTER(MfccCfg)]
        L.orc_window.argtypes = [C.c_int, C.c_int, _F]
        L.orc_mel_filterbank.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, _F]
        L.orc_dct_ortho.argtypes = [C.c_int, C.c_int, _F]
        L.orc_fft_real_forward.argtypes = [_F, C.c_int, C.c_int, C.c_int, _F]
        L.orc_compute_mfcc.argtypes = [C.POINTER(MfccCfg), _F, C.c_int, _F, C.c_int]
        L.orc_compute_mfcc.restype = C.c_int
        L.orc_mfcc_frames.argtypes = [C.POINTER(MfccCfg), _F, C.c_long, _F]
        L.orc_mfcc_frames_mt.argtypes = [C.POINTER(MfccCfg), _F, C.c_long, _F, C.c_int]
        L.orc_butter_bandpass.argtypes = [C.c_double, C.c_double, _D, _D]
        L.orc_butter_bandpass.restype = C.c_int
        L.orc_iir_df2_f64.argtypes = [_D, C.c_int, _D, _D, _D]
        L.orc_iir_df2_f32.argtypes = [_F, C.c_int, _F, _F, _F]
        L.orc_spectrogram_bins.argtypes = [C.c_int]
        L.orc_spectrogram_bins.restype = C.c_int
        L.orc_spectrogram_f32.argtypes = [_F, C.c_int, C.c_int, _F, _F, _F]
        L.orc_spectrogram_f32.restype = C.c_int
        L.orc_spectrogram_f64.argtypes = [_D, C.c_int, C.c_int, _D, _D, _D]
        L.orc_spectrogram_f64.restype = C.c_int
        L.orc_sum_intense.argtypes = [C.c_float, C.c_float, C.c_float, _F, C.c_int, _F, C.c_int, _F, C.c_float]
        L.orc_sum_intense.restype = C.c_float
        L.orc_find_midpoints.argtypes = [_F, C.c_int, C.c_int, _F, C.c_int]
        L.orc_find_midpoints.restype = C.c_int
        L.orc_classify.argtypes = [_F, C.c_int, C.POINTER(ClassifyTrace)]
        L.orc_classify.restype = C.c_int
        L.orc_mfcc_stats.argtypes = [_F, C.c_int, C.c_int, _F]
        L.orc_svm_predict.argtypes = [C.POINTER(SvmModel), _F, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_svm_predict.restype = C.c_int
        L.orc_stop_features.argtypes = [C.POINTER(StopModel), _F, C.c_int, _F]
        L.orc_stop_predict.argtypes = [C.POINTER(StopModel), _F]
        L.orc_stop_predict.restype = C.c_float
        L.orc_classify_signal.argtypes = [C.POINTER(StopModel), _F, C.c_int]
        L.orc_classify_signal.restype = C.c_float
        _I16 = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
        L.orc_gmm_log_likelihood.argtypes = [C.POINTER(Gmm), _I16]
        L.orc_gmm_log_likelihood.restype = C.c_int64
        L.orc_float_to_q6.argtypes = [_F, _I16, C.c_int]
        L.orc_speaker_llr_mean.argtypes = [C.POINTER(Gmm), C.POINTER(Gmm), _F, C.c_int]
        L.orc_speaker_llr_mean.restype = C.c_int64
        L.orc_classify_speaker.argtypes = [C.POINTER(Gmm), C.POINTER(Gmm), _F, C.c_int]
        L.orc_classify_speaker.restype = C.c_int
        L.orc_upsample_linear.argtypes = [_F, C.c_int, _F, C.c_int]
        L.orc_classify_f64.argtypes = [_D, C.c_int, C.POINTER(ClassifyCfgF64), C.POINTER(ClassifyTraceF64)]
        L.orc_classify_f64.restype = C.c_int
        L.orc_find_midpoints_f64.argtypes = [_D, C.c_int, C.c_int, C.c_double, _D, C.c_int]
        L.orc_find_midpoints_f64.restype = C.c_int
        L.orc_aubio_window_hanningz.argtypes = [C.c_int, _F]
        L.orc_aubio_filterbank_slaney.argtypes = [C.c_int, C.c_int, _F]
        L.orc_aubio_frames_for.argtypes = [C.c_int, C.c_int]
        L.orc_aubio_frames_for.restype = C.c_int
        L.orc_aubio_mfcc_clip.argtypes = [_F, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _F]
        L.orc_aubio_mfcc_clip.restype = C.c_int
        _lib = L
    return _lib


def default_cfg(**over) -> MfccCfg:
    cfg = MfccCfg()
    lib().orc_mfcc_default_cfg(C.byref(cfg))
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


# ---- tables ---------------------------------------------------------------

def window(kind: int, n: int) -> np.ndarray:
    out = np.empty(n, np.float32)
    lib().orc_window(kind, n, out)
    return out


def mel_filterbank(sr=16000, n_fft=512, n_mels=40, fmin=0.0, fmax=8000.0, norm=MELNORM_NONE) -> np.ndarray:
    out = np.empty((n_mels, n_fft // 2 + 1), np.float32)
    lib().orc_mel_filterbank(sr, n_fft, n_mels, fmin, fmax, norm, out)
    return out


def dct_ortho(n_mfcc=13, n_mels=40) -> np.ndarray:
    out = np.empty((n_mfcc, n_mels), np.float32)
    lib().orc_dct_ortho(n_mfcc, n_mels, out)
    return out


# ---- MFCC -------------------------------------------------------------------

def fft_real_forward(x: np.ndarray, n_fft=512, fft_mode=FFT_REFERENCE_ORDER) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(2 * n_fft, np.float32)
    lib().orc_fft_real_forward(x, x.size, n_fft, fft_mode, out)
    return out


def compute_mfcc(signal: np.ndarray, max_frames: int, cfg: MfccCfg | None = None) -> np.ndarray:
    """orc_compute_mfcc: returns [T][n_mfcc] (T may be 0)."""
    cfg = cfg or default_cfg()
    signal = np.ascontiguousarray(signal, np.float32)
    out = np.zeros((max(max_frames, 0), cfg.n_mfcc), np.float32)
    t = lib().orc_compute_mfcc(C.byref(cfg), signal, signal.size, out.reshape(-1) if out.size else np.zeros(1, np.float32), max_frames)
    return out[:t]


def mfcc_frames(frames: np.ndarray, cfg: MfccCfg, threads: int = 1) -> np.ndarray:
    frames = np.ascontiguousarray(frames, np.float32)
    assert frames.ndim == 2 and frames.shape[1] == cfg.frame_length
    out = np.empty((frames.shape[0], cfg.n_mfcc), np.float32)
    if frames.shape[0] == 0:
        return out
    if threads > 1:
        lib().orc_mfcc_frames_mt(C.byref(cfg), frames.reshape(-1), frames.shape[0], out.reshape(-1), threads)
    else:
        lib().orc_mfcc_frames(C.byref(cfg), frames.reshape(-1), frames.shape[0], out.reshape(-1))
    return out


# ---- Butterworth ------------------------------------------------------------

def butter_bandpass(lo: float, hi: float):
    b = np.zeros(9, np.float64)
    a = np.zeros(9, np.float64)
    ok = lib().orc_butter_bandpass(lo, hi, b, a)
    return bool(ok), b, a


def iir_f64(x, b, a) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float64)
    y = np.empty_like(x)
    lib().orc_iir_df2_f64(x, x.size, np.ascontiguousarray(b, np.float64), np.ascontiguousarray(a, np.float64), y)
    return y


def iir_f32(x, b, a) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().orc_iir_df2_f32(x, x.size, np.ascontiguousarray(b, np.float32), np.ascontiguousarray(a, np.float32), y)
    return y


# ---- spectrogram / classify ---------------------------------------------------

def spectrogram_f32(signal, fs=16000):
    signal = np.ascontiguousarray(signal, np.float32)
    t = lib().orc_spectrogram_bins(signal.size)
    freqs = np.empty(129, np.float32)
    times = np.empty(max(t, 1), np.float32)
    sxx = np.empty((129, max(t, 1)), np.float32)
    lib().orc_spectrogram_f32(signal, signal.size, fs, freqs, times, sxx.reshape(-1))
    return freqs, times[:t], sxx[:, :t]


def spectrogram_f64(signal, fs=16000):
    signal = np.ascontiguousarray(signal, np.float64)
    t = lib().orc_spectrogram_bins(signal.size)
    freqs = np.empty(129, np.float64)
    times = np.empty(max(t, 1), np.float64)
    sxx = np.empty((129, max(t, 1)), np.float64)
    lib().orc_spectrogram_f64(signal, signal.size, fs, freqs, times, sxx.reshape(-1))
    return freqs, times[:t], sxx[:, :t]


def sum_intense(lower, upper, half_range, freqs, times, db, midpoint) -> float:
    db = np.ascontiguousarray(db, np.float32)
    return float(lib().orc_sum_intense(lower, upper, half_range, np.ascontiguousarray(freqs, np.float32), db.shape[0],
                                       np.ascontiguousarray(times, np.float32), db.shape[1], db.reshape(-1), midpoint))


def find_midpoints(data, fs=16000) -> np.ndarray:
    data = np.ascontiguousarray(data, np.float32)
    out = np.zeros(64, np.float32)
    n = lib().orc_find_midpoints(data, data.size, fs, out, 64)
    return out[:n]


class ClassifyCfg(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("keep_lo", "keep_hi", "midpoint_db", "middle_max", "above_min", "below_min")]


# thresholds of the reference's variants (keep band, midpoint dB, rule middle < / above > / below >)
CLASSIFY_SYNC_LIB = (0.65, 0.80, 70.0, 100.0, 200.0, 80.0)        # sync/lib/classifier.cpp:67-68, :436, :109
CLASSIFY_MICROPHONE = (0.70, 0.85, 45.0, 100.0, 200.0, 150.0)     # microphone/src/classifier.cpp:79-80, :448, :123
CLASSIFY_MICROPHONE_C = (0.70, 0.85, 45.0, 50.0, 200.0, 200.0)    # microphone/src/classifier.c:120-121, :608, :164 (thresholds of the float64 file)
CLASSIFY_DONUT_C = (0.70, 0.85, 45.0, 75.0, 300.0, 100.0)         # donut-classifier/classifier.c:141-142, :660, :184 (thresholds of the float64 file)


def classify(data, cfg=None):
    data = np.ascontiguousarray(data, np.float32)
    tr = ClassifyTrace()
    if cfg is None:
        label = lib().orc_classify(data, data.size, C.byref(tr))
    else:
        L = lib()
        L.orc_classify_with.argtypes = [_F, C.c_int, C.POINTER(ClassifyCfg), C.POINTER(ClassifyTrace)]
        L.orc_classify_with.restype = C.c_int
        c = ClassifyCfg(*[float(v) for v in cfg])
        label = L.orc_classify_with(data, data.size, C.byref(c), C.byref(tr))
    mids = np.array(tr.midpoints[: tr.n_midpoints], np.float32)
    sums = np.array([[tr.sums[i][j] for j in range(3)] for i in range(tr.n_midpoints)], np.float32).reshape(-1, 3)
    return int(label), mids, sums


# ---- pooling + SVM ------------------------------------------------------------

def mfcc_stats(mfcc: np.ndarray) -> np.ndarray:
    mfcc = np.ascontiguousarray(mfcc, np.float32)
    out = np.empty(2 * mfcc.shape[1], np.float32)
    lib().orc_mfcc_stats(mfcc.reshape(-1), mfcc.shape[0], mfcc.shape[1], out)
    return out


def svm_predict(model: dict, x: np.ndarray):
    """model: dict with offset, scale, sv [n_sv][n_f], coef, gamma, rho, prob_a, prob_b."""
    keep = [np.ascontiguousarray(model[k], np.float32) for k in ("offset", "scale", "sv", "coef")]
    m = SvmModel()
    m.n_features = keep[0].size
    m.n_sv = keep[3].size
    m.gamma, m.rho = float(model["gamma"]), float(model["rho"])
    m.prob_a, m.prob_b = float(model["prob_a"]), float(model["prob_b"])
    m.offset, m.scale, m.sv, m.coef = [k.ctypes.data_as(C.POINTER(C.c_float)) for k in keep]
    dec, p1 = C.c_float(), C.c_float()
    label = lib().orc_svm_predict(C.byref(m), np.ascontiguousarray(x, np.float32), C.byref(dec), C.byref(p1))
    return int(label), float(dec.value), float(p1.value)


# ---- consumers of the MFCC matrix ------------------------------------------------

def _stop_model(model: dict):
    """model: dict with scaler_mean, scaler_scale [n_coef*max_frames], kernel0..3, bias0..3."""
    keep = {k: np.ascontiguousarray(model[k], np.float32) for k in
            ["scaler_mean", "scaler_scale"] + [f"kernel{i}" for i in range(4)] + [f"bias{i}" for i in range(4)]}
    m = StopModel()
    m.n_coef, m.max_frames = int(model.get("n_coef", 13)), int(model.get("max_frames", 500))
    for i in range(4):
        m.units[i] = keep[f"bias{i}"].size
        m.kernel[i] = keep[f"kernel{i}"].ctypes.data_as(C.POINTER(C.c_float))
        m.bias[i] = keep[f"bias{i}"].ctypes.data_as(C.POINTER(C.c_float))
    m.scaler_mean = keep["scaler_mean"].ctypes.data_as(C.POINTER(C.c_float))
    m.scaler_scale = keep["scaler_scale"].ctypes.data_as(C.POINTER(C.c_float))
    return m, keep


def stop_features(model: dict, mfcc: np.ndarray) -> np.ndarray:
    m, _keep = _stop_model(model)
    mfcc = np.ascontiguousarray(mfcc, np.float32)
    out = np.empty(m.n_coef * m.max_frames, np.float32)
    lib().orc_stop_features(C.byref(m), mfcc.reshape(-1), mfcc.shape[0], out)
    return out


def stop_predict(model: dict, feats: np.ndarray) -> float:
    m, _keep = _stop_model(model)
    return float(lib().orc_stop_predict(C.byref(m), np.ascontiguousarray(feats, np.float32).reshape(-1)))


def classify_signal(model: dict, signal: np.ndarray) -> float:
    m, _keep = _stop_model(model)
    signal = np.ascontiguousarray(signal, np.float32)
    return float(lib().orc_classify_signal(C.byref(m), signal, signal.size))


def _gmm(g: dict):
    keep = (np.ascontiguousarray(g["means"], np.int8), np.ascontiguousarray(g["inv_covs"], np.int32),
            np.ascontiguousarray(g["log_consts"], np.int16))
    s = Gmm()
    s.k, s.d = keep[0].shape
    s.means = keep[0].ctypes.data_as(C.POINTER(C.c_int8))
    s.inv_covs = keep[1].ctypes.data_as(C.POINTER(C.c_int32))
    s.log_consts = keep[2].ctypes.data_as(C.POINTER(C.c_int16))
    return s, keep


def float_to_q6(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.size, np.int16)
    lib().orc_float_to_q6(x.reshape(-1), out, x.size)
    return out.reshape(x.shape)


def gmm_log_likelihood(g: dict, x_q6: np.ndarray) -> int:
    s, _keep = _gmm(g)
    return int(lib().orc_gmm_log_likelihood(C.byref(s), np.ascontiguousarray(x_q6, np.int16)))


def speaker_llr_mean(target: dict, ubm: dict, mfcc: np.ndarray) -> int:
    t, _k1 = _gmm(target)
    u, _k2 = _gmm(ubm)
    mfcc = np.ascontiguousarray(mfcc, np.float32)
    return int(lib().orc_speaker_llr_mean(C.byref(t), C.byref(u), mfcc.reshape(-1), mfcc.shape[0]))


def classify_speaker(target: dict, ubm: dict, mfcc: np.ndarray) -> int:
    t, _k1 = _gmm(target)
    u, _k2 = _gmm(ubm)
    mfcc = np.ascontiguousarray(mfcc, np.float32)
    return int(lib().orc_classify_speaker(C.byref(t), C.byref(u), mfcc.reshape(-1), mfcc.shape[0]))


def upsample_linear(x: np.ndarray, new_size: int) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(new_size, np.float32)
    lib().orc_upsample_linear(x, x.size, out, new_size)
    return out


# ---- the reference itself (only where oracle/_ref was built) -------------------

def classify_f64(data, cfg=None):
    """donut-classifier/classifier.c per clip (float64): -> (label, midpoints[n_mid], sums[n_mid][3]).  cfg: dict of the six
    thresholds (keep_lo, keep_hi, midpoint_db, middle_max, above_min, below_min) or None for the file's own."""
    data = np.ascontiguousarray(data, np.float64)
    tr = ClassifyTraceF64()
    c = None
    if cfg is not None:
        c = ClassifyCfgF64(*[float(cfg[k]) for k in ("keep_lo", "keep_hi", "midpoint_db", "middle_max", "above_min", "below_min")])
    label = lib().orc_classify_f64(data, data.size, C.byref(c) if c is not None else None, C.byref(tr))
    n = tr.n_midpoints
    return label, np.array(tr.midpoints[:n], np.float64), np.array([list(r) for r in tr.sums[:n]], np.float64).reshape(n, 3)


def find_midpoints_f64(data, fs=16000, threshold_db=45.0) -> np.ndarray:
    data = np.ascontiguousarray(data, np.float64)
    out = np.empty(64, np.float64)
    n = lib().orc_find_midpoints_f64(data, data.size, fs, float(threshold_db), out, 64)
    return out[:n]


# ---- aubio front end of cepstrum/scrubjay_infer.c:21-53 (aubio_oracle.c, parity unpinned) ----

def aubio_window_hanningz(n: int) -> np.ndarray:
    w = np.empty(n, np.float32)
    lib().orc_aubio_window_hanningz(n, w)
    return w


def aubio_filterbank_slaney(sample_rate: int = 16000, win_s: int = 2048) -> np.ndarray:
    fb = np.empty((40, win_s // 2 + 1), np.float32)
    lib().orc_aubio_filterbank_slaney(sample_rate, win_s, fb)
    return fb


def aubio_frames_for(n: int, hop_s: int = 1024) -> int:
    return lib().orc_aubio_frames_for(n, hop_s)


def aubio_mfcc_clip(signal: np.ndarray, sample_rate: int = 16000, win_s: int = 2048, hop_s: int = 1024, n_filters: int = 40,
                    n_coefs: int = 20) -> np.ndarray:
    """scrubjay_infer.c:41-45 per hop: [T][n_coefs], T = ceil(n / hop_s)."""
    signal = np.ascontiguousarray(signal, np.float32)
    T = aubio_frames_for(signal.size, hop_s)
    out = np.empty((max(T, 1), n_coefs), np.float32)
    got = lib().orc_aubio_mfcc_clip(signal, signal.size, sample_rate, win_s, hop_s, n_filters, n_coefs, out)
    if got < 0:
        raise ValueError("orc_aubio_mfcc_clip: unsupported arguments")
    return out[:got]


def have_ref() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_mfcc.so"))


_ref_mfcc = None
_ref_cls = None


def ref_mfcc_lib() -> C.CDLL:
    global _ref_mfcc
    if _ref_mfcc is None:
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_mfcc.so"))
        L.compute_mfcc.argtypes = [_F, C.c_int, _F, C.c_int]
        L.compute_mfcc.restype = C.c_int
        L.fft_real_forward.argtypes = [_F, _F]
        for name in ("ref_hann_window",):
            getattr(L, name).argtypes = [C.POINTER(C.c_int)]
            getattr(L, name).restype = C.POINTER(C.c_float)
        for name in ("ref_mel_filter", "ref_dct_matrix"):
            getattr(L, name).argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
            getattr(L, name).restype = C.POINTER(C.c_float)
        _ref_mfcc = L
    return _ref_mfcc


def ref_compute_mfcc(signal: np.ndarray, max_frames: int) -> np.ndarray:
    signal = np.ascontiguousarray(signal, np.float32)
    out = np.zeros((max(max_frames, 1), 13), np.float32)
    t = ref_mfcc_lib().compute_mfcc(signal, signal.size, out.reshape(-1), max_frames)
    return out[:t]


def ref_tables():
    L = ref_mfcc_lib()
    n, r, c = C.c_int(), C.c_int(), C.c_int()
    p = L.ref_hann_window(C.byref(n))
    hann = np.ctypeslib.as_array(p, (n.value,)).copy()
    p = L.ref_mel_filter(C.byref(r), C.byref(c))
    mel = np.ctypeslib.as_array(p, (r.value, c.value)).copy()
    p = L.ref_dct_matrix(C.byref(r), C.byref(c))
    dct = np.ctypeslib.as_array(p, (r.value, c.value)).copy()
    return hann, mel, dct


def ref_classifier_lib() -> C.CDLL:
    global _ref_cls
    if _ref_cls is None:
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_classifier.so"))
        L.ref_classify.argtypes = [_F, C.c_int]
        L.ref_classify.restype = C.c_int
        L.ref_butter_bandpass.argtypes = [C.c_float, C.c_float, _F, _F]
        L.ref_butter_bandpass.restype = C.c_int
        L.ref_butter_bandpass_filter.argtypes = [_F, C.c_int, _F, _F, _F]
        L.ref_compute_spectrogram.argtypes = [_F, C.c_int, C.c_int, _F, _F, _F]
        L.ref_compute_spectrogram.restype = C.c_int
        L.ref_find_midpoints.argtypes = [_F, C.c_int, C.c_int, _F, C.c_int]
        L.ref_find_midpoints.restype = C.c_int
        L.ref_sum_intense.argtypes = [C.c_float, C.c_float, C.c_float, _F, C.c_int, _F, C.c_int, _F, C.c_float]
        L.ref_sum_intense.restype = C.c_float
        _ref_cls = L
    return _ref_cls


_ref_stop = None
_ref_gmm = None


def ref_stop_lib() -> C.CDLL:
    """The reference's stop detector (2fa/audio/word/c: mfcc.c + stop_detector.c + audio_classifier_inference.c)."""
    global _ref_stop
    if _ref_stop is None:
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_stop.so"))
        L.classify_signal.argtypes = [_F, C.c_int]
        L.classify_signal.restype = C.c_float
        L.audio_classifier_predict.argtypes = [_F]
        L.audio_classifier_predict.restype = C.c_float
        for name in ("ref_stop_scaler_mean", "ref_stop_scaler_scale"):
            getattr(L, name).restype = C.POINTER(C.c_float)
        for name in ("ref_stop_kernel", "ref_stop_bias"):
            getattr(L, name).argtypes = [C.c_int]
            getattr(L, name).restype = C.POINTER(C.c_float)
        L.ref_stop_units.argtypes = [C.POINTER(C.c_int)]
        _ref_stop = L
    return _ref_stop


def ref_stop_model() -> dict:
    """The trained parameters compiled into the reference (model_params.h), as arrays."""
    L = ref_stop_lib()
    n_in = L.ref_stop_input_size()
    units = (C.c_int * 4)()
    L.ref_stop_units(units)
    m = {"n_coef": 13, "max_frames": n_in // 13,
         "scaler_mean": np.ctypeslib.as_array(L.ref_stop_scaler_mean(), (n_in,)).copy(),
         "scaler_scale": np.ctypeslib.as_array(L.ref_stop_scaler_scale(), (n_in,)).copy()}
    fan_in = n_in
    for i in range(4):
        m[f"kernel{i}"] = np.ctypeslib.as_array(L.ref_stop_kernel(i), (fan_in * units[i],)).copy()
        m[f"bias{i}"] = np.ctypeslib.as_array(L.ref_stop_bias(i), (units[i],)).copy()
        fan_in = units[i]
    return m


def ref_gmm_lib() -> C.CDLL:
    """The reference's speaker GMM (2fa/audio/pico-audio/src/speaker_gmm.c + gmm_params.inc)."""
    global _ref_gmm
    if _ref_gmm is None:
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_gmm.so"))
        _I16 = np.ctypeslib.ndpointer(dtype=np.int16, flags="C_CONTIGUOUS")
        L.target_gmm_log_likelihood.argtypes = [_I16]
        L.target_gmm_log_likelihood.restype = C.c_int64
        L.ubm_gmm_log_likelihood.argtypes = [_I16]
        L.ubm_gmm_log_likelihood.restype = C.c_int64
        L.float_to_g6int16_arr.argtypes = [_F, _I16, C.c_int]
        L.mfcc_target_speaker_llr_mean.argtypes = [_F, C.c_int]
        L.mfcc_target_speaker_llr_mean.restype = C.c_int64
        L.classify_speaker.argtypes = [_F, C.c_int]
        L.classify_speaker.restype = C.c_int
        _ref_gmm = L
    return _ref_gmm


def ref_gmm_params() -> tuple[dict, dict]:
    """(target, ubm) parameter tables of the reference's gmm_params.inc (K = 32, D = 13)."""
    L = ref_gmm_lib()
    out = []
    for who in ("target", "ubm"):
        out.append({
            "means": np.ctypeslib.as_array((C.c_int8 * (32 * 13)).in_dll(L, f"{who}_means")).reshape(32, 13).copy(),
            "inv_covs": np.ctypeslib.as_array((C.c_int32 * (32 * 13)).in_dll(L, f"{who}_inv_covs")).reshape(32, 13).copy(),
            "log_consts": np.ctypeslib.as_array((C.c_int16 * 32).in_dll(L, f"{who}_log_consts")).copy(),
        })
    return out[0], out[1]
