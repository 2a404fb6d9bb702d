"""ctypes binding of libdsp_amd.so (include/dsp_amd.h).  Fails loudly: there is
no Python/CPU implementation behind these calls."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build


class DspError(RuntimeError):
    pass


class MfccConfig(C.Structure):
    """dsp_mfcc_config (include/dsp_amd.h); defaults = mfcc_params.h:6-12 of the reference."""

    _fields_ = [
        ("sample_rate", C.c_int), ("n_fft", C.c_int), ("frame_length", C.c_int),
        ("hop_length", C.c_int), ("n_mels", C.c_int), ("n_mfcc", C.c_int),
        ("window", C.c_int), ("mel_norm", C.c_int), ("log_mode", C.c_int),
        ("prefilter", C.c_int), ("win_length", C.c_int),
        ("fmin", C.c_float), ("fmax", C.c_float), ("amin", C.c_float), ("top_db", C.c_float),
        ("spectrum", C.c_int), ("framing", C.c_int),
    ]


class LaneTables512(C.Structure):
    """dsp::LaneTables512 (dsp_amd/csrc/tables.hpp): the per-lane kernel layout of the constant tables."""

    _fields_ = [
        ("win", (C.c_float * 64) * 8), ("tw1", (C.c_float * 64) * 6), ("tw2", (C.c_float * 64) * 6),
        ("tw3", (C.c_float * 64) * 6), ("twp", (C.c_float * 64) * 4),
        ("kappa", C.c_int32 * 64), ("partner", C.c_int32 * 64),
        ("mel_k0", C.c_int32 * 64), ("mel_w", (C.c_float * 64) * 12), ("mel_src", (C.c_int32 * 64) * 6),
        ("mel_gather", C.c_int32), ("mel_conflict_free", C.c_int32),
        ("dct_w", (C.c_float * 64) * 20), ("dct_split", C.c_int32), ("dct_len", C.c_int32),
        ("n_mels", C.c_int32), ("n_mfcc", C.c_int32),
        ("dct_a", ((C.c_float * 64) * 16) * 2),
    ]


class StopModelParams(C.Structure):
    """dsp_stop_model_params (include/dsp_amd.h)."""

    _fields_ = [("n_coef", C.c_int), ("max_frames", C.c_int), ("units", C.c_int * 4),
                ("scaler_mean", C.c_void_p), ("scaler_scale", C.c_void_p),
                ("kernel", C.c_void_p * 4), ("bias", C.c_void_p * 4)]


class GmmParams(C.Structure):
    """dsp_gmm_params (include/dsp_amd.h)."""

    _fields_ = [("k", C.c_int), ("d", C.c_int), ("means", C.c_void_p), ("inv_covs", C.c_void_p), ("log_consts", C.c_void_p)]


class ClassifyTrace(C.Structure):
    """dsp_classify_trace (include/dsp_amd.h)."""

    _fields_ = [("n_midpoints", C.c_int), ("midpoints", C.c_float * 64), ("sums", (C.c_float * 3) * 64)]


class ClassifyConfig(C.Structure):
    """dsp_classify_config (include/dsp_amd.h): the thresholds the reference's classify() variants differ in."""

    _fields_ = [("keep_lo", C.c_float), ("keep_hi", C.c_float), ("midpoint_db", C.c_float),
                ("middle_max", C.c_float), ("above_min", C.c_float), ("below_min", C.c_float)]


class ClassifyConfigF64(C.Structure):
    """dsp_classify_config_f64: the same thresholds as doubles, for the float64 classifier (donut-classifier/classifier.c)."""

    _fields_ = [(k, C.c_double) for k in ("keep_lo", "keep_hi", "midpoint_db", "middle_max", "above_min", "below_min")]


class ClassifyTraceF64(C.Structure):
    _fields_ = [("n_midpoints", C.c_int), ("midpoints", C.c_double * 64), ("sums", (C.c_double * 3) * 64)]


WINDOW_HANN, WINDOW_HAMMING, WINDOW_RECT = 0, 1, 2
MELNORM_NONE, MELNORM_SLANEY, MELNORM_LIBROSA, MELNORM_AUBIO_SLANEY = 0, 1, 2, 3
LOG_PER_FRAME_MAX, LOG_GLOBAL_REF1, LOG_LOG10_FLOOR = 0, 1, 2
SPECTRUM_POWER, SPECTRUM_MAGNITUDE = 0, 1
FRAMING_COMPLETE, FRAMING_STREAM = 0, 1
PREFILTER_NONE, PREFILTER_BUTTER_1000_3000, PREFILTER_BUTTER_3000_7500 = 0, 1, 2

# every symbol include/dsp_amd.h declares (tests check the library exports them all)
SYMBOLS = [
    "compute_mfcc", "fft_real_forward", "dsp_fft_real_forward_host", "dsp_classify",
    "dsp_classify_default_config_f64", "dsp_classify_batch_host_f64", "dsp_classify_batch_device_f64",
    "dsp_classify_batch_pcm16_host_f64", "dsp_classify_batch_pcm16_device_f64", "dsp_classify_release_f64", "dsp_classify_stats_f64", "dsp_classify_debug_f64",
    "dsp_classify_default_config", "dsp_classify_batch_host_cfg", "dsp_classify_batch_device_cfg", "dsp_sum_intense_f32",
    "dsp_butter_bandpass_filter_f32", "dsp_butter_bandpass_filter_f64", "dsp_compute_spectrogram_f32", "dsp_compute_spectrogram_f64",
    "dsp_classify_batch_host", "dsp_classify_batch_device", "dsp_classify_batch_pcm16_host", "dsp_classify_batch_pcm16_device", "dsp_classify_release", "dsp_classify_stats", "dsp_classify_ctx_create", "dsp_classify_ctx_destroy", "dsp_classify_batch_device_ctx", "dsp_debug_hold_classify_ctx", "dsp_find_midpoints", "dsp_classify_division_check",
    "dsp_mfcc_stats_device", "dsp_svm_create", "dsp_svm_destroy", "dsp_svm_predict_device",
    "dsp_mfcc_default_config", "dsp_mfcc_scrubjay_infer_config", "dsp_mfcc_plan_create", "dsp_mfcc_plan_destroy", "dsp_mfcc_plan_config",
    "dsp_mfcc_frames_for", "dsp_mfcc_frames_device", "dsp_mfcc_clips_device", "dsp_mfcc_frames_host",
    "dsp_mfcc_clips_host", "dsp_mfcc_clips_pcm16_device", "dsp_mfcc_plan_set_launch", "dsp_mfcc_plan_set_kernel", "dsp_butter_bandpass", "dsp_mfcc_tables", "dsp_mfcc_lane_tables", "dsp_prefilter_scan_check",
    "dsp_classify_batch_ragged_device_f64", "dsp_classify_batch_ragged_pcm16_device_f64", "dsp_classify_batch_ragged_host_f64", "dsp_classify_batch_ragged_pcm16_host_f64",
    "dsp_classify_batch_ragged_device", "dsp_classify_batch_ragged_pcm16_device", "dsp_classify_batch_ragged_host", "dsp_classify_batch_ragged_pcm16_host",
    "dsp_debug_fused_spans", "dsp_scrubjay_fused_ragged_device", "dsp_scrubjay_fused_ragged_pcm16_device", "dsp_classify_signal_batch_ragged_device", "dsp_classify_signal_batch_ragged_pcm16_device",
    "dsp_scrubjay_fused_device", "dsp_scrubjay_fused_pcm16_device", "dsp_classify_signal_batch_pcm16_device", "dsp_stop_model_create", "dsp_stop_model_destroy", "dsp_stop_predict_device", "dsp_classify_signal_batch_device",
    "dsp_classify_signal", "dsp_speaker_model_create", "dsp_speaker_model_destroy", "dsp_speaker_llr_device",
    "dsp_upsample_linear_device", "dsp_upsample_linear_host",
    "dsp_gather_create", "dsp_gather_destroy", "dsp_gather_n_devices", "dsp_gather_all",
    "dsp_last_error", "dsp_device_count", "dsp_version", "dsp_abi_sizeof",
]

# the reference's own C++-linkage names (sync/lib/classifier.h:14-19, include/dsp_amd_classifier.h), Itanium-mangled
CXX_SYMBOLS = {
    "butter_bandpass": "_Z15butter_bandpassffPfS_",
    "butter_bandpass_filter": "_Z22butter_bandpass_filterPfiS_S_S_",
    "compute_spectrogram": "_Z19compute_spectrogramPfiiPS_S0_PS0_PiS2_",
    "sum_intense": "_Z11sum_intensefffPfiS_iPS_f",
    "find_midpoints": "_Z14find_midpointsPfiiPi",
    "classify": "_Z8classifyPfi",
}

_lib = None


def load() -> C.CDLL:
    """Load (building first if the sources are newer) the in-tree libdsp_amd.so."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    explicit = bool(os.environ.get("DSP_AMD_LIB"))               # an explicit DSP_AMD_LIB is loaded as is
    if not explicit and _build.is_stale():
        try:
            _build.build()
        except Exception as e:  # noqa: BLE001  (never fall back to an older binary: it would pass for the current source)
            raise DspError(f"libdsp_amd.so is stale or missing and could not be rebuilt: {e}") from e
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own
    # libamdhip64.so (SONAME libamdhip64.so.7, the same as /opt/rocm's).  If torch is
    # going to share streams and HBM buffers with this library it must be loaded
    # first, so that our DT_NEEDED libamdhip64.so.7 binds to the runtime torch
    # already brought in instead of a second copy from /opt/rocm.
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001  (pure C-ABI use without torch is fine)
        pass
    try:
        L = C.CDLL(path)
    except OSError as e:
        raise DspError(f"cannot load {path}: {e} (the HIP extension is required; there is no fallback)") from e
    vp, fp, ip = C.c_void_p, C.POINTER(C.c_float), C.c_int
    cfgp = C.POINTER(MfccConfig)
    L.dsp_abi_sizeof.argtypes = [ip]; L.dsp_abi_sizeof.restype = ip
    for which, mirror in ((0, MfccConfig), (1, ClassifyTrace), (2, ClassifyTraceF64)):      # the ctypes mirrors against the library's own structs
        if L.dsp_abi_sizeof(which) != C.sizeof(mirror):
            raise DspError(f"{path}: struct {mirror.__name__} is {L.dsp_abi_sizeof(which)} bytes in the library, {C.sizeof(mirror)} in dsp_amd/lib.py")
    L.compute_mfcc.argtypes = [vp, ip, vp, ip]; L.compute_mfcc.restype = ip
    L.fft_real_forward.argtypes = [vp, vp]; L.fft_real_forward.restype = None
    L.dsp_fft_real_forward_host.argtypes = [vp, C.c_long, ip, C.c_long, ip, vp]; L.dsp_fft_real_forward_host.restype = ip
    L.dsp_mfcc_default_config.argtypes = [cfgp]; L.dsp_mfcc_default_config.restype = None
    L.dsp_mfcc_scrubjay_infer_config.argtypes = [cfgp, ip]; L.dsp_mfcc_scrubjay_infer_config.restype = None
    L.dsp_mfcc_plan_create.argtypes = [cfgp, ip, C.POINTER(vp)]; L.dsp_mfcc_plan_create.restype = ip
    L.dsp_mfcc_plan_destroy.argtypes = [vp]; L.dsp_mfcc_plan_destroy.restype = None
    L.dsp_mfcc_plan_config.argtypes = [vp, cfgp]; L.dsp_mfcc_plan_config.restype = ip
    L.dsp_mfcc_frames_for.argtypes = [cfgp, ip, ip]; L.dsp_mfcc_frames_for.restype = ip
    L.dsp_mfcc_frames_device.argtypes = [vp, vp, C.c_long, vp, vp]; L.dsp_mfcc_frames_device.restype = ip
    L.dsp_mfcc_clips_device.argtypes = [vp, vp, C.c_long, ip, C.c_long, vp, ip, vp]; L.dsp_mfcc_clips_device.restype = ip
    L.dsp_mfcc_clips_pcm16_device.argtypes = [vp, vp, C.c_long, ip, C.c_long, ip, ip, vp, ip, vp]; L.dsp_mfcc_clips_pcm16_device.restype = ip
    L.dsp_mfcc_frames_host.argtypes = [vp, vp, C.c_long, vp]; L.dsp_mfcc_frames_host.restype = ip
    L.dsp_mfcc_clips_host.argtypes = [vp, vp, C.c_long, ip, C.c_long, vp, ip]; L.dsp_mfcc_clips_host.restype = ip
    L.dsp_mfcc_plan_set_launch.argtypes = [vp, ip, ip]; L.dsp_mfcc_plan_set_launch.restype = ip
    L.dsp_mfcc_plan_set_kernel.argtypes = [vp, ip]; L.dsp_mfcc_plan_set_kernel.restype = ip
    L.dsp_butter_bandpass.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.dsp_butter_bandpass.restype = ip
    L.dsp_mfcc_tables.argtypes = [cfgp, vp, vp, vp]; L.dsp_mfcc_tables.restype = ip
    L.dsp_mfcc_lane_tables.argtypes = [cfgp, vp, ip]; L.dsp_mfcc_lane_tables.restype = ip
    L.dsp_classify.argtypes = [vp, ip]; L.dsp_classify.restype = ip
    L.dsp_butter_bandpass_filter_f32.argtypes = [vp, C.c_long, ip, C.c_long, vp, vp, vp]; L.dsp_butter_bandpass_filter_f32.restype = ip
    L.dsp_butter_bandpass_filter_f64.argtypes = [vp, C.c_long, ip, C.c_long, vp, vp, vp]; L.dsp_butter_bandpass_filter_f64.restype = ip
    L.dsp_compute_spectrogram_f32.argtypes = [vp, ip, ip, vp, vp, vp]; L.dsp_compute_spectrogram_f32.restype = ip
    L.dsp_compute_spectrogram_f64.argtypes = [vp, ip, ip, vp, vp, vp]; L.dsp_compute_spectrogram_f64.restype = ip
    L.dsp_classify_batch_host.argtypes = [vp, C.c_long, ip, C.c_long, vp, vp]; L.dsp_classify_batch_host.restype = ip
    L.dsp_find_midpoints.argtypes = [vp, ip, ip, vp, ip]; L.dsp_find_midpoints.restype = ip
    L.dsp_classify_batch_device.argtypes = [vp, C.c_long, ip, C.c_long, vp, vp]; L.dsp_classify_batch_device.restype = ip
    L.dsp_mfcc_stats_device.argtypes = [vp, C.c_long, ip, ip, vp, vp]; L.dsp_mfcc_stats_device.restype = ip
    L.dsp_svm_create.argtypes = [ip, ip, ip, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(vp)]
    L.dsp_svm_create.restype = ip
    L.dsp_svm_destroy.argtypes = [vp]; L.dsp_svm_destroy.restype = None
    L.dsp_svm_predict_device.argtypes = [vp, vp, C.c_long, vp, vp, vp, vp]; L.dsp_svm_predict_device.restype = ip
    L.dsp_scrubjay_fused_device.argtypes = [vp, vp, vp, C.c_long, ip, C.c_long, ip, vp, vp, vp, vp, vp]; L.dsp_scrubjay_fused_device.restype = ip
    L.dsp_scrubjay_fused_pcm16_device.argtypes = [vp, vp, vp, C.c_long, ip, C.c_long, ip, ip, ip, vp, vp, vp, vp, vp]; L.dsp_scrubjay_fused_pcm16_device.restype = ip
    lp = C.POINTER(C.c_long)
    L.dsp_classify_batch_ragged_device_f64.argtypes = [vp, vp, C.c_long, lp, vp, vp, vp]; L.dsp_classify_batch_ragged_device_f64.restype = ip
    L.dsp_classify_batch_ragged_pcm16_device_f64.argtypes = [vp, vp, C.c_long, lp, ip, ip, vp, vp, vp]; L.dsp_classify_batch_ragged_pcm16_device_f64.restype = ip
    L.dsp_classify_batch_ragged_host_f64.argtypes = [vp, vp, C.c_long, lp, vp, vp]; L.dsp_classify_batch_ragged_host_f64.restype = ip
    L.dsp_classify_batch_ragged_pcm16_host_f64.argtypes = [vp, vp, C.c_long, lp, ip, ip, vp, vp]; L.dsp_classify_batch_ragged_pcm16_host_f64.restype = ip
    L.dsp_classify_stats.argtypes = [ip, lp, lp]; L.dsp_classify_stats.restype = ip
    L.dsp_classify_batch_ragged_device.argtypes = [vp, vp, C.c_long, lp, vp, vp]; L.dsp_classify_batch_ragged_device.restype = ip
    L.dsp_classify_batch_ragged_pcm16_device.argtypes = [vp, vp, C.c_long, lp, ip, ip, vp, vp]; L.dsp_classify_batch_ragged_pcm16_device.restype = ip
    L.dsp_classify_batch_ragged_host.argtypes = [vp, vp, C.c_long, lp, vp, vp]; L.dsp_classify_batch_ragged_host.restype = ip
    L.dsp_classify_batch_ragged_pcm16_host.argtypes = [vp, vp, C.c_long, lp, ip, ip, vp, vp]; L.dsp_classify_batch_ragged_pcm16_host.restype = ip
    L.dsp_debug_fused_spans.argtypes = [cfgp, lp, C.c_long, ip, C.c_long, lp]; L.dsp_debug_fused_spans.restype = ip
    L.dsp_scrubjay_fused_ragged_device.argtypes = [vp, vp, vp, C.c_long, lp, ip, vp, vp, vp, vp, vp]; L.dsp_scrubjay_fused_ragged_device.restype = ip
    L.dsp_scrubjay_fused_ragged_pcm16_device.argtypes = [vp, vp, vp, C.c_long, lp, ip, ip, ip, vp, vp, vp, vp, vp]; L.dsp_scrubjay_fused_ragged_pcm16_device.restype = ip
    L.dsp_classify_signal_batch_ragged_device.argtypes = [vp, vp, vp, C.c_long, lp, vp, vp]; L.dsp_classify_signal_batch_ragged_device.restype = ip
    L.dsp_classify_signal_batch_ragged_pcm16_device.argtypes = [vp, vp, vp, C.c_long, lp, ip, ip, vp, vp]; L.dsp_classify_signal_batch_ragged_pcm16_device.restype = ip
    L.dsp_classify_signal_batch_pcm16_device.argtypes = [vp, vp, vp, C.c_long, ip, C.c_long, ip, ip, vp, vp]; L.dsp_classify_signal_batch_pcm16_device.restype = ip
    L.dsp_stop_model_create.argtypes = [C.POINTER(StopModelParams), ip, C.POINTER(vp)]; L.dsp_stop_model_create.restype = ip
    L.dsp_stop_model_destroy.argtypes = [vp]; L.dsp_stop_model_destroy.restype = None
    L.dsp_stop_predict_device.argtypes = [vp, vp, C.c_long, ip, vp, vp]; L.dsp_stop_predict_device.restype = ip
    L.dsp_classify_signal_batch_device.argtypes = [vp, vp, vp, C.c_long, ip, C.c_long, vp, vp]; L.dsp_classify_signal_batch_device.restype = ip
    L.dsp_classify_signal.argtypes = [vp, vp, ip]; L.dsp_classify_signal.restype = C.c_float
    L.dsp_speaker_model_create.argtypes = [C.POINTER(GmmParams), C.POINTER(GmmParams), ip, C.POINTER(vp)]; L.dsp_speaker_model_create.restype = ip
    L.dsp_speaker_model_destroy.argtypes = [vp]; L.dsp_speaker_model_destroy.restype = None
    L.dsp_speaker_llr_device.argtypes = [vp, vp, C.c_long, ip, vp, vp, vp, vp, vp]; L.dsp_speaker_llr_device.restype = ip
    L.dsp_upsample_linear_device.argtypes = [vp, C.c_long, ip, C.c_long, vp, ip, C.c_long, vp]; L.dsp_upsample_linear_device.restype = ip
    L.dsp_upsample_linear_host.argtypes = [vp, ip, vp, ip]; L.dsp_upsample_linear_host.restype = ip
    L.dsp_gather_create.argtypes = [vp, ip, C.POINTER(vp)]; L.dsp_gather_create.restype = ip
    L.dsp_gather_destroy.argtypes = [vp]; L.dsp_gather_destroy.restype = None
    L.dsp_gather_n_devices.argtypes = [vp]; L.dsp_gather_n_devices.restype = ip
    L.dsp_gather_all.argtypes = [vp, vp, vp, C.c_size_t, vp]; L.dsp_gather_all.restype = ip
    L.dsp_last_error.argtypes = []; L.dsp_last_error.restype = C.c_char_p
    L.dsp_device_count.argtypes = []; L.dsp_device_count.restype = ip
    L.dsp_version.argtypes = []; L.dsp_version.restype = C.c_char_p
    L.dsp_classify_default_config.argtypes = [C.POINTER(ClassifyConfig)]; L.dsp_classify_default_config.restype = None
    L.dsp_classify_batch_host_cfg.argtypes = [C.POINTER(ClassifyConfig), vp, C.c_long, ip, C.c_long, vp, vp]; L.dsp_classify_batch_host_cfg.restype = ip
    L.dsp_classify_default_config_f64.argtypes = [C.POINTER(ClassifyConfigF64)]; L.dsp_classify_default_config_f64.restype = None
    L.dsp_classify_batch_host_f64.argtypes = [C.POINTER(ClassifyConfigF64), vp, C.c_long, ip, C.c_long, vp, vp]; L.dsp_classify_batch_host_f64.restype = ip
    L.dsp_classify_batch_device_f64.argtypes = [C.POINTER(ClassifyConfigF64), vp, C.c_long, ip, C.c_long, vp, vp, vp]; L.dsp_classify_batch_device_f64.restype = ip
    L.dsp_classify_batch_pcm16_host.argtypes = [C.POINTER(ClassifyConfig), vp, C.c_long, ip, C.c_long, ip, ip, vp, vp]; L.dsp_classify_batch_pcm16_host.restype = ip
    L.dsp_classify_batch_pcm16_device.argtypes = [C.POINTER(ClassifyConfig), vp, C.c_long, ip, C.c_long, ip, ip, vp, vp]; L.dsp_classify_batch_pcm16_device.restype = ip
    L.dsp_classify_release.argtypes = [ip]; L.dsp_classify_release.restype = ip
    L.dsp_classify_ctx_create.argtypes = [ip, C.POINTER(vp)]; L.dsp_classify_ctx_create.restype = ip
    L.dsp_classify_ctx_destroy.argtypes = [vp]; L.dsp_classify_ctx_destroy.restype = None
    L.dsp_classify_batch_device_ctx.argtypes = [vp, C.POINTER(ClassifyConfig), vp, C.c_long, ip, C.c_long, vp, vp]; L.dsp_classify_batch_device_ctx.restype = ip
    L.dsp_debug_hold_classify_ctx.argtypes = [ip, ip]; L.dsp_debug_hold_classify_ctx.restype = ip
    L.dsp_classify_batch_pcm16_host_f64.argtypes = [C.POINTER(ClassifyConfigF64), vp, C.c_long, ip, C.c_long, ip, ip, vp, vp]; L.dsp_classify_batch_pcm16_host_f64.restype = ip
    L.dsp_classify_batch_pcm16_device_f64.argtypes = [C.POINTER(ClassifyConfigF64), vp, C.c_long, ip, C.c_long, ip, ip, vp, vp, vp]; L.dsp_classify_batch_pcm16_device_f64.restype = ip
    L.dsp_classify_release_f64.argtypes = [ip]; L.dsp_classify_release_f64.restype = ip
    L.dsp_classify_stats_f64.argtypes = [ip, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]; L.dsp_classify_stats_f64.restype = ip
    L.dsp_classify_batch_device_cfg.argtypes = [C.POINTER(ClassifyConfig), vp, C.c_long, ip, C.c_long, vp, vp]; L.dsp_classify_batch_device_cfg.restype = ip
    L.dsp_sum_intense_f32.argtypes = [C.c_float, C.c_float, C.c_float, vp, ip, vp, ip, vp, C.c_float, C.POINTER(C.c_float)]
    L.dsp_sum_intense_f32.restype = ip
    if not explicit:
        # the binary says which sources it was built from: a mismatch means the mtime check was fooled (copied tree,
        # clock skew) and the library on disk is not the code in dsp_amd/csrc
        built = L.dsp_version().decode().rsplit("src:", 1)[-1]
        if built != _build.source_hash():
            raise DspError(f"{path} was built from other sources (library {built}, tree {_build.source_hash()}): "
                           "rebuild with `python -m dsp_amd.build`")
    _lib = L
    return L


def last_error() -> str:
    return load().dsp_last_error().decode()


def c_offsets(offsets):
    """offsets[n_clips + 1] (any integer sequence) -> (ctypes long array, n_clips) for the *_ragged_* entry points."""
    import numpy as np
    a = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    assert a.ndim == 1 and a.size >= 1, "offsets must hold n_clips + 1 positions"
    return (C.c_long * a.size)(*a.tolist()), a.size - 1


def check(rc: int, what: str) -> int:
    if rc < 0:
        raise DspError(f"{what} failed ({rc}): {last_error()}")
    return rc
