"""Host-side mirror of the reference's consumers of the MFCC matrix and of its resampler, over the C ABI:

  StopModel      classify_signal / audio_classifier_predict   2fa/audio/word/c/stop_detector.c:12-55,
                                                              audio_classifier_inference.c:38-90
  SpeakerModel   mfcc_target_speaker_llr_mean / classify_speaker   2fa/audio/pico-audio/src/speaker_gmm.c:127-141
  upsample_linear   upsampleLinear                             sync/particle/main.cpp:62-77

Trained parameters are passed in as arrays (the reference compiles them in from model_params.h / gmm_params.inc).
Tensors are HBM-resident torch tensors; Python only moves pointers.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .mfcc import MfccPlan, default_config


def _stream(t):
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class StopModel:
    """dsp_stop_model: StandardScaler + 4 dense layers (ReLU, ReLU, ReLU, sigmoid)."""

    def __init__(self, params: dict, device: int = 0):
        """params: scaler_mean, scaler_scale [n_coef * max_frames], kernel0..3 ((in, out) row-major), bias0..3,
        optional n_coef (13), max_frames (500)."""
        self._L = _lib.load()
        keep = {k: np.ascontiguousarray(params[k], np.float32).reshape(-1) for k in
                ["scaler_mean", "scaler_scale"] + [f"kernel{i}" for i in range(4)] + [f"bias{i}" for i in range(4)]}
        p = _lib.StopModelParams()
        p.n_coef, p.max_frames = int(params.get("n_coef", 13)), int(params.get("max_frames", 500))
        if keep["scaler_mean"].size != p.n_coef * p.max_frames or keep["scaler_scale"].size != p.n_coef * p.max_frames:
            raise _lib.DspError("scaler arrays must have n_coef * max_frames entries")
        fan_in = p.n_coef * p.max_frames
        for i in range(4):
            p.units[i] = keep[f"bias{i}"].size
            if keep[f"kernel{i}"].size != fan_in * p.units[i]:
                raise _lib.DspError(f"kernel{i} must have {fan_in} x {p.units[i]} entries")
            p.kernel[i] = keep[f"kernel{i}"].ctypes.data
            p.bias[i] = keep[f"bias{i}"].ctypes.data
            fan_in = p.units[i]
        p.scaler_mean, p.scaler_scale = keep["scaler_mean"].ctypes.data, keep["scaler_scale"].ctypes.data
        h = C.c_void_p()
        _lib.check(self._L.dsp_stop_model_create(C.byref(p), device, C.byref(h)), "dsp_stop_model_create")
        self._h, self.device, self.n_coef, self.max_frames = h, device, p.n_coef, p.max_frames

    def close(self):
        if getattr(self, "_h", None):
            self._L.dsp_stop_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def predict(self, mfcc):
        """mfcc: cuda float32 [n_clips][T][n_coef] (frame-major, as compute_mfcc writes) -> prob float32 [n_clips]."""
        import torch
        n, t = mfcc.shape[0], mfcc.shape[1]
        prob = torch.empty(n, dtype=torch.float32, device=mfcc.device)
        _lib.check(self._L.dsp_stop_predict_device(self._h, mfcc.contiguous().data_ptr(), n, t, prob.data_ptr(), _stream(mfcc)),
                   "dsp_stop_predict_device")
        return prob

    def classify_signal_batch(self, plan: MfccPlan, clips):
        """clips: cuda float32 [n_clips][samples] -> P("stop") float32 [n_clips] (classify_signal per clip)."""
        import torch
        assert clips.dim() == 2 and clips.dtype == torch.float32
        prob = torch.empty(clips.shape[0], dtype=torch.float32, device=clips.device)
        _lib.check(self._L.dsp_classify_signal_batch_device(plan._h, self._h, clips.data_ptr(), clips.shape[0], clips.shape[1],
                                                            clips.stride(0), prob.data_ptr(), _stream(clips)),
                   "dsp_classify_signal_batch_device")
        return prob

    def classify_signal_batch_pcm16(self, plan: MfccPlan, pcm, stereo_mode: int = 0):
        """pcm: cuda int16 [n_clips][samples] or [n_clips][samples][2] -> P("stop") (dsp_classify_signal_batch_pcm16_device: what
        main_test.c:198-217 decodes in front of classify_signal, converted in the kernel's load)."""
        import torch
        assert pcm.dim() in (2, 3) and pcm.dtype == torch.int16 and pcm.stride(-1) == 1
        channels = 2 if pcm.dim() == 3 else 1
        prob = torch.empty(pcm.shape[0], dtype=torch.float32, device=pcm.device)
        _lib.check(self._L.dsp_classify_signal_batch_pcm16_device(plan._h, self._h, pcm.data_ptr(), pcm.shape[0], pcm.shape[1], pcm.stride(0) // channels,
                                                                  channels, int(stereo_mode), prob.data_ptr(), _stream(pcm)),
                   "dsp_classify_signal_batch_pcm16_device")
        return prob

    def classify_signal_ragged(self, plan: MfccPlan, signal, offsets, stereo_mode: int = 0):
        """Clips of different lengths in ONE launch (the files main_test.c:254-331 loops over): `signal` is a flat cuda buffer (float32
        [total], int16 [total] or stereo int16 [total][2]), clip c = samples [offsets[c], offsets[c + 1]) -> P("stop") per clip."""
        import torch
        off, n = offsets if isinstance(offsets, tuple) else _lib.c_offsets(offsets)      # (a prepared (ctypes array, n_clips) pair: no conversion per call)
        assert signal.is_cuda and signal.stride(-1) == 1 and int(off[n]) <= signal.shape[0]
        prob = torch.empty(n, dtype=torch.float32, device=signal.device)
        if signal.dtype == torch.float32:
            assert signal.dim() == 1
            _lib.check(self._L.dsp_classify_signal_batch_ragged_device(plan._h, self._h, signal.data_ptr(), n, off, prob.data_ptr(), _stream(signal)),
                       "dsp_classify_signal_batch_ragged_device")
        else:
            assert signal.dtype == torch.int16 and signal.dim() in (1, 2)
            _lib.check(self._L.dsp_classify_signal_batch_ragged_pcm16_device(plan._h, self._h, signal.data_ptr(), n, off, signal.dim(), int(stereo_mode),
                                                                             prob.data_ptr(), _stream(signal)), "dsp_classify_signal_batch_ragged_pcm16_device")
        return prob

    def classify_signal(self, signal: np.ndarray) -> float:
        """The reference's classify_signal(signal, num_samples) on a host buffer."""
        signal = np.ascontiguousarray(signal, np.float32)
        return float(self._L.dsp_classify_signal(self._h, signal.ctypes.data, signal.size))


def _gmm_params(g: dict):
    keep = (np.ascontiguousarray(g["means"], np.int8), np.ascontiguousarray(g["inv_covs"], np.int32),
            np.ascontiguousarray(g["log_consts"], np.int16))
    p = _lib.GmmParams()
    p.k, p.d = keep[0].shape
    p.means, p.inv_covs, p.log_consts = keep[0].ctypes.data, keep[1].ctypes.data, keep[2].ctypes.data
    return p, keep


class SpeakerModel:
    """dsp_speaker_model: target GMM vs UBM in the reference's fixed point (Q6 / Q11 / Q8)."""

    def __init__(self, target: dict, ubm: dict, device: int = 0):
        self._L = _lib.load()
        pt, _k1 = _gmm_params(target)
        pu, _k2 = _gmm_params(ubm)
        h = C.c_void_p()
        _lib.check(self._L.dsp_speaker_model_create(C.byref(pt), C.byref(pu), device, C.byref(h)), "dsp_speaker_model_create")
        self._h, self.device, self.d = h, device, pt.d

    def close(self):
        if getattr(self, "_h", None):
            self._L.dsp_speaker_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def llr(self, mfcc, per_frame: bool = False):
        """mfcc: cuda float32 [n_clips][T][d] -> (llr_mean int64 [n], label int32 [n][, ll_target, ll_ubm int64 [n][T]])."""
        import torch
        n, t = mfcc.shape[0], mfcc.shape[1]
        mean = torch.empty(n, dtype=torch.int64, device=mfcc.device)
        label = torch.empty(n, dtype=torch.int32, device=mfcc.device)
        lt = torch.empty((n, t), dtype=torch.int64, device=mfcc.device) if per_frame else None
        lu = torch.empty((n, t), dtype=torch.int64, device=mfcc.device) if per_frame else None
        _lib.check(self._L.dsp_speaker_llr_device(self._h, mfcc.contiguous().data_ptr(), n, t, mean.data_ptr(), label.data_ptr(),
                                                  lt.data_ptr() if per_frame else None, lu.data_ptr() if per_frame else None,
                                                  _stream(mfcc)), "dsp_speaker_llr_device")
        return (mean, label, lt, lu) if per_frame else (mean, label)


def upsample_linear(x, new_size: int):
    """x: cuda float32 [n_clips][old] (or [old]) -> [n_clips][new_size]; numpy input goes through the host entry point."""
    L = _lib.load()
    if isinstance(x, np.ndarray):
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(new_size, np.float32)
        _lib.check(L.dsp_upsample_linear_host(x.ctypes.data, x.size, out.ctypes.data, new_size), "dsp_upsample_linear_host")
        return out
    import torch
    squeeze = x.dim() == 1
    x2 = x[None] if squeeze else x
    assert x2.stride(1) == 1
    out = torch.empty((x2.shape[0], new_size), dtype=torch.float32, device=x.device)
    _lib.check(L.dsp_upsample_linear_device(x2.data_ptr(), x2.shape[0], x2.shape[1], x2.stride(0), out.data_ptr(), new_size,
                                            new_size, _stream(x)), "dsp_upsample_linear_device")
    return out[0] if squeeze else out


__all__ = ["StopModel", "SpeakerModel", "upsample_linear", "MfccPlan", "default_config"]
