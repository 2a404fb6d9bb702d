"""Host-side mirror of cepstrum/scrubjay_infer.c: clip -> MFCC -> mean|std pooling
(`mfcc_stats`, :19-77) -> Scaler + RBF-SVM (`ort_predict`, :105-141, graph
cepstrum/scrubjay_svm.onnx) over the C ABI, all on HBM-resident tensors.

The reference's MFCC front end is aubio (unvendored, unpinned) with a 2048/1024 framing;
here the front end is this library's MFCC chain with `n_mfcc = 20`,
so feature VALUES are not comparable with the reference's -- the pooling and the SVM
arithmetic are.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from .mfcc import MfccPlan, default_config


class SvmModel:
    """dsp_svm: Scaler + SVMClassifier attributes as decoded from the ONNX file."""

    def __init__(self, attrs: dict, device: int = 0):
        self._L = _lib.load()
        a = {k: np.ascontiguousarray(attrs[k], np.float32) for k in ("offset", "scale", "sv", "coef")}
        self.n_features = a["offset"].size
        self.n_sv = a["coef"].size
        assert a["sv"].shape == (self.n_sv, self.n_features)
        h = C.c_void_p()
        _lib.check(self._L.dsp_svm_create(device, self.n_features, self.n_sv, a["offset"].ctypes.data, a["scale"].ctypes.data,
                                          a["sv"].ctypes.data, a["coef"].ctypes.data,
                                          float(np.ravel(attrs["kernel_params"])[0]), float(np.ravel(attrs["rho"])[0]),
                                          float(np.ravel(attrs["prob_a"])[0]), float(np.ravel(attrs["prob_b"])[0]), C.byref(h)),
                   "dsp_svm_create")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.dsp_svm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def predict(self, feat):
        """feat: cuda float32 [n][n_features] -> (labels int32, decision float32, prob1 float32)."""
        import torch
        n = feat.shape[0]
        labels = torch.empty(n, dtype=torch.int32, device=feat.device)
        dec = torch.empty(n, dtype=torch.float32, device=feat.device)
        p1 = torch.empty(n, dtype=torch.float32, device=feat.device)
        st = C.c_void_p(torch.cuda.current_stream(feat.device).cuda_stream)
        _lib.check(self._L.dsp_svm_predict_device(self._h, feat.contiguous().data_ptr(), n, labels.data_ptr(), dec.data_ptr(),
                                                  p1.data_ptr(), st), "dsp_svm_predict_device")
        return labels, dec, p1


def mfcc_stats(mfcc):
    """mfcc: cuda float32 [n_clips][T][n_coef] -> cuda float32 [n_clips][2*n_coef] (mean | std)."""
    import torch
    n, t, c = mfcc.shape
    feat = torch.empty((n, 2 * c), dtype=torch.float32, device=mfcc.device)
    st = C.c_void_p(torch.cuda.current_stream(mfcc.device).cuda_stream)
    _lib.check(_lib.load().dsp_mfcc_stats_device(mfcc.contiguous().data_ptr(), n, t, c, feat.data_ptr(), st), "dsp_mfcc_stats_device")
    return feat


def scrubjay_infer_config(sample_rate: int = 16000, aubio: bool = True):
    """The front end cepstrum/scrubjay_infer.c itself runs (:9-13, :28-30, :39-53).  aubio=True (default):
    dsp_mfcc_scrubjay_infer_config -- aubio 0.4's semantics for the calls the file makes: streaming 2048 / 1024 frames with zero
    history and the zero-padded last hop (T = ceil(n / 1024)), periodic Hann, MAGNITUDE spectrum, the 40-filter Slaney bank,
    log10, orthonormal DCT-II, 20 coefficients (aubio is unvendored: restated from its published algorithm, parity unpinned).
    aubio=False: only the file's numbers (2048 / 1024 / 40 / 20) on this library's mfcc.c semantics (round 2's config5_2048)."""
    if not aubio:
        return default_config(sample_rate=sample_rate, n_fft=2048, frame_length=2048, hop_length=1024, n_mels=40, n_mfcc=20,
                              fmin=0.0, fmax=sample_rate / 2.0)
    from .lib import MfccConfig
    cfg = MfccConfig()
    _lib.load().dsp_mfcc_scrubjay_infer_config(C.byref(cfg), int(sample_rate))
    return cfg


class ScrubJay:
    """clips -> label / probability, the scrubjay_infer.c main loop (:158-177) for a batch in HBM."""

    def __init__(self, svm_attrs: dict, device: int = 0, n_mfcc: int = 20, config=None):
        """config: an MfccConfig (its n_mfcc must be half the SVM's feature count), e.g. scrubjay_infer_config();
        default: the reference's 512-point keyword front end with n_mfcc coefficients (BASELINE config 5)."""
        self.plan = MfccPlan(config if config is not None else default_config(n_mfcc=n_mfcc), device)
        self.svm = SvmModel(svm_attrs, device)

    def __call__(self, clips, max_frames: int = 1 << 20, fused: bool = True):
        """-> (labels int32, decision float32, prob1 float32, features float32 [n][2 n_mfcc]).  fused: one kernel from PCM to
        label (dsp_scrubjay_fused_device, BASELINE config 5); otherwise MFCC -> pooling -> SVM as three kernels."""
        if not fused:
            mfcc = self.plan.clips(clips, max_frames)
            feat = mfcc_stats(mfcc)
            return self.svm.predict(feat) + (feat,)
        import torch
        n = clips.shape[0]
        labels = torch.empty(n, dtype=torch.int32, device=clips.device)
        dec = torch.empty(n, dtype=torch.float32, device=clips.device)
        p1 = torch.empty(n, dtype=torch.float32, device=clips.device)
        feat = torch.empty((n, self.svm.n_features), dtype=torch.float32, device=clips.device)
        st = C.c_void_p(torch.cuda.current_stream(clips.device).cuda_stream)
        assert clips.dim() == 2 and clips.stride(1) == 1
        _lib.check(_lib.load().dsp_scrubjay_fused_device(self.plan._h, self.svm._h, clips.data_ptr(), n, clips.shape[1], clips.stride(0),
                                                         int(min(max_frames, 1 << 30)), labels.data_ptr(), dec.data_ptr(), p1.data_ptr(),
                                                         feat.data_ptr(), st), "dsp_scrubjay_fused_device")
        return labels, dec, p1, feat

    def pcm16(self, pcm, max_frames: int = 1 << 20, stereo_mode: int = 0):
        """The fused clip -> label kernel on int16 PCM [n_clips][n] (mono) or [n_clips][n][2] (interleaved stereo: channel 0 or the
        channels' average), converted in the kernel's load (dsp_scrubjay_fused_pcm16_device): the float path's results, bit for bit."""
        import torch
        assert pcm.dtype == torch.int16 and pcm.dim() in (2, 3) and pcm.stride(-1) == 1
        channels = 2 if pcm.dim() == 3 else 1
        n = pcm.shape[0]
        labels = torch.empty(n, dtype=torch.int32, device=pcm.device)
        dec = torch.empty(n, dtype=torch.float32, device=pcm.device)
        p1 = torch.empty(n, dtype=torch.float32, device=pcm.device)
        feat = torch.empty((n, self.svm.n_features), dtype=torch.float32, device=pcm.device)
        st = C.c_void_p(torch.cuda.current_stream(pcm.device).cuda_stream)
        _lib.check(_lib.load().dsp_scrubjay_fused_pcm16_device(self.plan._h, self.svm._h, pcm.data_ptr(), n, pcm.shape[1], pcm.stride(0) // channels,
                                                               channels, int(stereo_mode), int(min(max_frames, 1 << 30)), labels.data_ptr(), dec.data_ptr(),
                                                               p1.data_ptr(), feat.data_ptr(), st), "dsp_scrubjay_fused_pcm16_device")
        return labels, dec, p1, feat

    def ragged(self, signal, offsets, max_frames: int = 1 << 20, stereo_mode: int = 0):
        """Clips of different lengths in ONE launch (the files scrubjay_infer.c:158-176 loops over): `signal` is a flat cuda buffer --
        float32 [total], int16 [total] or interleaved stereo int16 [total][2] -- and clip c is samples [offsets[c], offsets[c + 1]).
        Results as a one-clip call per clip gives them (dsp_scrubjay_fused_ragged_device / _pcm16_device)."""
        import torch
        off, n = offsets if isinstance(offsets, tuple) else _lib.c_offsets(offsets)      # (a prepared (ctypes array, n_clips) pair: no conversion per call)
        assert signal.is_cuda and signal.stride(-1) == 1 and int(off[n]) <= signal.shape[0]
        labels = torch.empty(n, dtype=torch.int32, device=signal.device)
        dec = torch.empty(n, dtype=torch.float32, device=signal.device)
        p1 = torch.empty(n, dtype=torch.float32, device=signal.device)
        feat = torch.empty((n, self.svm.n_features), dtype=torch.float32, device=signal.device)
        st = C.c_void_p(torch.cuda.current_stream(signal.device).cuda_stream)
        mf = int(min(max_frames, 1 << 30))
        if signal.dtype == torch.float32:
            assert signal.dim() == 1
            _lib.check(_lib.load().dsp_scrubjay_fused_ragged_device(self.plan._h, self.svm._h, signal.data_ptr(), n, off, mf, labels.data_ptr(), dec.data_ptr(),
                                                                    p1.data_ptr(), feat.data_ptr(), st), "dsp_scrubjay_fused_ragged_device")
        else:
            assert signal.dtype == torch.int16 and signal.dim() in (1, 2)
            _lib.check(_lib.load().dsp_scrubjay_fused_ragged_pcm16_device(self.plan._h, self.svm._h, signal.data_ptr(), n, off, signal.dim(), int(stereo_mode), mf,
                                                                          labels.data_ptr(), dec.data_ptr(), p1.data_ptr(), feat.data_ptr(), st),
                       "dsp_scrubjay_fused_ragged_pcm16_device")
        return labels, dec, p1, feat
